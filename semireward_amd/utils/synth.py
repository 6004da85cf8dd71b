"""Deterministic synthetic parameters / batches (numpy PCG64 -> identical on every host).

Used by bench.py, smoke(), the tests and the golden-vector generator.  There is no
network for datasets or checkpoints, so weights are random-init of the reference's
architecture and batches have the reference's shapes (SURVEY.md 8(d) "Synthetic inputs").
"""
import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def synth_tensor(rng, name, shape):
    shape = tuple(int(s) for s in shape)
    if "norm" in name and name.endswith("weight"):
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if name in ("cls_token", "pos_embed"):
        return (0.02 * rng.standard_normal(shape)).astype(np.float32)
    if "embedding" in name:
        return rng.standard_normal(shape).astype(np.float32)
    if len(shape) == 1:
        return (0.02 * rng.standard_normal(shape)).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    b = 1.0 / np.sqrt(fan_in)
    return rng.uniform(-b, b, size=shape).astype(np.float32)


def synth_params(names_shapes, seed):
    """names_shapes: iterable of (name, shape) or dict -> dict name -> float32 ndarray."""
    if isinstance(names_shapes, dict):
        names_shapes = list(names_shapes.items())
    rng = _rng(seed)
    return {n: synth_tensor(rng, n, s) for n, s in names_shapes}


def synth_batch(seed, num_lb, num_ulb, img_size=32, num_classes=100, ulb_dest_len=50000, in_chans=3):
    """x ~ N(0,1) (CIFAR is mean/std normalised), y ~ U{0..C-1}, idx_ulb unique in [0, ulb_dest_len)."""
    rng = _rng(seed)
    shp = (in_chans, img_size, img_size)
    return dict(
        x_lb=rng.standard_normal((num_lb,) + shp).astype(np.float32),
        y_lb=rng.integers(0, num_classes, size=(num_lb,), dtype=np.int64),
        idx_ulb=rng.permutation(ulb_dest_len)[:num_ulb].astype(np.int64),
        x_ulb_w=rng.standard_normal((num_ulb,) + shp).astype(np.float32),
        x_ulb_s=rng.standard_normal((num_ulb,) + shp).astype(np.float32),
    )


def synth_droppath(seed, probs, batch):
    """Per-sample DropPath scales [depth, 2, batch]: 0 or 1/keep (timm DropPath semantics)."""
    rng = _rng(seed)
    depth = len(probs)
    out = np.ones((depth, 2, batch), dtype=np.float32)
    for i, p in enumerate(probs):
        if p > 0.0:
            keep = 1.0 - p
            out[i] = (rng.random((2, batch)) < keep).astype(np.float32) / np.float32(keep)
    return out
