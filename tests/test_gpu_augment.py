"""Device-side augmentation (csrc/augment.hip, semireward_amd/data/augment.py) against the numpy oracle (pinned bit for bit to the reference's
Pillow-backed RandAugment functions) and against the golden vectors produced by those functions: byte-exact images, exact fp32 tensors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment_ref as A                    # noqa: E402
from oracle.gen_golden import synth_image              # noqa: E402
from semireward_amd.data.augment import OPS, RANGES, DevicePrefetcher, GpuAugment   # noqa: E402

DEV = "cuda:0"
MEAN, STD = (0.507, 0.487, 0.441), (0.267, 0.256, 0.276)        # cifar100 statistics of the reference (cv_datasets/cifar.py:16-17)


def test_op_tables_match_reference():
    assert OPS == A.OPS and RANGES == A.RANGES


def _run(aug, imgs, draws, strong):
    out, u8 = aug(torch.from_numpy(np.stack(imgs)).to(DEV), strong, draws=draws, return_u8=True)
    return out.cpu().numpy(), u8.cpu().numpy()


@pytest.mark.parametrize("S", [32, 96])
def test_every_op_bit_exact(S):
    """Each of the 14 ops alone, 12 random magnitudes each, no crop shift / flip / cutout: bytes equal the oracle's."""
    rng = np.random.Generator(np.random.PCG64(S))
    aug = GpuAugment(S, 0, MEAN, STD, n_ops=1, device=DEV)
    for oi in range(len(OPS)):
        B = 12
        imgs = [synth_image(100 * oi + t, S, S, t % 3) for t in range(B)]
        lo, hi = RANGES[oi]
        vals = lo + (hi - lo) * rng.random((B, 1))
        d = dict(i=np.zeros(B, int), j=np.zeros(B, int), flip=np.zeros(B, bool), ops=np.full((B, 1), oi), vals=vals, cut_v=np.zeros(B),
                 ux=np.zeros(B), uy=np.zeros(B))
        _, u8 = _run(aug, imgs, d, True)
        for t in range(B):
            want = A.apply_op(oi, imgs[t], float(vals[t, 0]))
            assert np.array_equal(u8[t], want), (OPS[oi], t, int(np.abs(u8[t].astype(int) - want.astype(int)).max()))


def test_golden_chains_and_ops(golden):
    """The reference's own outputs (tests/golden/augment.npz): single ops and whole RandAugment(3) + Cutout chains, square images."""
    g = golden("augment")
    for oi in range(len(OPS)):
        for t in range(4):
            seed, H, W, kind = [int(v) for v in g[f"op/{oi}/{t}/meta"]]
            if H != W:
                continue
            aug = GpuAugment(H, 0, MEAN, STD, n_ops=1, device=DEV)
            d = dict(i=np.zeros(1, int), j=np.zeros(1, int), flip=np.zeros(1, bool), ops=np.array([[oi]]), vals=np.array([[float(g[f"op/{oi}/{t}/v"])]]),
                     cut_v=np.zeros(1), ux=np.zeros(1), uy=np.zeros(1))
            _, u8 = _run(aug, [synth_image(seed, H, W, kind)], d, True)
            assert np.array_equal(u8[0], g[f"op/{oi}/{t}/out"]), (OPS[oi], t)
    for t in range(6):
        seed, H, W, kind = [int(v) for v in g[f"chain/{t}/meta"]]
        if H != W:
            continue
        aug = GpuAugment(H, 0, MEAN, STD, n_ops=3, device=DEV)
        cv, ux, uy = [float(v) for v in g[f"chain/{t}/cut"]]
        d = dict(i=np.zeros(1, int), j=np.zeros(1, int), flip=np.zeros(1, bool), ops=g[f"chain/{t}/ops"][None], vals=g[f"chain/{t}/vals"][None],
                 cut_v=np.array([cv]), ux=np.array([ux]), uy=np.array([uy]))
        _, u8 = _run(aug, [synth_image(seed, H, W, kind)], d, True)
        assert np.array_equal(u8[0], g[f"chain/{t}/out"]), t


@pytest.mark.parametrize("S,pad", [(32, 4), (96, 12)])
def test_full_pipelines_match_oracle(S, pad):
    """transform_weak / transform_strong with drawn randomness: crop window of the reflect-padded image, flip, 3 ops, cutout, normalise --
    bytes and fp32 tensors equal the oracle's, for a batch that shares source images through src_index."""
    B = 24
    aug = GpuAugment(S, pad, MEAN, STD, n_ops=3, device=DEV, seed=7)
    imgs = [synth_image(900 + t, S, S, t % 3) for t in range(B)]
    for strong in (False, True):
        d = aug.draw(B, strong)
        out, u8 = _run(aug, imgs, d, strong)
        for t in range(B):
            if strong:
                want = A.strong(imgs[t], pad, S, int(d["i"][t]), int(d["j"][t]), bool(d["flip"][t]), d["ops"][t], d["vals"][t], float(d["cut_v"][t]),
                                float(d["ux"][t]), float(d["uy"][t]), MEAN, STD)
            else:
                want = A.weak(imgs[t], pad, S, int(d["i"][t]), int(d["j"][t]), bool(d["flip"][t]), MEAN, STD)
            assert np.array_equal(out[t], want), (strong, t)
    if True:                                         # several augmented views of the same stored image (the K + 1 passes share the source)
        idx = np.array([3, 3, 0, 5])
        d = aug.draw(4, True)
        o = aug(torch.from_numpy(np.stack(imgs)).to(DEV), True, draws=d, src_index=idx).cpu().numpy()
        for t in range(4):
            want = A.strong(imgs[idx[t]], pad, S, int(d["i"][t]), int(d["j"][t]), bool(d["flip"][t]), d["ops"][t], d["vals"][t], float(d["cut_v"][t]),
                            float(d["ux"][t]), float(d["uy"][t]), MEAN, STD)
            assert np.array_equal(o[t], want)


def test_device_prefetcher_roundtrip():
    batches = [{"x_lb": torch.randn(4, 3, 8, 8), "y_lb": torch.arange(4) + i, "x_ulb_w": {"input_ids": torch.arange(12).view(3, 4) + i}} for i in range(5)]
    got = list(DevicePrefetcher(batches, DEV))
    assert len(got) == 5
    for a, b in zip(batches, got):
        assert b["x_lb"].is_cuda and torch.equal(b["x_lb"].cpu(), a["x_lb"]) and torch.equal(b["y_lb"].cpu(), a["y_lb"])
        assert torch.equal(b["x_ulb_w"]["input_ids"].cpu(), a["x_ulb_w"]["input_ids"])


def test_whole_transforms_against_the_reference_chain(golden):
    """srhip_augment against tests/golden/augment_tv.npz: the reference's transform_weak / transform_strong executed as torchvision's PIL op
    sequence + the reference's RandAugment class + torch's ToTensor / Normalize (oracle/gen_golden.py:gen_augment_tv) -- bytes of the strong
    view and fp32 tensors of both views, exact, with the fixture's draws."""
    g = golden("augment_tv")
    mean, std = tuple(float(v) for v in g["meta/mean"]), tuple(float(v) for v in g["meta/std"])
    cases = {}
    for n in range(int(g["meta/n"])):
        seed, S, kind, pad, i, j, flip = [int(v) for v in g[f"case/{n}/meta"]]
        cases.setdefault((S, pad), []).append((n, seed, kind, i, j, flip))
    assert set(cases) == {(32, 4), (96, 12)}
    for (S, pad), cs in cases.items():
        aug = GpuAugment(S, pad, mean, std, n_ops=3, device=DEV, seed=1)
        imgs = [synth_image(seed, S, S, kind) for _, seed, kind, _, _, _ in cs]
        d = dict(i=np.array([c[3] for c in cs]), j=np.array([c[4] for c in cs]), flip=np.array([bool(c[5]) for c in cs]))
        out, _ = _run(aug, imgs, d, False)
        for t, c in enumerate(cs):
            assert np.array_equal(out[t], g[f"case/{c[0]}/weak"]), ("weak", c[0])
        d.update(ops=np.stack([g[f"case/{c[0]}/ops"] for c in cs]), vals=np.stack([g[f"case/{c[0]}/vals"] for c in cs]),
                 cut_v=np.array([g[f"case/{c[0]}/cut"][0] for c in cs]), ux=np.array([g[f"case/{c[0]}/cut"][1] for c in cs]),
                 uy=np.array([g[f"case/{c[0]}/cut"][2] for c in cs]))
        out, u8 = _run(aug, imgs, d, True)
        for t, c in enumerate(cs):
            assert np.array_equal(u8[t], g[f"case/{c[0]}/strong_u8"]), ("strong bytes", c[0])
            assert np.array_equal(out[t], g[f"case/{c[0]}/strong"]), ("strong", c[0])
