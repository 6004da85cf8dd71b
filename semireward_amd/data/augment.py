"""Device-side weak / strong augmentation of the USB CV datasets: one HIP launch per batch instead of 12 PIL worker
processes.

Mirrors the transform pipelines of ``semilearn/datasets/cv_datasets/cifar.py:34-49`` (RandomCrop with reflect padding, RandomHorizontalFlip,
[RandAugment(3, 5)], ToTensor, Normalize) and ``semilearn/datasets/augmentation/randaugment.py`` (14 ops with their magnitude ranges, Cutout
with the (125, 123, 114) fill; ``RandAugment.__call__`` :189-196).  The pixel work is srhip_augment (csrc/augment.hip), bit-exact with
Pillow's algorithms; this module draws the random decisions (numpy Generator; the reference uses python ``random`` / ``np.random`` global
state) and turns them into the per-image parameter blocks -- including the part Pillow computes in its Python layer in float64 (rotation
matrix rounded to 15 decimals, 16.16 fixed-point coefficients of the nearest-neighbour affine walk).
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib, ops

OPS = ["AutoContrast", "Brightness", "Color", "Contrast", "Equalize", "Identity", "Posterize", "Rotate", "Sharpness", "ShearX", "ShearY",
       "Solarize", "TranslateX", "TranslateY"]                       # augment_list() order (randaugment.py:149-166)
RANGES = [(0, 1), (0.05, 0.95), (0.05, 0.95), (0.05, 0.95), (0, 1), (0, 1), (4, 8), (-30, 30), (0.05, 0.95), (-0.3, 0.3), (-0.3, 0.3),
          (0, 256), (-0.3, 0.3), (-0.3, 0.3)]
IPN, DPN, MAX_OPS = 64, 32, 4


def _fix(v):
    return int(math.floor(v * 65536.0 + 0.5))


def _affine_matrix(op, v, S):
    name = OPS[op]
    if name == "Rotate":                    # Image.rotate: inverse matrix about the centre, entries rounded to 15 decimals
        ang = -math.radians(v % 360.0)
        m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
        c = S / 2.0
        m[2] = m[0] * (-c) + m[1] * (-c) + m[2]
        m[5] = m[3] * (-c) + m[4] * (-c) + m[5]
        m[2] += c; m[5] += c
        return m
    return {"ShearX": [1, v, 0, 0, 1, 0], "ShearY": [1, 0, 0, v, 1, 0], "TranslateX": [1, 0, v * S, 0, 1, 0],
            "TranslateY": [1, 0, 0, 0, 1, v * S]}[name]


class GpuAugment:
    def __init__(self, size, pad, mean, std, n_ops=3, device="cuda", seed=0):
        assert n_ops <= MAX_OPS
        self.size, self.pad, self.n_ops, self.device = size, pad, n_ops, torch.device(device)
        self.mean = (ctypes.c_float * 3)(*[float(m) for m in mean])
        self.std = (ctypes.c_float * 3)(*[float(s) for s in std])
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self._scratch = None

    def draw(self, B, strong, src_hw=None):
        """The random decisions of one batch: crop offsets in [0, H0 + 2 pad - size], flips, and for strong: op picks, magnitudes, cutout."""
        H0, W0 = src_hw or (self.size, self.size)
        r = self.rng
        d = dict(i=r.integers(0, H0 + 2 * self.pad - self.size + 1, size=B), j=r.integers(0, W0 + 2 * self.pad - self.size + 1, size=B),
                 flip=r.random(B) < 0.5)
        if strong:
            ops_ = r.integers(0, len(OPS), size=(B, self.n_ops))
            lo = np.array([a for a, _ in RANGES], dtype=np.float64)[ops_]
            hi = np.array([b for _, b in RANGES], dtype=np.float64)[ops_]
            d.update(ops=ops_, vals=lo + (hi - lo) * r.random((B, self.n_ops)), cut_v=0.5 * r.random(B), ux=r.uniform(0, self.size, B),
                     uy=r.uniform(0, self.size, B))
        return d

    def pack(self, d, src_index=None):
        """Parameter blocks of srhip_augment (layout: csrc/augment.hip)."""
        B, S = len(d["i"]), self.size
        ip, dp = np.zeros((B, IPN), dtype=np.int32), np.zeros((B, DPN), dtype=np.float64)
        ip[:, 0], ip[:, 1], ip[:, 2] = d["i"], d["j"], d["flip"]
        ip[:, 4] = -1
        ip[:, 8] = np.arange(B) if src_index is None else src_index
        if "ops" in d:
            n = d["ops"].shape[1]
            ip[:, 3] = n
            for b in range(B):
                for k in range(n):
                    op, v = int(d["ops"][b, k]), float(d["vals"][b, k])
                    q, e = ip[b, 16 + 12 * k:], dp[b, 8 * k:]
                    q[0], e[0] = op, v
                    if OPS[op] == "Posterize":
                        q[8] = ~(2 ** (8 - max(1, int(v))) - 1) & 0xFF
                    elif OPS[op] in ("Rotate", "ShearX", "ShearY", "TranslateX", "TranslateY"):
                        a = _affine_matrix(op, v, S)
                        if a[1] == 0 and a[3] == 0:            # Pillow: ImagingScaleAffine (float64 walk)
                            q[1], e[1], e[2], e[3], e[4] = 1, a[2] + a[0] * 0.5, a[5] + a[4] * 0.5, a[0], a[4]
                        else:                                  # Pillow: affine_fixed (16.16)
                            q[2:8] = [_fix(a[0]), _fix(a[1]), _fix(a[2] + a[0] * 0.5 + a[1] * 0.5), _fix(a[3]), _fix(a[4]),
                                      _fix(a[5] + a[3] * 0.5 + a[4] * 0.5)]
                cv = float(d["cut_v"][b])                      # Cutout / CutoutAbs (randaugment.py:116-146)
                if cv > 0.0:
                    v = cv * S
                    x0, y0 = int(max(0, float(d["ux"][b]) - v / 2.0)), int(max(0, float(d["uy"][b]) - v / 2.0))
                    ip[b, 4:8] = [x0, y0, int(min(S, x0 + v)), int(min(S, y0 + v))]
        return ip, dp

    def __call__(self, src_u8, strong, draws=None, src_index=None, return_u8=False):
        """src_u8: uint8 [N, H, W, 3] on the device.  Returns fp32 [B, 3, size, size] (and the uint8 HWC image before normalisation)."""
        assert src_u8.dtype == torch.uint8 and src_u8.is_cuda and src_u8.is_contiguous() and src_u8.shape[-1] == 3
        N, H0, W0, _ = src_u8.shape
        B = N if src_index is None else len(src_index)
        d = draws if draws is not None else self.draw(B, strong, (H0, W0))
        ip, dp = self.pack(d, src_index)
        ipt, dpt = torch.from_numpy(ip).to(self.device, non_blocking=True), torch.from_numpy(dp).to(self.device, non_blocking=True)
        S = self.size
        if self._scratch is None or self._scratch.numel() < B * 2 * S * S * 3:
            self._scratch = torch.empty(B * 2 * S * S * 3, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, 3, S, S, dtype=torch.float32, device=self.device)
        u8 = torch.empty(B, S, S, 3, dtype=torch.uint8, device=self.device) if return_u8 else None
        _lib.check(_lib.lib().srhip_augment(src_u8.data_ptr(), N, H0, W0, B, S, self.pad, ipt.data_ptr(), dpt.data_ptr(), self._scratch.data_ptr(),
                                            out.data_ptr(), u8.data_ptr() if return_u8 else None, self.mean, self.std, ops._s()), "srhip_augment")
        return (out, u8) if return_u8 else out


class DevicePrefetcher:
    """Host -> device double buffering for an iterator of dict batches (process_batch's H2D copy, algorithmbase.py:282-306, taken off the
    step's critical path): batch t+1 is pinned and copied on a dedicated stream while batch t trains."""

    def __init__(self, it, device="cuda"):
        self.it, self.device = iter(it), torch.device(device)
        self.stream = ops.concurrent_stream(self.device)      # (seen to run beside the training stream: HIP streams may share a hardware queue)
        self._next = None
        self._preload()

    def _to(self, v):
        if isinstance(v, dict):
            return {k: self._to(x) for k, x in v.items()}
        if torch.is_tensor(v):
            return (v if v.is_pinned() else v.pin_memory()).to(self.device, non_blocking=True)
        return v

    def _preload(self):
        try:
            b = next(self.it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.stream):
            self._next = {k: self._to(v) for k, v in b.items()}

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        torch.cuda.current_stream().wait_stream(self.stream)
        b = self._next

        def rec(v):
            if isinstance(v, dict):
                for x in v.values():
                    rec(x)
            elif torch.is_tensor(v):
                v.record_stream(torch.cuda.current_stream())
        rec(b)
        self._preload()
        return b
