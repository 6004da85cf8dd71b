"""Tuning build only (SRHIP_TUNING_BUILD=1): wall-clock phases of the fused LN2 + MLP workgroups (one 128-row tile each)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from semireward_amd import ops, _lib

dev = "cuda:0"
lib = _lib.lib()
lib.srhip_mlp_debug.argtypes = [ctypes.c_void_p, ctypes.c_int]
D, Hd = 384, 1536
g, b = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
W1 = (torch.randn(Hd, D, device=dev) * 0.05).to(torch.bfloat16)
W2 = (torch.randn(D, Hd, device=dev) * 0.02).to(torch.bfloat16)
b1, b2 = torch.randn(Hd, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
for imgs in (1, 73, 127, 200):
    M = imgs * 257
    x = torch.randn(M, D, device=dev)
    for _ in range(3):
        ops.mlp_fused(x, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd)
    torch.cuda.synchronize()
    nwg = min((M + 127) // 128, 256)
    buf = (ctypes.c_longlong * (4 * nwg))()
    assert lib.srhip_mlp_debug(buf, 4 * nwg) == 0
    a = np.array(buf, dtype=np.int64).reshape(nwg, 4).astype(np.float64) / 100.0
    print("%3d images, %3d WGs (last tile of each WG): LayerNorm prologue %5.1f (max %5.1f) | main loop %5.1f (max %5.1f) | epilogue %5.1f (max %5.1f) us" % (
        imgs, nwg, (a[:, 1] - a[:, 0]).mean(), (a[:, 1] - a[:, 0]).max(), (a[:, 2] - a[:, 1]).mean(), (a[:, 2] - a[:, 1]).max(),
        (a[:, 3] - a[:, 2]).mean(), (a[:, 3] - a[:, 2]).max()))
