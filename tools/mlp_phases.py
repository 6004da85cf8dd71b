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
Wp = (torch.randn(D, D, device=dev) * 0.05).to(torch.bfloat16)
bp, gn, bn = torch.randn(D, device=dev) * 0.1, torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
proj = len(sys.argv) > 1 and sys.argv[1] == "proj"
for imgs in (1, 73, 95, 105, 127, 200):
    M = imgs * 257
    x = torch.randn(M, D, device=dev)
    ao = torch.randn(M, D, device=dev).to(torch.bfloat16)
    rs = torch.full((imgs,), 1.0 / 0.9, device=dev)
    ln = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        if proj:
            ops.mlp_fused_proj(x, ao, Wp, bp, rs, g, b, 1e-6, W1, b1, W2, b2, rs, 257, M, D, Hd, ln_next=ln, next_gamma=gn, next_beta=bn)
        else:
            ops.mlp_fused(x, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd)
    torch.cuda.synchronize()
    nwg = min((M + 127) // 128, 256)
    buf = (ctypes.c_longlong * (8 * nwg))()
    assert lib.srhip_mlp_debug(buf, 8 * nwg) == 0
    a = np.array(buf, dtype=np.int64).reshape(nwg, 8).astype(np.float64) / 100.0
    d = lambda i, j: ((a[:, j] - a[:, i]).mean(), (a[:, j] - a[:, i]).max())   # noqa: E731
    if proj:
        print("%3d images, %3d WGs: prologue %5.1f (max %5.1f) | projection %5.1f (%5.1f) | residual + LN %5.1f (%5.1f) | MLP loop %5.1f (%5.1f) | "
              "epilogue + next LN %5.1f (%5.1f) | total %5.1f (%5.1f) us" % (imgs, nwg, *d(0, 1), *d(1, 4), *d(4, 5), *d(5, 2), *d(2, 3), *d(0, 3)))
    else:
        print("%3d images, %3d WGs (last tile of each WG): LayerNorm prologue %5.1f (max %5.1f) | main loop %5.1f (max %5.1f) | epilogue %5.1f (max %5.1f) us"
              % (imgs, nwg, *d(0, 1), *d(1, 2), *d(2, 3)))
