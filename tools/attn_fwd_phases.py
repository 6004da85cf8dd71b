"""Tuning build only (SRHIP_TUNING_BUILD=1): wall-clock phases of the attention forward workgroups (100 MHz timestamps)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from semireward_amd import ops, _lib

N, H = 257, 6
D = H * 64
dev = "cuda:0"
lib = _lib.lib()
lib.srhip_attn_debug.argtypes = [ctypes.c_void_p, ctypes.c_int]
for B in (16, 43, 86, 200):
    qkv = torch.randn(B * N, 3 * D, device=dev).to(torch.bfloat16)
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.attn_fwd(qkv, out, None, B, N, H, 0.125)
    torch.cuda.synchronize()
    n = 4 * B * H
    buf = (ctypes.c_longlong * n)()
    assert lib.srhip_attn_debug(buf, n) == 0
    a = np.array(buf, dtype=np.int64).reshape(B * H, 4)[:, :3].astype(np.float64) / 100.0      # us
    t0 = a[:, 0].min()
    print("B=%3d (%4d WGs): start spread %5.1f us | staging %5.1f (max %5.1f) | loop %5.1f (max %5.1f) | last end %5.1f" % (
        B, B * H, (a[:, 0] - t0).max(), (a[:, 1] - a[:, 0]).mean(), (a[:, 1] - a[:, 0]).max(), (a[:, 2] - a[:, 1]).mean(),
        (a[:, 2] - a[:, 1]).max(), (a[:, 2] - t0).max()))
