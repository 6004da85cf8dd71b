"""Tuning build only (SRHIP_TUNING_BUILD=1): wall-clock phases of the attention backward workgroups (100 MHz timestamps)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from semireward_amd import ops, _lib

B, N, H = 16, 257, 6
D = H * 64
dev = "cuda:0"
qkv = torch.randn(B * N, 3 * D, device=dev).to(torch.bfloat16)
out = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, N, device=dev)
ops.attn_fwd(qkv, out, lse, B, N, H, 0.125)
dout = torch.randn(B * N, D, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, N, device=dev)
for _ in range(3):
    ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, N, H, 0.125)
torch.cuda.synchronize()
n = 4 * 2 * B * H
buf = (ctypes.c_longlong * n)()
lib = _lib.lib()
lib.srhip_attn_debug.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.srhip_attn_debug(buf, n) == 0
t = np.array(buf, dtype=np.int64).reshape(2, B * H, 4)[:, :, :3].astype(np.float64) / 100.0      # us
t0 = t[:, :, 0].min()
for z, name in ((0, "dq "), (1, "dkv")):
    a = t[z]
    print("%s: start %5.1f..%5.1f us after the first workgroup | staging %5.1f (max %5.1f) | loop %5.1f (max %5.1f) | end at %5.1f (max %5.1f)" % (
        name, (a[:, 0] - t0).min(), (a[:, 0] - t0).max(), (a[:, 1] - a[:, 0]).mean(), (a[:, 1] - a[:, 0]).max(),
        (a[:, 2] - a[:, 1]).mean(), (a[:, 2] - a[:, 1]).max(), (a[:, 2] - t0).mean(), (a[:, 2] - t0).max()))
