"""Tuning build only (SRHIP_TUNING_BUILD=1 python -m semireward_amd.build --force): wall-clock phases of the two-wave-group GEMM's workgroups
(wave 0's stamps: K loop, closing barrier + vmcnt(0), epilogue) on the D = 768 shapes.  GPU box."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SRHIP_GEMM"] = "big256"
import numpy as np
import torch
from semireward_amd import ops, _lib
dev = "cuda:0"
lib = _lib.lib()
lib.srhip_gemm_debug.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (M, N, K, epi, name) in [(13952, 3072, 768, ops.EPI_BF16, "fc1 bf16+bias"), (13952, 3072, 768, ops.EPI_GELU_BF16, "fc1 GELU"), (13952, 768, 3072, ops.EPI_BF16, "fc2 bf16"),
                             (13952, 768, 3072, ops.EPI_RESID_F32, "fc2 resid f32"), (8192, 8192, 8192, ops.EPI_BF16, "8k")]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Bm = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    C = torch.zeros(M, N, dtype=torch.float32 if epi == ops.EPI_RESID_F32 else torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm_nt(epi, A, Bm, C, M, N, K, bias=bias)
    torch.cuda.synchronize()
    nwg = min(256, ((M + 255) // 256) * ((N + 255) // 256))
    buf = (ctypes.c_longlong * (16 * nwg))()
    assert lib.srhip_gemm_debug(buf, 16 * nwg) == 0
    a = np.array(buf, dtype=np.int64).reshape(nwg, 16).astype(np.float64) / 100.0
    t0 = a[:, 12].min()
    d = lambda i, j: (a[:, i] - a[:, j]).mean()
    print("%-16s %3d WGs | prologue %5.2f | tile 0: K loop %6.2f closing barrier + vmcnt(0) %5.2f epilogue %5.2f | last tile: K loop %6.2f wait %5.2f epilogue %5.2f | kernel %7.2f us"
          % (name, nwg, d(13, 12), d(1, 0), d(2, 1), d(3, 2), d(7, 6), d(8, 7), d(9, 8), (a[:, [3, 9]].max() - t0)), flush=True)
