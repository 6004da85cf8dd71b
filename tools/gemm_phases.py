"""Tuning build only (SRHIP_TUNING_BUILD=1): wall-clock phases of the 128x128-tile GEMM workgroups on the 4112-row products."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from semireward_amd import ops, _lib

dev = "cuda:0"
lib = _lib.lib()
lib.srhip_gemm_debug.argtypes = [ctypes.c_void_p, ctypes.c_int]
os.environ["SRHIP_GEMM"] = "tile"
for (M, N, K, epi, name) in [(4112, 1536, 384, ops.EPI_DGELU_BF16, "dfc2->dh (DGELU)"), (4112, 1536, 384, ops.EPI_GELU_BF16, "fc1 fwd (GELU)"),
                             (4112, 1536, 384, ops.EPI_BF16, "plain bf16"), (4112, 1152, 384, ops.EPI_BF16, "qkv fwd"),
                             (18761, 1152, 384, ops.EPI_BF16, "qkv 73 images"), (32639, 1152, 384, ops.EPI_BF16, "qkv 127 images"),
                             (32639, 384, 384, ops.EPI_RESID_F32, "proj 127 images")]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Bm = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    C = torch.zeros(M, N, dtype=torch.float32 if epi == ops.EPI_RESID_F32 else torch.bfloat16, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == ops.EPI_DGELU_BF16 else None
    for _ in range(3):
        ops.gemm_nt(epi, A, Bm, C, M, N, K, bias=bias, aux_in=aux)
    torch.cuda.synchronize()
    nwg = ((M + 127) // 128) * ((N + 127) // 128)
    buf = (ctypes.c_longlong * (4 * nwg))()
    assert lib.srhip_gemm_debug(buf, 4 * nwg) == 0
    a = np.array(buf, dtype=np.int64).reshape(nwg, 4)[:, :3].astype(np.float64) / 100.0
    t0 = a[:, 0].min()
    late = a[:, 0] - t0 > 1.0
    if late.any():
        print("   first-round WGs: K loop %5.1f epilogue %5.1f | later WGs: K loop %5.1f epilogue %5.1f" % (
            (a[~late, 1] - a[~late, 0]).mean(), (a[~late, 2] - a[~late, 1]).mean(), (a[late, 1] - a[late, 0]).mean(), (a[late, 2] - a[late, 1]).mean()))
    print("%-18s %4d WGs: start spread %5.1f | K loop %5.1f (max %5.1f) | epilogue %5.1f (max %5.1f) | last end %5.1f us" % (
        name, nwg, (a[:, 0] - t0).max(), (a[:, 1] - a[:, 0]).mean(), (a[:, 1] - a[:, 0]).max(), (a[:, 2] - a[:, 1]).mean(),
        (a[:, 2] - a[:, 1]).max(), (a[:, 2] - t0).max()))
