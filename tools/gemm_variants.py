"""qkv / proj products of the two inference launches of a step on every GEMM tiling (SRHIP_GEMM switch), HIP-event medians."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from semireward_amd import ops
    from tools.microbench import timeit
    DEV = "cuda:0"
    out = []
    for (M, N, K, epi, name) in [(18761, 1152, 384, ops.EPI_BF16, "qkv73"), (32639, 1152, 384, ops.EPI_BF16, "qkv127"),
                                 (18761, 384, 384, ops.EPI_RESID_F32, "proj73"), (32639, 384, 384, ops.EPI_RESID_F32, "proj127"),
                                 (51400, 1152, 384, ops.EPI_BF16, "qkv200")]:
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        Bm = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        C = torch.zeros(M, N, dtype=torch.float32 if epi == ops.EPI_RESID_F32 else torch.bfloat16, device=DEV)
        out.append("%s %6.1f" % (name, timeit(lambda: ops.gemm_nt(epi, A, Bm, C, M, N, K, bias=bias), reps=30)))
    print("%-8s" % sys.argv[1], " | ".join(out), flush=True)
else:
    for mode in ("tile", "big256", "big128", "big2wg", "default"):
        env = dict(os.environ)
        if mode != "default":
            env["SRHIP_GEMM"] = mode
        subprocess.run([sys.executable, __file__, mode], env=env)
