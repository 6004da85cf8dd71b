"""srhip_gemm_nt under each tile kernel (SRHIP_GEMM test hook, one child process per mode) on a grid of (M, N, K): which kernel should the plan pick.
GPU box:  python tools/gemm_dispatch_probe.py"""
import json, os, subprocess, sys
sys.path.insert(0, ".")
SHAPES = [(M, N, K) for M in (2048, 4096, 8192, 12288, 16384, 30248) for (N, K) in ((768, 768), (768, 3072), (3072, 768), (2304, 768))] + \
         [(4112, 384, 1536), (4112, 1536, 384), (4112, 1152, 384), (26985, 1152, 384), (24415, 1152, 384)]
if "--child" in sys.argv:
    import torch
    from semireward_amd import ops
    DEV = "cuda:0"
    out = {}
    for (M, N, K) in SHAPES:
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        C = torch.empty(M, N, dtype=torch.float32, device=DEV)
        fn = lambda: ops.gemm_nt(ops.EPI_RESID_F32, A, W, C, M, N, K, bias=bias)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        out["%d,%d,%d" % (M, N, K)] = e0.elapsed_time(e1) * 1000 / 20
    print(json.dumps(out))
    sys.exit(0)
res = {}
for mode in ("default", "tile", "small", "big256"):
    env = dict(os.environ)
    env.pop("SRHIP_GEMM", None)
    if mode != "default": env["SRHIP_GEMM"] = mode
    res[mode] = json.loads(subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
for (M, N, K) in SHAPES:
    k = "%d,%d,%d" % (M, N, K)
    fl = 2.0 * M * N * K / 1e6
    best = min(res, key=lambda m: res[m][k] if m != "default" else 1e9)
    print("%-18s (residual epilogue) %s | best %s" % (k, " | ".join("%s %6.1f us %5.0f TF/s" % (m, res[m][k], fl / res[m][k]) for m in res), best), flush=True)
