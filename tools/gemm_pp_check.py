"""The two-wave-group 256 x 256 x 64 GEMM (gemm_pp_kernel) forced onto a list of shapes (SRHIP_GEMM=big256): every epilogue against an fp64 product,
bitwise repeatability over many launches (a missed LDS-DMA wait shows as a result that changes between launches), and its time beside the
lockstep persistent kernel (SRHIP_GEMM=bigold, child process) and the vendor library.  GPU box:  python tools/gemm_pp_check.py [--time-only]"""
import math, os, subprocess, sys, json
sys.path.insert(0, ".")
MODE = os.environ.get("SRHIP_GEMM")
if MODE is None:
    os.environ["SRHIP_GEMM"] = MODE = "big256"
elif MODE == "default":
    del os.environ["SRHIP_GEMM"]
import torch
from semireward_amd import ops
DEV = "cuda:0"
SHAPES = [(13952, 2304, 768), (13952, 768, 768), (13952, 3072, 768), (13952, 768, 3072), (5373, 3072, 768), (5373, 768, 3072), (8192, 8192, 8192),
          (51400, 1152, 384), (70001, 1152, 384), (1030, 516, 128), (256, 256, 128), (700, 1000, 256), (4112, 1536, 384), (77824, 768, 3072)]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def times():
    out = {}
    for (M, N, K) in SHAPES:
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        out["%d,%d,%d" % (M, N, K)] = timeit(lambda: ops.gemm_nt(ops.EPI_BF16, A, W, C, M, N, K, bias=bias))
    return out


if "--child" in sys.argv:
    print(json.dumps(times()))
    sys.exit(0)

if "--time-only" not in sys.argv:
    for (M, N, K) in SHAPES:
        if M * N > 4.0e8: continue
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
        W = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        ref = (A.double() @ W.double().t()) + bias.double()
        C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm_nt(ops.EPI_BF16, A, W, C, M, N, K, bias=bias)
        e0 = relerr(C, ref)
        C0 = C.clone()
        changed = 0
        for _ in range(30):
            C.zero_()
            ops.gemm_nt(ops.EPI_BF16, A, W, C, M, N, K, bias=bias)
            changed += int(not torch.equal(C, C0))
        pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm_nt(ops.EPI_GELU_BF16, A, W, C, M, N, K, bias=bias, aux_out=pre, ldaux=N)
        e1 = max(relerr(C, gelu(ref)), relerr(pre, ref))
        X0 = torch.randn(M, N, generator=g).to(DEV)
        X = X0.clone()
        ops.gemm_nt(ops.EPI_RESID_F32, A, W, X, M, N, K, bias=bias)
        e2 = relerr(X, X0.double() + ref)
        pin = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
        ops.gemm_nt(ops.EPI_DGELU_BF16, A, W, C, M, N, K, aux_in=pin, ldaux=N)
        p = pin.double().requires_grad_(True)
        gelu(p).sum().backward()
        e3 = relerr(C, (ref - bias.double()) * p.grad)
        ok = e0 < 4e-3 and e1 < 5e-3 and e2 < 2e-3 and e3 < 5e-3 and changed == 0
        print("%-22s bf16 %.1e gelu %.1e resid %.1e dgelu %.1e  launches that differ from the first: %d / 30  %s"
              % ("%d x %d x %d" % (M, N, K), e0, e1, e2, e3, changed, "ok" if ok else "FAIL"), flush=True)

mine = times()
def child(mode):
    env = dict(os.environ, SRHIP_GEMM=mode)
    return json.loads(subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
old, r8, dflt = child("bigoldf"), child("big256r8"), child("default")
for (M, N, K) in SHAPES:
    k = "%d,%d,%d" % (M, N, K)
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=DEV).to(torch.bfloat16)
    t_lib = timeit(lambda: torch.nn.functional.linear(A, W, b))
    fl = 2.0 * M * N * K / 1e6
    print("%-18s two-group ring 10 %7.1f us %6.0f TF/s | ring 8 %7.1f us %6.0f | lockstep %7.1f us %6.0f | default dispatch %7.1f us %6.0f | library %7.1f us %6.0f"
          % (k, mine[k], fl / mine[k], r8[k], fl / r8[k], old[k], fl / old[k], dflt[k], fl / dflt[k], t_lib, fl / t_lib), flush=True)
