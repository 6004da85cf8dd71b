"""srhip_gemm_nt against the vendor library (torch.nn.functional.linear = hipBLASLt / rocBLAS) on the GEMM shapes of the legs -- a yardstick for
the hand-written kernels, not a code path: the product never calls the library.  GPU box: python tools/lib_gemm_probe.py"""
import torch, sys
sys.path.insert(0, ".")
from semireward_amd import ops
DEV = "cuda:0"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for (M, N, K, name) in [(13952, 2304, 768, "bert qkv"), (13952, 768, 768, "bert proj"), (13952, 3072, 768, "bert fc1"), (13952, 768, 3072, "bert fc2"),
                        (5373, 3072, 768, "w2v fc1"), (5373, 768, 3072, "w2v fc2"), (51400, 1152, 384, "vit qkv"), (4112, 1536, 384, "vit grad fc1"),
                        (8192, 8192, 8192, "8k")]:
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=DEV).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    bias32 = b.float()
    t_lib = timeit(lambda: torch.nn.functional.linear(A, W, b))
    t_lib_nb = timeit(lambda: torch.matmul(A, W.t()))
    t_mine = timeit(lambda: ops.gemm_nt(ops.EPI_BF16, A, W, C, M, N, K, bias=bias32))
    fl = 2.0 * M * N * K
    print("%-12s M=%6d N=%5d K=%5d | library linear+bias %7.1f us %7.1f TF/s | matmul %7.1f us %7.1f TF/s | srhip_gemm_nt %7.1f us %7.1f TF/s"
          % (name, M, N, K, t_lib, fl / t_lib / 1e6, t_lib_nb, fl / t_lib_nb / 1e6, t_mine, fl / t_mine / 1e6), flush=True)
