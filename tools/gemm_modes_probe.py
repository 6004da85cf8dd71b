"""srhip_gemm_nt on the D = 768 legs' shapes under one tile-dispatch mode (SRHIP_GEMM=tile|big256|big128|big2wg is read once per process):
    for m in default tile big256 big128 big2wg; do SRHIP_GEMM=$m python tools/gemm_modes_probe.py; done      (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, ".")
from semireward_amd import ops  # noqa: E402

DEV = "cuda:0"
if os.environ.get("SRHIP_GEMM") == "default":
    del os.environ["SRHIP_GEMM"]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


SHAPES = [(13952, 2304, 768, "bert qkv"), (13952, 768, 768, "bert proj"), (13952, 3072, 768, "bert fc1"), (13952, 768, 3072, "bert fc2"),
          (4096, 2304, 768, "bert grad qkv"), (4096, 3072, 768, "bert grad fc1"), (4096, 768, 3072, "bert grad fc2"),
          (5373, 2304, 768, "w2v qkv"), (5373, 3072, 768, "w2v fc1"), (5373, 768, 3072, "w2v fc2"), (8192, 8192, 8192, "8k")]
out = []
for (M, N, K, name) in SHAPES:
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    bias32 = torch.randn(N, device=DEV)
    t = timeit(lambda: ops.gemm_nt(ops.EPI_BF16, A, W, C, M, N, K, bias=bias32))
    out.append("%s %.0f" % (name.replace(" ", "_"), 2.0 * M * N * K / t / 1e6))
print("%-8s" % os.environ.get("SRHIP_GEMM", "default"), " | ".join(out), flush=True)
