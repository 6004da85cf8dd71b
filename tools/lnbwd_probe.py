"""Times srhip_layernorm_bwd_cast on the gradient rows of a step (M = 4112, D = 384) with and without the gamma / beta atomics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semireward_amd import ops
from tools.microbench import timeit

M, D, N = 4112, 384, 257
dev = "cuda"
dy = torch.randn(M, D, device=dev).bfloat16()
x = torch.randn(M, D, device=dev)
mean, rstd = x.mean(1), 1.0 / x.std(1)
gamma = torch.ones(D, device=dev)
dx = torch.zeros(M, D, device=dev)
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
out = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
rs = torch.ones(16, device=dev)
print("with atomics  %.1f us" % (timeit(lambda: ops.layernorm_bwd_cast(dy, x, mean, rstd, gamma, dx, dg, db, out, rs, N, M, D), reps=200)))
print("no atomics    %.1f us" % (timeit(lambda: ops.layernorm_bwd_cast(dy, x, mean, rstd, gamma, dx, None, None, out, rs, N, M, D), reps=200)))
part = torch.zeros(16, 2, D, device=dev)
for R in (4, 8, 16):
    print("replicas %2d   %.1f us" % (R, timeit(lambda: ops.layernorm_bwd_part(dy, x, mean, rstd, gamma, dx, part, R, out, rs, N, M, D), reps=200)))
