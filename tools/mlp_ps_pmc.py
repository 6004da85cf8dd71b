"""Target of tools/pmc.sh: the two fused proj + MLP launches (8 x 16 rows vs producer / consumer) at the 105-image launch size, 12 launches each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semireward_amd import ops

DEV = "cuda:0"
D, Hd, N, B = 384, 1536, 257, int(os.environ.get("PS_IMAGES", "105"))
M = B * N
x = torch.randn(M, D, device=DEV)
ao = torch.randn(M, D, device=DEV).to(torch.bfloat16)
Wp = (torch.randn(D, D, device=DEV) * 0.05).to(torch.bfloat16)
bp = torch.randn(D, device=DEV) * 0.1
g, b = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV) * 0.1
W1 = (torch.randn(Hd, D, device=DEV) * 0.05).to(torch.bfloat16)
W2 = (torch.randn(D, Hd, device=DEV) * 0.02).to(torch.bfloat16)
b1, b2 = torch.randn(Hd, device=DEV) * 0.1, torch.randn(D, device=DEV) * 0.1
ln = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
pk = torch.empty(ops.mlp_ps_pack_bytes(D, Hd), dtype=torch.uint8, device=DEV)
ops.mlp_ps_pack(Wp, W1, W2, pk, D, Hd)
for _ in range(12):
    ops.mlp_fused_proj(x, ao, Wp, bp, None, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd, ln_next=ln, next_gamma=g, next_beta=b)
    x.normal_()
    ops.mlp_ps_proj(x, ao, pk, bp, None, g, b, 1e-6, b1, b2, None, 0, M, D, Hd, ln_next=ln, next_gamma=g, next_beta=b)
    x.normal_()
torch.cuda.synchronize()
