"""srhip_attn_masked_fwd / _bwd on the BERT-base (L = 512) and Wav2Vec2 (L = 199) shapes of the legs, with and without dropout on the probabilities:
how much of the kernel is the counter-based generator.  GPU box: python tools/attn_drop_probe.py"""
import sys, torch
sys.path.insert(0, ".")
from semireward_amd import ops
DEV = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for (B, N, H, name) in [(68, 512, 12, "bert read launch"), (16, 512, 12, "bert gradient rows"), (68, 199, 12, "wav2vec2 read launch"), (16, 199, 12, "wav2vec2 gradient rows")]:
    D = 64 * H
    qkv = (torch.randn(B * N, 3 * D, device=DEV) * 0.5).to(torch.bfloat16)
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B * H * N, dtype=torch.float32, device=DEV)
    klen = torch.full((B,), N, dtype=torch.int32, device=DEV)
    klen[::3] = N - 37
    d_out = torch.randn(B * N, D, device=DEV).to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(B * H * N, dtype=torch.float32, device=DEV)
    fl = 4.0 * B * H * N * N * 64 / 1e6
    for drop in (None, ops.Drop(1234567, 3, 0.1)):
        tf = timeit(lambda: ops.attn_masked_fwd(qkv, out, lse, klen, B, N, H, 0.125, drop=drop))
        tb = timeit(lambda: ops.attn_masked_bwd(qkv, out, d_out, lse, dqkv, ws, klen, B, N, H, 0.125, drop=drop))
        print("%-24s B=%3d N=%3d dropout %-4s | forward %7.1f us %6.0f TF/s | backward %7.1f us %6.0f TF/s" % (name, B, N, "off" if drop is None else "0.1", tf, fl / tf, tb, 2.5 * fl / tb), flush=True)
