"""Weight-gradient launches, 128 x 128 kernel vs the 256 x 256 persistent kernel, on the shapes of the legs (GPU box):
    python tools/gemm_tn_pp_bench.py            -> one line per (table, kernel): us per launch, TFLOP/s, max rel. difference between the two"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semireward_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def table(D, I, K, layers):
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(torch.bfloat16).to(DEV)   # noqa: E731
    dys = dict(w2=mk(K, D), w1=mk(K, I), o=mk(K, D), qkv=mk(K, 3 * D))
    xs = dict(w2=mk(K, I), w1=mk(K, D), o=mk(K, D), qkv=mk(K, D))
    probs = []
    for _ in range(layers):
        for k, (M, N) in dict(w2=(D, I), w1=(I, D), o=(D, D), qkv=(3 * D, D)).items():
            probs.append((dys[k], xs[k], torch.zeros(M, N, device=DEV), torch.zeros(M, device=DEV), M, N, K))
    return probs


def run(name, probs, pp, reps=20):
    desc, npb, nt, flops, _ = ops.make_group_tn_desc(probs, DEV, tile=256 if pp else 128)
    for p in probs:
        p[2].zero_(); p[3].zero_()
    ops.gemm_tn_grouped_f32(desc, npb, nt, 1.0, 1.0, pp=pp)
    torch.cuda.synchronize()
    out = [p[2].clone() for p in probs[:4]] + [p[3].clone() for p in probs[:4]]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm_tn_grouped_f32(desc, npb, nt, 1.0, 1.0, pp=pp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%-28s %-8s entries %3d tiles %5d  %8.1f us  %7.1f TFLOP/s" % (name, "pp256" if pp else "tile128", npb, nt, us, flops / us * 1e-6), flush=True)
    return out


def main():
    for name, D, I, K, layers in (("bert layer K=8192", 768, 3072, 8192, 1), ("bert 12 layers K=8192", 768, 3072, 8192, 12),
                                  ("w2v 12 layers K=3184", 768, 3072, 3184, 12), ("vit-s 12 blocks K=4112", 384, 1536, 4112, 12),
                                  ("vit-s 4 blocks K=4112", 384, 1536, 4112, 4)):
        probs = table(D, I, K, layers)
        a = run(name, probs, False)
        b = run(name, probs, True)
        err = max(float((x - y).norm() / x.norm()) for x, y in zip(a, b))
        print("    max rel. difference tile128 vs pp256: %.2e" % err, flush=True)


if __name__ == "__main__":
    main()
