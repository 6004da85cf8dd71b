"""Diagnostic for choosing the srflexmatch trace fixture (oracle/gen_golden.py --search): runs the HIP engine over a candidate fixture and reports,
per iteration, the largest |engine max-prob - reference max-prob| and the smallest slack = (distance of the reference max-prob from the nearer of
its two thresholds) - (engine deviation of that element).  slack > 0 for every element <=> every mask / selection decision is the reference's.

    python tools/trace_diag.py <fixture.npz> '<json of the TRACE dict>' [...]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import semireward_ref as S        # noqa: E402
from oracle import vit_ref as V               # noqa: E402
from oracle.gen_golden import trace_vit_params   # noqa: E402
from semireward_amd.algorithms import get_algorithm   # noqa: E402
from semireward_amd.nets import vit           # noqa: E402
from semireward_amd.utils import synth        # noqa: E402


def run(path, tr):
    g = np.load(path)
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_TINY_TEST)
    d = dict(algorithm="srflexmatch", num_classes=C, num_train_iter=2000, epoch=1, ema_m=0.0, ulb_loss_ratio=1.0, use_cat=True, amp=False,
             lr=tr.get("lr", 5e-4), weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=50, optim="AdamW", T=0.5, p_cutoff=tr["p_cutoff"],
             hard_label=True, thresh_warmup=True, ulb_dest_len=tr["ulb_dest_len"], N_k=10, start_timing=100, feature_dim=128, sr_lr=5e-4,
             sr_ema=False, sr_ema_m=0.99, gpu=0, rank=0, world_size=1, distributed=False)
    alg = get_algorithm(argparse.Namespace(**d), vit.vit_tiny_test)
    T = lambda x: {k: torch.from_numpy(v) for k, v in x.items()}   # noqa: E731
    alg.model.load_state_dict(T(trace_vit_params(cfg, seed, tr.get("head_gain", 1.0), tr.get("hot_classes", 0), tr.get("cold_scale", 0.25))))
    alg.rewarder.load_state_dict(T(synth.synth_params(S.rewarder_shapes(cfg.embed_dim, C), seed + 1)))
    alg.generator.load_state_dict(T(synth.synth_params(S.generator_shapes(cfg.embed_dim), seed + 2)))
    rows, flips = [], 0
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        alg.it = it
        alg.optimizer.sched_step = it
        K = int(g[f"{p}/K"])
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
        alg.inject_droppath = [torch.from_numpy(synth.synth_droppath(seed + 1000 * (n + 1) + k, V.drop_path_probs(cfg), Bl + 2 * Bu)) for k in range(K + 1)]
        alg.trace = {}
        out, log = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))
        alg.out_dict, alg.log_dict = out, log
        alg.call_hook("after_train_step")
        want = g[f"{p}/masks"]
        masks = np.stack([m.cpu().numpy() for m in alg.trace["masks"]])
        mpv = alg.trace["max_probs"].cpu().numpy().reshape(want.shape)
        ref = g[f"{p}/mask_probs"]
        dev = np.abs(mpv - ref)
        margin = np.minimum(np.abs(ref - g[f"{p}/mask_thr"]), np.abs(ref - tr["p_cutoff"]))
        flips += int((masks != want).sum())
        rows.append((it, float(dev.max()), float(margin.min()), float((margin - dev).min())))
    return rows, flips


if __name__ == "__main__":
    a = sys.argv[1:]
    for path, js in zip(a[0::2], a[1::2]):
        tr = json.loads(js)
        rows, flips = run(path, tr)
        print(os.path.basename(path), "flips", flips, "min slack %.4f" % min(r[3] for r in rows), "worst dev %.4f" % max(r[1] for r in rows))
        for r in rows:
            print("   it %4d  dev %.4f  margin %.4f  slack %.4f" % r)
