"""Build-container cross-check promised in BASELINE.md 2: the REFERENCE's own SRFlexMatch.train_step + backward + AdamW step timed on the host
cores next to the oracle port that bench.py's cpu_baseline times (same batch 8/8/8, ViT-S/2, fp32, K = 8 regime).  Needs /root/reference:
runs in the build container only (python tools/ref_cpu_step.py [steps]); prints one JSON line.  Test infrastructure, not shipped."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden as G          # noqa: E402
from oracle import vit_ref as V             # noqa: E402
from semireward_amd.utils import synth      # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.manual_seed(0)
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    model = G.build_ref_vit(V.VIT_SMALL_P2_32, 100, synth.synth_params(V.param_shapes(cfg), 0))
    model.train()
    tr = dict(G.TRACE, C=100, num_train_iter=204800, start_timing=20000, N_k=10, ulb_dest_len=50000, p_cutoff=0.95)
    alg = G.build_headless_srflexmatch(model, 100, 384, dict(tr, algorithm="srflexmatch"))
    bu = G.R.mod("semilearn.core.utils.build")
    alg.optimizer = bu.get_optimizer(model, "AdamW", 5e-4, 0.9, 5e-4, 0.5)
    alg.scheduler = bu.get_cosine_schedule_with_warmup(alg.optimizer, 204800, num_warmup_steps=5120)
    b = synth.synth_batch(0, 8, 8, 32, 100, 50000)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))   # noqa: E731
    ts = []
    for i in range(steps + 1):
        alg.it = 30001 + i                                    # sr_decay() = 8, no rewarder update (it % N_k != 0) except every 10th
        t0 = time.perf_counter()
        o, log = alg.train_step(T(b["x_lb"]), T(b["y_lb"]), T(b["idx_ulb"]), T(b["x_ulb_w"]), T(b["x_ulb_s"]))
        o["loss"].backward()
        alg.optimizer.step(); alg.scheduler.step(); model.zero_grad()
        ts.append(time.perf_counter() - t0)
    ts = ts[1:]                                               # first step warms the allocator / thread pool
    med = sorted(ts)[len(ts) // 2]
    print(json.dumps({"what": "reference SRFlexMatch.train_step + backward + AdamW, ViT-S/2 fp32, 8/8/8, K=8", "cores": torch.get_num_threads(),
                      "s_per_step": ts, "median_s": med, "unlabeled_images_per_s": 8 / med}))


if __name__ == "__main__":
    main()
