#!/usr/bin/env python
"""End-to-end on one MI355X: uint8 "dataset" resident in HBM -> device-side weak / strong views (data/augment.py) -> SRFlexMatch steps
(ViT-S/2) with evaluation, the way config/SemiReward/usb_cv/flexmatch/flexmatch_cifar100_200_0.yaml wires the reference.  Data are synthetic
class-dependent blobs (no network for CIFAR); the point is the plumbing: dataset arrays, per-step views, train loop, evaluate().

    python examples/train_synthetic_cifar.py --steps 60
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semireward_amd.algorithms import get_algorithm          # noqa: E402
from semireward_amd.data.augment import GpuAugment           # noqa: E402
from semireward_amd.nets import vit                          # noqa: E402

MEAN, STD = (0.507, 0.487, 0.441), (0.267, 0.256, 0.276)     # cifar100 statistics (cv_datasets/cifar.py:16-17)


def synth_dataset(n, num_classes, seed):
    """uint8 [n, 32, 32, 3] images whose colour / stripe pattern depends on the class, + labels."""
    rng = np.random.Generator(np.random.PCG64(seed))
    y = rng.integers(0, num_classes, size=n)
    base = np.random.Generator(np.random.PCG64(12345)).integers(40, 216, size=(num_classes, 1, 1, 3))     # class prototypes: same for every split
    freq = 1 + np.arange(num_classes) % 7
    xs = np.arange(32)[None, None, :, None]
    img = base[y] + 35 * np.sin(xs * freq[y][:, None, None, None] * 0.4) + rng.normal(0, 18, size=(n, 32, 32, 3))
    return np.clip(img, 0, 255).astype(np.uint8), y.astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    dev = "cuda:0"
    C, B = a.classes, a.batch
    args = argparse.Namespace(
        algorithm="srflexmatch", num_classes=C, num_train_iter=max(a.steps, 40), epoch=1, ema_m=0.0, ulb_loss_ratio=1.0, use_cat=True, amp=False,
        lr=5e-4, weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=5, optim="AdamW", T=0.5, p_cutoff=0.95, hard_label=True, thresh_warmup=True,
        ulb_dest_len=2048, N_k=10, start_timing=20, feature_dim=384, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99, gpu=0, rank=0, world_size=1,
        distributed=False)
    alg = get_algorithm(args, vit.vit_small_patch2_32)
    x_lb, y_lb = synth_dataset(40 * C // 10, C, 1)                 # "40 labels"-style split
    x_ulb, _ = synth_dataset(2048, C, 2)
    x_te, y_te = synth_dataset(512, C, 3)
    lb, ulb = torch.from_numpy(x_lb).to(dev), torch.from_numpy(x_ulb).to(dev)     # the dataset lives in HBM as uint8
    aug = GpuAugment(32, 4, MEAN, STD, n_ops=3, device=dev, seed=0)               # crop_ratio 0.875 -> padding 4
    rng = np.random.Generator(np.random.PCG64(0))

    def batches():
        while True:
            il, iu = rng.integers(0, len(x_lb), size=B), rng.permutation(len(x_ulb))[:B]
            yield {"x_lb": aug(lb, False, src_index=il), "y_lb": torch.from_numpy(y_lb[il]).to(dev), "idx_ulb": torch.from_numpy(iu).to(dev),
                   "x_ulb_w": aug(ulb, False, src_index=iu), "x_ulb_s": aug(ulb, True, src_index=iu)}

    def eval_loader():
        te = torch.from_numpy(x_te).to(dev)
        for s in range(0, len(x_te), 128):
            idx = np.arange(s, min(s + 128, len(x_te)))
            d = dict(i=np.full(len(idx), 0), j=np.full(len(idx), 0), flip=np.zeros(len(idx), bool))       # transform_val: no crop shift / flip
            yield {"x_lb": GpuAugment(32, 0, MEAN, STD, device=dev)(te, False, draws=d, src_index=idx), "y_lb": torch.from_numpy(y_te[idx])}

    it = batches()
    for step in range(a.steps):
        alg.it = step
        alg.optimizer.sched_step = step
        alg.out_dict, alg.log_dict = alg.train_step(**alg.process_batch(**next(it)))
        alg.call_hook("after_train_step")
        if step % 10 == 0 or step == a.steps - 1:
            print("it %3d  sup %.3f  unsup %.3f  util %.2f" % (step, float(alg.log_dict["train/sup_loss"]), float(alg.log_dict["train/unsup_loss"]),
                                                              float(alg.log_dict["train/util_ratio"])), flush=True)
    ev = alg.evaluate(loader=list(eval_loader()))
    print("eval:", {k: round(float(v), 4) for k, v in ev.items()})
    return ev


if __name__ == "__main__":
    main()
