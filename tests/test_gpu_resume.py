"""Resume == the run that never stopped (semilearn/core/algorithmbase.py:459-496 save_model / load_model, srflexmatch.py:219-231 and the other
algorithms' hook state): train k steps across ``start_timing`` and an ``N_k`` boundary, save, load into a FRESH algorithm, continue -- against an
identical algorithm that simply keeps going.
  * after load_model EVERY piece of state equals the saved run's bit for bit: backbone parameters (+ BatchNorm buffers, EMA shadow), optimizer
    moments / step counters / schedule position, rewarder + generator parameters, the rewarder's Adam moments and step count, max_reward, the
    thresholding hook's state, the DropPath draw counter;
  * the next 3 steps: every mask and pseudo label of every pass identical, hook state identical, parameters / rewarder / Adam moments within 1e-6
    (bit-equal whenever the backward's atomics happen to fall in the same order: reported, not required -- the engine is not run-to-run
    deterministic at 1e-9, with or without a checkpoint in between)."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vit_ref as V               # noqa: E402
from oracle import wrn_ref as W               # noqa: E402
from semireward_amd.algorithms import get_algorithm   # noqa: E402
from semireward_amd.nets import vit, wrn      # noqa: E402
from semireward_amd.utils import synth        # noqa: E402

BASE = dict(num_classes=10, num_train_iter=40, epoch=1, ema_m=0.0, ulb_loss_ratio=1.0, use_cat=True, amp=False, lr=5e-4, weight_decay=5e-4,
            layer_decay=0.5, num_warmup_iter=2, optim="AdamW", T=0.5, p_cutoff=0.3, hard_label=True, thresh_warmup=True, ulb_dest_len=64, N_k=2,
            start_timing=3, feature_dim=128, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99, gpu=0, rank=0, world_size=1, distributed=False)
CASES = {
    "srflexmatch": (dict(algorithm="srflexmatch"), "vit"),
    "srsoftmatch": (dict(algorithm="srsoftmatch", dist_align=True, dist_uniform=True, ema_p=0.9, n_sigma=2, per_class=False), "vit"),
    "srfreematch": (dict(algorithm="srfreematch", ema_p=0.9, use_quantile=True, clip_thresh=False, ent_loss_ratio=0.01), "vit"),
    "srpseudolabel_wrn": (dict(algorithm="srpseudolabel", optim="SGD", lr=0.03, momentum=0.9, weight_decay=1e-3, layer_decay=1.0, ema_m=0.999,
                               unsup_warm_up=0.4, p_cutoff=0.12), "wrn"),
}


def build(name):
    extra, net = CASES[name]
    d = dict(BASE)
    d.update(extra)
    if net == "wrn":
        d["feature_dim"] = W.channels(W.WrnCfg(num_classes=10, **W.WRN_TINY_TEST))[3]
    alg = get_algorithm(argparse.Namespace(**d), vit.vit_tiny_test if net == "vit" else wrn.wrn_tiny_test)
    if net == "vit":
        alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(V.param_shapes(V.VitCfg(num_classes=10, **V.VIT_TINY_TEST)), 3).items()})
    alg.model.seed = 4242
    alg.model.train()
    return alg


def batch_of(alg, name, i):
    img = 8 if CASES[name][1] == "vit" else 16
    b = {k: torch.from_numpy(v) for k, v in synth.synth_batch(700 + i, 4, 4, img, 10, 64).items()}
    import inspect
    keep = set(inspect.signature(alg.train_step).parameters)
    return alg.process_batch(**{k: v for k, v in b.items() if k in keep})


def step(alg, name, i, masks_out=None):
    alg.trace = {} if masks_out is not None else None
    alg.out_dict, alg.log_dict = alg.train_step(**batch_of(alg, name, i))
    if masks_out is not None:
        torch.cuda.synchronize()
        masks_out.append(([m.clone() for m in alg.trace["masks"]], alg.trace["pseudo"].clone() if "pseudo" in alg.trace else None))
    alg.trace = None
    alg.call_hook("after_train_step")
    alg.it += 1


def state_of(alg):
    """Every tensor / counter a continuation depends on, by name."""
    st = {"model.flat": alg.model.flat, "rewarder.flat": alg.rewarder.flat, "generator.flat": alg.generator.flat,
          "rewarder.adam.m": alg.rewarder_optimizer.m, "rewarder.adam.v": alg.rewarder_optimizer.v,
          "rewarder.adam.steps": torch.tensor(alg.rewarder_optimizer.steps), "max_reward": alg.max_reward.reshape(1),
          "it": torch.tensor(alg.it), "sched_step": torch.tensor(alg.optimizer.sched_step), "opt.step_count": torch.tensor(alg.optimizer.step_count),
          "rng.draws": torch.tensor(getattr(alg.model, "_rng_calls", 0)), "rng.seed": torch.tensor(getattr(alg.model, "seed", 0))}
    for k, v in alg.optimizer.state_dict().items():
        if torch.is_tensor(v):
            st["opt." + k] = v
    for k, v in getattr(alg.model, "buffers", {}).items() if isinstance(getattr(alg.model, "buffers", None), dict) else []:
        st["buffer." + k] = v
    if alg.ema_model is not alg.model:
        st["ema.flat"] = alg.ema_model.flat
        for k, v in getattr(alg.ema_model, "buffers", {}).items():
            st["ema.buffer." + k] = v
    for hn, h in alg.hooks_dict.items():
        for k, v in vars(h).items():
            if torch.is_tensor(v) and not k.startswith("_"):
                st["hook.%s.%s" % (hn, k)] = v
        if hasattr(h, "selected_label"):
            st["hook.%s.selected_label" % hn] = h.selected_label
    return {k: v.detach().clone().cpu() for k, v in st.items()}


@pytest.mark.parametrize("name", list(CASES))
def test_resume_equals_the_uninterrupted_run(tmp_path, name):
    a = build(name)
    for i in range(6):                        # it = 0 .. 5: the stage-1 updates, start_timing = 3 (K = sr_decay() = 11 / 9 passes from it = 4), the N_k boundary at it = 4
        step(a, name, i)
    torch.cuda.synchronize()
    a.it -= 1                                 # save_model runs inside the loop, before ``it`` advances (get_save_dict stores it + 1)
    a.save_model("latest_model.pth", str(tmp_path))
    a.it += 1
    saved = state_of(a)
    assert a.rewarder_optimizer.steps >= 3 and float(saved["max_reward"]) > -float("inf")
    b = build(name)
    b.load_model(str(tmp_path / "latest_model.pth"))
    loaded = state_of(b)
    assert set(saved) == set(loaded)
    for k in saved:
        assert torch.equal(saved[k], loaded[k]), k
    ma, mb = [], []
    for i in range(6, 9):                     # it = 6, 7, 8: K = 8 / 7 ... passes, the N_k boundaries at it = 6 and 8
        step(a, name, i, ma)
        step(b, name, i, mb)
    torch.cuda.synchronize()
    for (m1, p1), (m2, p2) in zip(ma, mb):
        assert len(m1) == len(m2) >= 2
        assert all(torch.equal(x, y) for x, y in zip(m1, m2))
        assert p1 is None or torch.equal(p1, p2)
    sa, sb = state_of(a), state_of(b)
    exact = True
    for k in sa:
        if sa[k].dtype in (torch.int64, torch.int32, torch.bool) or k.startswith("hook."):
            if k.startswith("hook.") and sa[k].is_floating_point():
                assert torch.allclose(sa[k], sb[k], rtol=1e-5, atol=1e-7), k
            else:
                assert torch.equal(sa[k], sb[k]), k
            continue
        if not bool(torch.isfinite(sa[k]).all()):          # (max_reward right after the reset of an N_k step: -inf on both sides)
            assert torch.equal(sa[k], sb[k]), k
            continue
        num, den = float((sa[k].double() - sb[k].double()).norm()), float(sa[k].double().norm()) + 1e-30
        assert num / den <= 1e-6, (k, num / den)
        exact = exact and bool(torch.equal(sa[k], sb[k]))
    print("resume[%s]: continuation bit-equal: %s" % (name, exact))
