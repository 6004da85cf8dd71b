"""core/stepgraph.py: SRFlexMatch.train_step + ParamUpdateHook captured as HIP graphs (one per step variant) and replayed, against the same
steps launched eagerly from the same state: same masks, same FlexMatch table, same losses, parameters on the same trajectory."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from semireward_amd.algorithms import get_algorithm, srflexmatch as SF   # noqa: E402
from semireward_amd.core.stepgraph import StepGraph                      # noqa: E402
from semireward_amd.nets import vit                                       # noqa: E402
from semireward_amd.utils import synth                                    # noqa: E402

DEV = "cuda:0"
NSa = dict(algorithm="srflexmatch", num_classes=100, num_train_iter=204800, epoch=1, ema_m=0.0, ulb_loss_ratio=1.0, use_cat=True, amp=False,
           lr=5e-4, weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=5120, optim="AdamW", T=0.5, p_cutoff=0.95, hard_label=True,
           thresh_warmup=True, ulb_dest_len=50000, N_k=10, start_timing=20000, feature_dim=384, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99, gpu=0,
           rank=0, world_size=1, distributed=False)


def _make(graphed, it0, monkeypatch):
    monkeypatch.setattr(SF, "_DEFER_AUTOTUNE", False)
    alg = get_algorithm(argparse.Namespace(**NSa), vit.vit_small_patch2_32)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.model.seed = 4321
    alg.it = it0
    alg.optimizer.sched_step = it0
    alg.optimizer.step_count = 7                     # (bias corrections that are not 1)
    return alg, (StepGraph(alg, warm=1) if graphed else None)


def _one_step(alg, sg, batch):
    if sg is not None:
        out, log = sg.step(**batch)
    else:
        out, log = alg.train_step(**batch)
        alg.out_dict, alg.log_dict = out, log
        alg.hooks_dict["ParamUpdateHook"].after_train_step(alg)
    alg.it += 1
    torch.cuda.synchronize()
    h = alg.hooks_dict["MaskingHook"]
    return dict(loss=[float(log["train/" + k]) for k in ("sup_loss", "unsup_loss", "total_loss", "util_ratio")],
                sel=h.selected_label.clone(), acc=h.classwise_acc.clone(), flat=alg.model.flat.clone(), rew=alg.rewarder.flat.clone(),
                maxr=float(alg.max_reward), feat=out["feat"]["x_ulb_w"].clone())


def _run(graphed, it0, nsteps, monkeypatch):
    alg, sg = _make(graphed, it0, monkeypatch)
    batches = [alg.process_batch(**{k: torch.from_numpy(v) for k, v in synth.synth_batch(700 + i, 8, 8, 32, 100, 50000).items()}) for i in range(nsteps)]
    return alg, sg, [_one_step(alg, sg, batches[i]) for i in range(nsteps)]


def _copy_state(dst, src):
    """Everything a step reads, IN PLACE (a captured graph keeps the addresses): parameters and their derived operands, both optimizers, the
    rewarder / generator, the running maximum, the FlexMatch table."""
    dst.model.load_state_dict(src.model.state_dict())
    dst.optimizer.load_state_dict(src.optimizer.state_dict())
    dst.rewarder.load_state_dict(src.rewarder.state_dict()); dst.generator.load_state_dict(src.generator.state_dict())
    dst.rewarder_optimizer.load_state_dict(src.rewarder_optimizer.state_dict())
    dst.max_reward.copy_(src.max_reward)
    hs, hd = src.hooks_dict["MaskingHook"], dst.hooks_dict["MaskingHook"]
    hd.selected_label.copy_(hs.selected_label); hd.classwise_acc.copy_(hs.classwise_acc)
    if hasattr(hs, "hist"):
        hd.hist.copy_(hs.hist)
    dst.model._rng_calls = src.model._rng_calls
    torch.cuda.synchronize()


@pytest.mark.parametrize("regime", ["sr", "pre"])
def test_graph_replay_equals_eager_steps(regime, monkeypatch):
    """Every step starts from the SAME state in both engines (the eager one's, copied in place into the graphed one after each step): fp32
    atomics reorder the weight-gradient sums, so two free-running trajectories -- eager vs eager as much as eager vs replayed -- part ways
    after a handful of steps once a reward within round-off of the batch mean flips a 0/1 mask (tools/stepgraph_diag.py); from equal states
    the forward quantities of a step must agree exactly and the parameters to the round-off of one AdamW step."""
    it0, n = (30008, 16) if regime == "sr" else (1000, 6)        # sr: K = 8, SemiReward updates at it = 30010 and 30020; pre: K = 0, update every step
    a0, _ = _make(False, it0, monkeypatch)
    a1, sg = _make(True, it0, monkeypatch)
    batches = [a0.process_batch(**{k: torch.from_numpy(v) for k, v in synth.synth_batch(700 + i, 8, 8, 32, 100, 50000).items()}) for i in range(n)]
    p0 = a0.model.flat.clone()
    rew_changed = False
    for i in range(n):
        before = a0.model.flat.clone()
        rew_before = a0.rewarder.flat.clone()
        x, y = _one_step(a0, None, batches[i]), _one_step(a1, sg, batches[i])
        upd = float((x["flat"] - before).abs().max())
        assert torch.equal(x["feat"], y["feat"]), i                                                  # same state, same kernels: the forward agrees bit for bit
        np.testing.assert_allclose(y["loss"], x["loss"], rtol=1e-5, atol=1e-6, err_msg="step %d" % i)
        assert torch.equal(x["sel"], y["sel"]) and torch.equal(x["acc"], y["acc"]), i              # FlexMatch table / class accuracies
        # one AdamW step from equal moments: |difference| <= a couple of updates where a gradient within round-off of zero flips its sign
        assert float((x["flat"] - y["flat"]).abs().max()) <= 2.1 * upd, i
        assert float((x["flat"] - y["flat"]).abs().mean()) <= 1e-2 * upd, i
        assert float((x["rew"] - y["rew"]).abs().max()) <= 1e-5, i
        assert x["maxr"] == y["maxr"] or abs(x["maxr"] - y["maxr"]) <= 1e-6, i
        rew_changed = rew_changed or not torch.equal(x["rew"], rew_before)
        assert a1.optimizer.step_count == a0.optimizer.step_count and a1.optimizer.sched_step == a0.optimizer.sched_step
        assert a1.rewarder_optimizer.steps == a0.rewarder_optimizer.steps
        _copy_state(a1, a0)
    assert len(sg.graphs) == (2 if regime == "sr" else 1) and sg.replays >= n - 4, (len(sg.graphs), sg.replays, sg.eager_steps)
    assert rew_changed and not torch.equal(p0, a0.model.flat)        # the rewarder did get updated inside the sequence, the backbone moved


def test_two_graphed_runs_are_reproducible_in_their_forward(monkeypatch):
    """Same seed, same state, same batches: the DropPath draws of replayed steps follow the device-side seed (they differ from step to step and
    repeat from run to run)."""
    _, _, a = _run(True, 30001, 6, monkeypatch)
    _, _, b = _run(True, 30001, 6, monkeypatch)
    assert torch.equal(a[0]["feat"], b[0]["feat"])
    assert all(torch.equal(x["sel"], y["sel"]) for x, y in zip(a, b))
    assert not torch.equal(a[2]["feat"], a[3]["feat"])
