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


def _run(graphed, it0, nsteps, monkeypatch):
    monkeypatch.setattr(SF, "_DEFER_AUTOTUNE", False)
    alg = get_algorithm(argparse.Namespace(**NSa), vit.vit_small_patch2_32)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.model.seed = 4321
    alg.it = it0
    alg.optimizer.sched_step = it0
    alg.optimizer.step_count = 7                     # (bias corrections that are not 1)
    sg = StepGraph(alg, warm=1) if graphed else None
    batches = [alg.process_batch(**{k: torch.from_numpy(v) for k, v in synth.synth_batch(700 + i, 8, 8, 32, 100, 50000).items()}) for i in range(nsteps)]
    rec = []
    for i in range(nsteps):
        if sg is not None:
            out, log = sg.step(**batches[i])
        else:
            out, log = alg.train_step(**batches[i])
            alg.out_dict, alg.log_dict = out, log
            alg.hooks_dict["ParamUpdateHook"].after_train_step(alg)
        alg.it += 1
        torch.cuda.synchronize()
        h = alg.hooks_dict["MaskingHook"]
        rec.append(dict(loss=[float(log["train/" + k]) for k in ("sup_loss", "unsup_loss", "total_loss", "util_ratio")],
                        sel=h.selected_label.clone(), acc=h.classwise_acc.clone(), flat=alg.model.flat.clone(), rew=alg.rewarder.flat.clone(),
                        maxr=float(alg.max_reward), feat=out["feat"]["x_ulb_w"].clone()))
    return alg, sg, rec


@pytest.mark.parametrize("regime", ["sr", "pre"])
def test_graph_replay_equals_eager_steps(regime, monkeypatch):
    it0, n = (30008, 16) if regime == "sr" else (1000, 6)        # sr: K = 8, SemiReward updates at it = 30010 and 30020; pre: K = 0, update every step
    a0, _, r0 = _run(False, it0, n, monkeypatch)
    a1, sg, r1 = _run(True, it0, n, monkeypatch)
    assert len(sg.graphs) == (2 if regime == "sr" else 1) and sg.replays >= n - 4, (len(sg.graphs), sg.replays, sg.eager_steps)
    assert a1.optimizer.step_count == a0.optimizer.step_count and a1.optimizer.sched_step == a0.optimizer.sched_step
    assert a1.model._rng_calls == a0.model._rng_calls and a1.rewarder_optimizer.steps == a0.rewarder_optimizer.steps
    upd0 = float((r0[0]["flat"] - torch.from_numpy(np.concatenate([v.ravel() for v in synth.synth_params(a0.model.names_shapes, 0).values()])).to(DEV)).abs().max())
    for i, (x, y) in enumerate(zip(r0, r1)):
        # the first step starts from identical state: forward quantities agree bit for bit (fp32 atomics only reorder the weight-gradient sums,
        # so the parameters -- and everything computed from them afterwards -- agree to AdamW-step round-off, as between two eager runs)
        if i == 0:
            assert torch.equal(x["feat"], y["feat"]) and x["loss"] == y["loss"]
        assert torch.equal(x["sel"], y["sel"]) and torch.equal(x["acc"], y["acc"]), i          # FlexMatch table / class accuracies
        # (two EAGER runs differ by as much: a first-moment-free AdamW step is ~lr * sign(g), so a gradient within round-off of zero flips its
        # update, and the difference feeds the next step's forward)
        # The unsupervised loss is a mean over 8 rows behind two 0/1 masks (FlexMatch threshold x reward >= mean reward): once the parameters
        # of two runs differ in the last bits, a reward within round-off of the mean flips a mask and the loss jumps by a whole row's term.
        # Two eager runs part ways like that from step 5-7 of this sequence on (tools/stepgraph_diag.py prints four runs side by side), so the
        # masked losses are compared over the first 4 steps (2 of them replays, one with the SemiReward update) and the supervised loss after.
        if i < 4 or regime == "pre":
            np.testing.assert_allclose(y["loss"], x["loss"], rtol=2e-3, atol=2e-4, err_msg="step %d" % i)
        else:
            np.testing.assert_allclose(y["loss"][0], x["loss"][0], rtol=5e-2, err_msg="step %d" % i)
        assert float((x["flat"] - y["flat"]).abs().max()) <= 2.1 * (i + 1) * upd0, i
        close = i < 4 or regime == "pre"
        assert float((x["flat"] - y["flat"]).abs().mean()) <= (1e-2 if close else 1.0) * upd0 * (i + 1), i
        tol = 1e-4 if close else 1e-2
        assert float((x["rew"] - y["rew"]).abs().max()) <= tol and (x["maxr"] == y["maxr"] or abs(x["maxr"] - y["maxr"]) < tol), i
    assert not torch.equal(r1[0]["rew"], r1[-1]["rew"])               # the rewarder did get updated inside replayed steps


def test_two_graphed_runs_are_reproducible_in_their_forward(monkeypatch):
    """Same seed, same state, same batches: the DropPath draws of replayed steps follow the device-side seed (they differ from step to step and
    repeat from run to run)."""
    _, _, a = _run(True, 30001, 6, monkeypatch)
    _, _, b = _run(True, 30001, 6, monkeypatch)
    assert torch.equal(a[0]["feat"], b[0]["feat"])
    assert all(torch.equal(x["sel"], y["sel"]) for x, y in zip(a, b))
    assert not torch.equal(a[2]["feat"], a[3]["feat"])
