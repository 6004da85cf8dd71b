"""End-to-end SRFlexMatch.train_step + ParamUpdateHook on the HIP engine against the reference trace
(tests/golden/srflexmatch_trace.npz, produced by running the reference itself on a tiny ViT)."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import semireward_ref as S        # noqa: E402
from oracle import vit_ref as V               # noqa: E402
from oracle.gen_golden import TRACE, TRACE_C100, TRACE_FIX   # noqa: E402
from semireward_amd.algorithms import get_algorithm   # noqa: E402
from semireward_amd.nets import vit           # noqa: E402
from semireward_amd.utils import synth        # noqa: E402

DEV = "cuda:0"


def make_args(**kw):
    d = dict(algorithm="srflexmatch", num_classes=10, num_train_iter=2000, epoch=1, ema_m=0.0, ulb_loss_ratio=1.0,
             use_cat=True, amp=False, lr=5e-4, weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=50, optim="AdamW",
             T=0.5, p_cutoff=0.95, hard_label=True, thresh_warmup=True, ulb_dest_len=256, N_k=10, start_timing=100,
             feature_dim=128, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99, gpu=0, rank=0, world_size=1, distributed=False)
    d.update(kw)
    return argparse.Namespace(**d)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


from oracle.gen_golden import TRACE_FREE, TRACE_SOFT   # noqa: E402


def samp_of(arr, gs):
    return np.asarray(arr, np.float32).ravel()[::int(gs["stride"])]


@pytest.mark.parametrize("name,tr", [("srflexmatch_trace", TRACE), ("srflexmatch_c100_trace", TRACE_C100), ("srfixmatch_trace", TRACE_FIX),
                                     ("srfreematch_trace", TRACE_FREE), ("srsoftmatch_trace", TRACE_SOFT)])
def test_sr_train_step_trace(golden, name, tr):
    """train_step + ParamUpdateHook on the HIP engine against traces of the REFERENCE's own train_step.  The two srflexmatch traces (10 and
    100 classes) are non-degenerate by construction (oracle/gen_golden.py: labels are selected, rows are rejected, classwise_acc and the convex
    threshold are non-zero, mask2 takes both values; tests/test_oracle_golden.py asserts it on the fixture), and no max-prob the reference
    thresholds is closer than tr['min_margin'] to its threshold, while the engine's max-probs stay closer to the reference's than that (per row:
    deviation < distance to the nearer threshold; candidates were screened on the engine with tools/trace_diag.py) -- so a bf16-operand
    backbone has to reproduce EVERY mask, selected label and classwise_acc bit of every pass of all 12 iterations."""
    from oracle.gen_golden import trace_vit_params
    g = golden(name)
    flex = tr["algorithm"] == "srflexmatch"
    fix = tr["algorithm"] in ("srfixmatch", "srfreematch", "srsoftmatch")
    free = tr["algorithm"] == "srfreematch"
    soft = tr["algorithm"] == "srsoftmatch"
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_TINY_TEST)
    extra = dict(ema_p=tr["ema_p"], use_quantile=tr["use_quantile"], clip_thresh=tr["clip_thresh"], ent_loss_ratio=tr["ent_loss_ratio"]) if free else {}
    if soft:
        extra = dict(ema_p=tr["ema_p"], n_sigma=tr["n_sigma"], dist_uniform=tr["dist_uniform"], dist_align=True, per_class=False)
    alg = get_algorithm(make_args(algorithm=tr["algorithm"], p_cutoff=tr["p_cutoff"], num_classes=C, ulb_dest_len=tr["ulb_dest_len"],
                                  lr=tr.get("lr", 5e-4), **extra), vit.vit_tiny_test)
    T = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}   # noqa: E731
    P0 = trace_vit_params(cfg, seed, tr.get("head_gain", 1.0), tr.get("hot_classes", 0), tr.get("cold_scale", 0.25))
    alg.model.load_state_dict(T(P0))
    alg.rewarder.load_state_dict(T(synth.synth_params(S.rewarder_shapes(cfg.embed_dim, C), seed + 1)))
    alg.generator.load_state_dict(T(synth.synth_params(S.generator_shapes(cfg.embed_dim), seed + 2)))
    names = [nme for nme, _ in alg.model.named_parameters()]
    flips, worst_dev, min_slack, grad_rels, sign_stats, m2_clear = 0, 0.0, 1.0, [], [], [0, 0]
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        alg.it = it
        alg.optimizer.sched_step = it                 # LambdaLR position of the reference at iteration `it`
        K = int(g[f"{p}/K"])
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
        alg.inject_droppath = [torch.from_numpy(synth.synth_droppath(seed + 1000 * (n + 1) + k, V.drop_path_probs(cfg), Bl + 2 * Bu))
                               for k in range(K + 1)]
        alg.trace = {}
        before = alg.rewarder.flat.clone()
        pbefore = {nme: v.detach().clone() for nme, v in alg.model.named_parameters()}
        out, log = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))   # idx_ulb dropped for srfixmatch
        alg.out_dict, alg.log_dict = out, log
        assert alg.optimizer.lr_factor() == pytest.approx(float(g[f"{p}/lr_factor"]), rel=1e-9, abs=1e-12)
        # ---- the step's combined backward (supervised rows of pass 0 + the mask * mask2-weighted strong rows of the LAST pass, through the engine's
        # hand-written backward) against the reference's autograd gradients of the same step, BEFORE the optimizer consumes them
        if flex:
            num = den = 0.0
            for nme, gv in alg.model.named_grads():
                gs = g.samp(f"{p}/grad/{nme}")
                a = samp_of(gv.cpu().numpy(), gs).astype(np.float64)
                num += float(((a - gs["sample"]) ** 2).sum()); den += float((gs["sample"].astype(np.float64) ** 2).sum())
            grad_rels.append((num / max(den, 1e-30)) ** 0.5)
        alg.call_hook("after_train_step")
        assert alg.trace["K"] == K
        masks = np.stack([m.cpu().numpy() for m in alg.trace["masks"]])
        want = g[f"{p}/masks"]
        if soft:      # the mask is a continuous weight (SoftMatch): compare values; bf16 logits move max-probs by ~1e-2 relative
            np.testing.assert_allclose(masks, want, rtol=0.0, atol=6e-2)
            assert float(log["train/util_ratio"]) == pytest.approx(float(g[f"{p}/log/util_ratio"]), abs=4e-2)
        else:
            flips += int((masks != want).sum())
        if flex:
            # what the hook thresholds: the engine's max-probs stay inside the fixture's margin of the reference's, so every comparison
            # (>= p_cutoff * acc / (2 - acc): mask; >= p_cutoff: select) has the reference's outcome
            mpv = alg.trace["max_probs"].cpu().numpy().reshape(want.shape)
            refp = g[f"{p}/mask_probs"]
            devs = np.abs(mpv - refp)
            margin = np.minimum(np.abs(refp - g[f"{p}/mask_thr"]), np.abs(refp - tr["p_cutoff"]))      # to the nearer of the two thresholds, per row
            worst_dev, min_slack = max(worst_dev, float(devs.max())), min(min_slack, float((margin - devs).min()))
            assert float(margin.min()) >= tr["min_margin"] and float(devs.max()) < 5e-2, (p, float(margin.min()), float(devs.max()))
            assert (devs < margin).all(), (p, float((margin - devs).min()))      # every decision is the reference's with room to spare
            assert np.array_equal(alg.trace["pseudo"].cpu().numpy().reshape(want.shape), g[f"{p}/pseudo_label"]), p
            assert np.array_equal(masks, want), p
            if K:
                r = alg.trace["reward"].cpu().numpy().reshape(K, Bu)
                np.testing.assert_allclose(r, g[f"{p}/reward"], rtol=0, atol=5e-3)
                # mask2 = reward >= mean(reward) of the pass (:100-101).  The rewards of a pass sit within a few 1e-2 of their mean (the
                # rewarder's softmax couples the batch), so: identical wherever the reference's reward is further from its pass mean than
                # twice the largest reward deviation of that pass (the mean moves by at most that deviation), and exactly consistent with
                # the engine's own rewards everywhere
                rg = g[f"{p}/reward"]
                clear = np.abs(rg - rg.mean(axis=1, keepdims=True)) > 2.0 * np.abs(r - rg).max(axis=1, keepdims=True) + 1e-6
                m2 = alg.trace["mask2"].cpu().numpy().reshape(K, Bu)
                assert np.array_equal(m2[clear], g[f"{p}/mask2"][clear]), p
                assert np.array_equal(m2, (r >= r.mean(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)), p
                m2_clear[0] += int(clear.sum()); m2_clear[1] += clear.size
        for k_ in ("sup_loss", "unsup_loss", "total_loss"):
            assert float(log["train/" + k_]) == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=6e-2, abs=5e-3), (p, k_)
        # AdamW moves every weight by ~lr per step whatever |g| is, so bf16-operand gradient noise turns into a
        # slowly growing parameter gap vs the fp32 trajectory: tolerance = 2e-2 + 1.5e-2 per full-lr (5e-4) step taken
        ftol = 2e-2 + 1.5e-2 * (tr.get("lr", 5e-4) / 5e-4) * sum(1 for j in tr["its"][:n] if j >= tr["num_warmup_iter"])
        for k_ in ("x_lb", "x_ulb_w", "x_ulb_s"):
            assert rel(out["feat"][k_].cpu(), g[f"{p}/feat/{k_}"]) < ftol, (p, k_)
        assert int(not torch.equal(before, alg.rewarder.flat)) == int(g[f"{p}/rewarder_updated"]), p
        # ---- direction of the AdamW update: the engine's parameter change of this step against the reference's (its parameters after this
        # iteration minus after the previous one -- the reference's optimizer only steps at the trace's iterations), on the sampled elements whose
        # reference gradient is above the noise floor of a bf16-operand backward (>= 10 % of the tensor's rms gradient)
        if flex and float(g[f"{p}/lr_factor"]) > 0.0:
            agree = tot = 0
            for nme, v in alg.model.named_parameters():
                gs, gg = g.samp(f"{p}/param/{nme}"), g.samp(f"{p}/grad/{nme}")
                prev = samp_of(P0[nme], gs) if n == 0 else g.samp(f"it{tr['its'][n - 1]}/param/{nme}")["sample"]
                d_ref = gs["sample"].astype(np.float64) - prev.astype(np.float64)
                d_eng = samp_of((v.detach() - pbefore[nme]).cpu().numpy(), gs).astype(np.float64)
                gr = gg["sample"].astype(np.float64)
                keep = (np.abs(gr) >= 0.1 * np.sqrt((gr ** 2).mean() + 1e-30)) & (d_ref != 0.0)
                agree += int((np.sign(d_eng[keep]) == np.sign(d_ref[keep])).sum()); tot += int(keep.sum())
            sign_stats.append((it, agree, tot))
        if soft:
            sm, da = alg.hooks_dict["MaskingHook"], alg.hooks_dict["DistAlignHook"]
            assert float(sm.prob_max_mu_t) == pytest.approx(float(g[f"{p}/mu"]), rel=2e-2)
            assert float(sm.prob_max_var_t) == pytest.approx(float(g[f"{p}/var"]), rel=5e-2)
            assert rel(da.p_model.cpu(), g[f"{p}/p_model"]) < 2e-2 and rel(da.p_target.cpu(), g[f"{p}/p_target"]) < 2e-2
        if free:
            h = alg.hooks_dict["MaskingHook"]
            assert float(h.time_p) == pytest.approx(float(g[f"{p}/time_p"]), rel=2e-2)
            assert rel(h.p_model.cpu(), g[f"{p}/p_model"]) < 2e-2
            # label_hist is an EMA of an argmax HISTOGRAM (freematch/utils.py:44-49): identical pseudo labels -> equal to round-off
            assert rel(h.label_hist.cpu(), g[f"{p}/label_hist"]) < 2e-2, rel(h.label_hist.cpu(), g[f"{p}/label_hist"])
        mr = float(g[f"{p}/max_reward"])
        assert (np.isinf(mr) and np.isinf(float(alg.max_reward))) or float(alg.max_reward) == pytest.approx(mr, rel=1e-2), p
        if not fix and masks.shape == want.shape and (masks == want).all():
            sel = alg.hooks_dict["MaskingHook"].selected_label.cpu().numpy()
            nz = np.nonzero(sel != -1)[0]
            assert np.array_equal(nz, g[f"{p}/sel_idx"]) and np.array_equal(sel[nz], g[f"{p}/sel_val"]), p
            acc = alg.hooks_dict["MaskingHook"].classwise_acc.cpu().numpy()
            assert np.array_equal(acc.view(np.uint32), g[f"{p}/accs"][-1].view(np.uint32)), p
    # bf16 logits vs the fp32 reference may flip a row that sits on a threshold; on these traces none does
    assert flips == 0, flips
    if flex:
        print("\n%s: worst |max-prob - reference| %.2e, smallest (margin - deviation) %.2e (fixture margin %.1e); grad rel-L2 per iteration %s; update-sign agreement %s; mask2 rows compared with the reference %d / %d" % (
            name, worst_dev, min_slack, tr["min_margin"], ["%.3f" % r for r in grad_rels], ["%d: %d/%d" % t for t in sign_stats], m2_clear[0], m2_clear[1]))
        # the fixture is not degenerate for the engine either: rejections, selections and a non-zero table through train_step
        allm = np.concatenate([g[f"it{it}/masks"].ravel() for it in tr["its"]])
        h = alg.hooks_dict["MaskingHook"]
        assert 0.2 < allm.mean() < 0.9 and int((h.selected_label != -1).sum()) > 0 and float(h.classwise_acc.max()) > 0
        # gradients: rel-L2 over the sampled elements of all parameters, every iteration (pre-SR steps: K = 0; SR steps: loss of the LAST pass)
        # (measured on MI355X: 0.005 .. 0.010 in every iteration of both traces; the model-level bound of tests/test_gpu_vit.py is 6e-2)
        assert max(grad_rels) < 2.5e-2, grad_rels
        ag, tt = sum(t[1] for t in sign_stats), sum(t[2] for t in sign_stats)
        assert tt > 2000 and ag >= 0.99 * tt, (ag, tt, sign_stats)
        assert m2_clear[0] > 0.5 * m2_clear[1], m2_clear          # most (pass, row) mask2 decisions were compared with the reference's
    else:
        # backbone parameters after the trace's AdamW steps (bf16-operand gradients) stay close to the fp32 reference trajectory
        worst = 0.0
        for nme, v in alg.model.named_parameters():
            gs = g.samp(f"it{tr['its'][-1]}/param/{nme}")
            a = v.reshape(-1).cpu().numpy()[::gs["stride"]]
            worst = max(worst, float(np.abs(a - gs["sample"]).max()))
        assert worst < 4e-3, worst      # <= a handful of lr-sized (5e-4) steps


def test_train_loop_and_checkpoint(tmp_path):
    """AlgorithmBase.train() drives train_step + hooks; get_save_dict / load_model round-trip incl. hook state."""
    alg = get_algorithm(make_args(num_train_iter=6, start_timing=2, N_k=2, num_warmup_iter=0), vit.vit_tiny_test)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(V.param_shapes(V.VitCfg(num_classes=10, **V.VIT_TINY_TEST)), 3).items()})
    batches = []
    for i in range(6):
        b = {k: torch.from_numpy(v) for k, v in synth.synth_batch(100 + i, 4, 4, 8, 10, 256).items()}
        batches.append(({"x_lb": b["x_lb"], "y_lb": b["y_lb"], "idx_lb": torch.arange(4)},
                        {"idx_ulb": b["idx_ulb"], "x_ulb_w": b["x_ulb_w"], "x_ulb_s": b["x_ulb_s"], "y_ulb": b["y_lb"]}))
    alg.train(batches)
    assert alg.it == 6 and alg.optimizer.step_count == 6 and alg.rewarder_optimizer.steps >= 2
    assert np.isfinite(float(alg.log_dict["train/total_loss"]))
    assert float(alg.model.grad.abs().max()) == 0.0             # zero_grad fused into the optimizer launch
    alg.save_model("latest_model.pth", str(tmp_path))
    alg2 = get_algorithm(make_args(num_train_iter=6, start_timing=2, N_k=2, num_warmup_iter=0), vit.vit_tiny_test)
    alg2.load_model(str(tmp_path / "latest_model.pth"))
    assert torch.equal(alg2.model.flat, alg.model.flat) and torch.equal(alg2.rewarder.flat, alg.rewarder.flat)
    h1, h2 = alg.hooks_dict["MaskingHook"], alg2.hooks_dict["MaskingHook"]
    assert torch.equal(h1.selected_label, h2.selected_label) and torch.equal(h1.hist, h2.hist)


def test_srpseudolabel_trace(golden):
    from oracle.gen_golden import TRACE_PL as tr
    g = golden("srpseudolabel_trace")
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_TINY_TEST)
    alg = get_algorithm(make_args(algorithm="srpseudolabel", p_cutoff=tr["p_cutoff"], unsup_warm_up=tr["unsup_warm_up"]), vit.vit_tiny_test)
    T = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}   # noqa: E731
    alg.model.load_state_dict(T(synth.synth_params(V.param_shapes(cfg), seed)))
    alg.rewarder.load_state_dict(T(synth.synth_params(S.rewarder_shapes(cfg.embed_dim, C), seed + 1)))
    alg.generator.load_state_dict(T(synth.synth_params(S.generator_shapes(cfg.embed_dim), seed + 2)))
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        alg.it = it
        alg.optimizer.sched_step = it
        K = int(g[f"{p}/K"])
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
        dpl = torch.from_numpy(synth.synth_droppath(seed + 1000 * (n + 1), V.drop_path_probs(cfg), Bl))
        dpu = [torch.from_numpy(synth.synth_droppath(seed + 1000 * (n + 1) + 1 + k, V.drop_path_probs(cfg), Bu)) for k in range(K + 1)]
        alg.inject_droppath = [(dpl, dpu[0])] + dpu[1:]
        alg.trace = {}
        before = alg.rewarder.flat.clone()
        out, log = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))   # idx_ulb / x_ulb_s dropped
        alg.out_dict, alg.log_dict = out, log
        alg.call_hook("after_train_step")
        assert alg.trace["K"] == K
        masks = np.stack([m.cpu().numpy() for m in alg.trace["masks"]])
        # the fixture's cut-off (0.165) is at least 6.6e-3 away from every max-prob the reference thresholds (`mask_probs`, swept in
        # oracle/gen_golden.py): the bf16-operand backbone moves a max-prob by far less, so EVERY mask of every pass must match
        mpv = alg.trace["max_probs"].cpu().numpy().reshape(masks.shape)
        assert float(np.abs(g[f"{p}/mask_probs"] - tr["p_cutoff"]).min()) > 6e-3
        # before the first parameter update the probabilities differ by bf16 operand rounding only; afterwards the AdamW trajectory of
        # bf16-operand gradients drifts from the fp32 one (the same growth as the feature tolerance below) -- the MASKS must still all agree
        dev = float(np.abs(mpv - g[f"{p}/mask_probs"]).max())
        assert dev < (3e-3 if n == 0 else 2.5e-2), (p, dev)
        assert np.array_equal(masks, g[f"{p}/masks"]), (p, mpv[masks != g[f"{p}/masks"]], g[f"{p}/mask_probs"][masks != g[f"{p}/masks"]])
        for k_ in ("sup_loss", "unsup_loss", "total_loss"):
            assert float(log["train/" + k_]) == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=6e-2, abs=5e-3), (p, k_)
        ftol = 2e-2 + 1.5e-2 * sum(1 for j in tr["its"][:n] if j >= tr["num_warmup_iter"])
        for k_ in ("x_lb", "x_ulb_w"):
            assert rel(out["feat"][k_].cpu(), g[f"{p}/feat/{k_}"]) < ftol, (p, k_)
        assert int(not torch.equal(before, alg.rewarder.flat)) == int(g[f"{p}/rewarder_updated"]), p


def test_evaluate_matches_oracle_forward():
    """AlgorithmBase.evaluate (algorithmbase.py:377-457): eval-mode forward of the EMA weights over a loader, CE loss with
    ignore_index = -1, sklearn-convention metrics -- against the CPU oracle's forward on the same parameters."""
    import torch.nn.functional as F
    cfg = V.VitCfg(num_classes=10, **V.VIT_TINY_TEST)
    alg = get_algorithm(make_args(num_train_iter=6, start_timing=2, N_k=2, num_warmup_iter=0), vit.vit_tiny_test)
    P = synth.synth_params(V.param_shapes(cfg), 7)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    rng = np.random.Generator(np.random.PCG64(11))
    batches, xs, ys = [], [], []
    for n in (5, 8, 3):
        x = rng.standard_normal((n, 3, cfg.img_size, cfg.img_size)).astype(np.float32)
        y = rng.integers(0, 10, size=(n,), dtype=np.int64)
        y[0] = -1 if n == 8 else y[0]                                        # one ignored row
        batches.append({"x_lb": torch.from_numpy(x), "y_lb": torch.from_numpy(y)})
        xs.append(x); ys.append(y)
    out = alg.evaluate("eval", loader=batches, return_logits=True)
    Pt = {k: torch.from_numpy(v) for k, v in P.items()}
    ref = V.vit_forward(Pt, torch.from_numpy(np.concatenate(xs)), cfg)["logits"]
    y = torch.from_numpy(np.concatenate(ys))
    assert rel(out["eval/logits"], ref.numpy()) < 2e-2
    tot, off = 0.0, 0
    for b in ys:                                                             # reference: sum_b CE_mean(batch b) * len(b) / N
        tot += float(F.cross_entropy(ref[off:off + len(b)], y[off:off + len(b)], ignore_index=-1)) * len(b); off += len(b)
    assert out["eval/loss"] == pytest.approx(tot / off, rel=2e-2)
    yp = ref.argmax(-1).numpy()
    if (out["eval/logits"].argmax(-1) == yp).all():
        from semireward_amd.core.algorithmbase import AlgorithmBase
        keep = np.ones_like(yp, dtype=bool)
        m = AlgorithmBase.classification_metrics(np.concatenate(ys), yp)
        assert out["eval/top-1-acc"] == pytest.approx(m["top-1-acc"]) and out["eval/F1"] == pytest.approx(m["F1"])
    assert set(out) == {"eval/loss", "eval/top-1-acc", "eval/balanced_acc", "eval/precision", "eval/recall", "eval/F1", "eval/logits"}


def test_full_size_step_properties():
    """BASELINE.json's north-star configuration at FULL size (ViT-S/2, C = 100, 8/8/8, K = sr_decay() = 8, ulb_dest_len 50 000): the CPU
    oracle cannot step this in seconds, so parity goes through size-independent properties --
      * integer work bit-exact: every pass's FlexMatch mask / selected_label / classwise_acc equals the numpy oracle fed with the
        engine's own max-probs and argmax of that pass (sequential state, 9 passes); reward mask2 == (reward >= per-pass mean);
      * K = 8 passes, (1 + K) * 24 image rows, finite losses, util_ratio = mean(mask0);
      * linearity of the hand-written backward: grads(2 * dlogits) == 2 * grads(dlogits) to fp32 round-off (no atomics on bf16);
      * the step is reproducible: same state + same inputs + same DropPath draws -> same masks and logits."""
    from oracle import hooks_ref as H
    NSa = dict(algorithm="srflexmatch", num_classes=100, num_train_iter=204800, ulb_dest_len=50000, start_timing=20000, feature_dim=384,
               num_warmup_iter=5120)
    # A random-init backbone with the stock classifier never reaches p_cutoff = 0.95 at 100 classes (weak-row max-probs 0.03-0.1) and an empty
    # selected_label table keeps every threshold at 0: all masks would be 1 and the table untouched.  So the step starts from a MID-TRAINING hook
    # state (46 000 of the 50 000 entries selected, the class counts skewed towards the classes this backbone predicts, so their classwise_acc
    # -- and the convex threshold 0.95 acc / (2 - acc) -- is high; the batch's own entries unselected) and a classifier loud enough (x 24) that
    # the weak rows' max-probs straddle 0.95 (0.4 .. 1.0 on the CPU oracle): rows are selected, rows are rejected, the table changes.
    HEAD_GAIN = 24.0
    P0 = synth.synth_params(V.param_shapes(V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)), 0)
    P0["head.weight"] = P0["head.weight"] * np.float32(HEAD_GAIN)
    rs = np.random.Generator(np.random.PCG64(77))
    w = np.ones(100); w[[97, 11, 45, 20, 84, 90, 26, 52, 35]] = [60, 55, 50, 40, 35, 30, 25, 20, 15]
    sel0 = rs.choice(100, size=50000, p=w / w.sum()).astype(np.int64)
    sel0[rs.permutation(50000)[:4000]] = -1
    b = synth.synth_batch(100, 8, 8, 32, 100, 50000)
    sel0[b["idx_ulb"]] = -1
    st0 = H.FlexMatchState(50000, 100, True)
    st0.selected_label[:] = sel0
    st0.update()                                   # classwise_acc as the previous step's last masking call left it (utils.py:61)
    assert st0.classwise_acc.max() == 1.0 and (st0.classwise_acc > 0.3).sum() >= 6

    def make():
        alg = get_algorithm(make_args(**NSa), vit.vit_small_patch2_32)
        assert [n for n, _ in alg.model.names_shapes] == list(P0)
        alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in P0.items()})
        h = alg.hooks_dict["MaskingHook"]
        h.selected_label = torch.from_numpy(sel0)                       # (the setter recounts the histogram, as load_model does)
        h.classwise_acc = torch.from_numpy(st0.classwise_acc.copy()).to(DEV)
        alg.it = 30000
        alg.optimizer.sched_step = alg.it
        return alg
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    dps = [torch.from_numpy(synth.synth_droppath(900 + k, V.drop_path_probs(cfg), 24)) for k in range(9)]
    runs = []
    for _ in range(2):
        alg = make()
        alg.inject_droppath = dps
        alg.trace = {}
        out, log = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))
        torch.cuda.synchronize()
        runs.append((alg, out, log, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in alg.trace.items()}))
    alg, out, log, tr = runs[0]
    K, nu, C = tr["K"], 8, 100
    assert K == 8 and tr["logits"].shape[:2] == (9, 24)
    # --- integer work against the oracle on the engine's own numbers
    st = H.FlexMatchState(50000, C, True)
    st.selected_label[:] = sel0
    st.classwise_acc[:] = st0.classwise_acc
    mp, mi = tr["max_probs"].cpu().numpy().reshape(9, nu), tr["pseudo"].cpu().numpy().reshape(9, nu)
    idx = b["idx_ulb"]
    thr_at_reject, allmask = [], []
    for k in range(9):
        probs = np.zeros((nu, C), np.float32)
        probs[np.arange(nu), mi[k]] = mp[k]                      # masking only looks at (max, argmax) of each row
        acc_k = st.classwise_acc[mi[k]].copy()
        want = st.masking(probs, idx, 0.95)
        assert np.array_equal(tr["masks"][k].cpu().numpy(), want), k
        thr_at_reject += list((np.float32(0.95) * (acc_k / (np.float32(2.0) - acc_k)))[want == 0.0])
        allmask.append(want)
    h = alg.hooks_dict["MaskingHook"]
    assert np.array_equal(h.selected_label.cpu().numpy(), st.selected_label)
    assert np.array_equal(h.classwise_acc.cpu().numpy().view(np.uint32), st.classwise_acc.view(np.uint32))
    # ... and that work was not trivial: rows above and below 0.95, rejected rows (against a non-zero convex threshold), accepted rows, entries
    # of the batch newly selected in the 50 000-entry table and a class histogram that moved
    allmask = np.stack(allmask)
    assert mp.min() < 0.9 and mp.max() > 0.95, (mp.min(), mp.max())
    assert 0.0 < allmask.mean() < 1.0 and len(thr_at_reject) >= 3 and min(thr_at_reject) > 0.05, (allmask.mean(), thr_at_reject)
    assert (st.selected_label[idx] != -1).any() and (sel0[idx] == -1).all()
    assert not np.array_equal(st.classwise_acc, st0.classwise_acc)
    assert 0.0 < float(log["train/util_ratio"]) <= 1.0
    r = tr["reward"].cpu().numpy().reshape(K, nu)
    assert np.array_equal(tr["mask2"].cpu().numpy().reshape(K, nu), (r >= r.mean(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32))
    assert 0.0 <= r.min() and r.max() <= 1.0
    assert float(log["train/util_ratio"]) == pytest.approx(float(tr["masks"][0].mean()), abs=1e-7)
    for k_ in ("sup_loss", "unsup_loss", "total_loss"):
        assert np.isfinite(float(log["train/" + k_]))
    # --- the step's logits against the fp32 CPU oracle (oracle/vit_ref.py, pinned to the reference by tests/golden/vit.npz) on the same 24 images
    # and the same DropPath draws: pass 0 (labelled rows = gradient rows with activations kept; weak rows = the 105-image launch that is read;
    # strong rows = read or deferred launch, both the production fused chain attn_block -> mlp_fused_proj + ln_next) and the last pass
    # (strong rows = gradient rows).  One 24-image CPU forward of ViT-S/2 takes a few seconds.
    Pt = {k: torch.from_numpy(v) for k, v in P0.items()}
    x24 = torch.from_numpy(np.concatenate([b["x_lb"], b["x_ulb_w"], b["x_ulb_s"]]))
    plan = alg._plans[(8, 8, 8, True, False)]
    assert plan.inf_cols.numel() * 257 >= vit._FUSED_MLP_MIN_ROWS and plan.rest_cols.numel() * 257 >= vit._FUSED_MLP_MIN_ROWS   # fused launches
    L = tr["logits"].cpu().numpy()
    for k in (0, K):
        with torch.no_grad():
            ref = V.vit_forward(Pt, x24, cfg, dps[k])
        want_l, want_f = ref["logits"].numpy(), ref["feat"].numpy()
        for name, rows in (("lb", slice(0, 8)), ("weak", slice(8, 16)), ("strong", slice(16, 24))):
            assert rel(L[k, rows], want_l[rows]) < 2e-2, (k, name, rel(L[k, rows], want_l[rows]))
            assert rel(tr["feats"][k, rows].cpu().numpy(), want_f[rows]) < 2e-2, (k, name)
        # ... so the engine's pseudo labels are the oracle's wherever the oracle's own top-2 margin exceeds the logit tolerance
        top2 = np.sort(want_l[8:16], axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 4e-2 * np.abs(want_l[8:16]).max()
        assert np.array_equal(mi[k][clear], want_l[8:16].argmax(1)[clear]), k
    # --- reproducibility of the forward / filter
    tr2 = runs[1][3]
    assert torch.equal(tr["logits"], tr2["logits"]) and all(torch.equal(a, b_) for a, b_ in zip(tr["masks"], tr2["masks"]))
    # --- linearity of the backward
    m = alg.model
    x = torch.from_numpy(np.concatenate([b["x_lb"], b["x_ulb_s"]])).to(DEV)
    dp = dps[0][:, :, :16].contiguous().to(DEV)
    lg, _, ctx = m.forward_features(x, None, dp, save=True)
    dl = torch.from_numpy((np.random.Generator(np.random.PCG64(3)).standard_normal((16, C)) * 1e-2).astype(np.float32)).to(DEV)
    m.zero_grad(); m.backward(ctx, dl); g1 = m.grad.clone()
    lg, _, ctx = m.forward_features(x, None, dp, save=True)
    m.zero_grad(); m.backward(ctx, 2.0 * dl); g2 = m.grad.clone()
    assert rel(g2.cpu(), (2.0 * g1).cpu().numpy()) < 2e-3        # bf16 rounding of the scaled output gradients is not exactly linear


WORST_TENSOR_REL = 0.12         # per-tensor bound of the step gradient at full size: measured 0.062 (pos_embed, it = 1000) / 0.028 (it = 30000)


def test_full_size_reference_trace(golden):
    """BASELINE.json configs[1] at FULL size against the REFERENCE's own train_step (tests/golden/srflexmatch_full_trace.npz, oracle/gen_golden.py
    gen_trace_full: ViT-S/2, 100 classes, 8 / 8 / 8, ulb_dest_len 50 000, the yaml's hyper-parameters; two single steps from the same mid-training
    state -- it = 1000: K = 0 and the stage-1 rewarder update, it = 30000: K = 8).  Reference-side values for everything the step decides or
    returns: per pass the max-probs / thresholds / pseudo labels / masks the hook saw, the K reward vectors and mask2, the three losses, the
    sampled step gradients of every parameter tensor, the features, the table entries of the batch after the step.  Masks, pseudo labels and
    the table are bit-exact (asserted per row: the engine's max-prob deviates from the reference's by less than that row's distance to its
    nearer threshold and to the runner-up class); rewards 3e-2; losses 6e-2 relative; gradients 5e-2 rel-L2 (pooled samples of every tensor)."""
    from oracle.gen_golden import FULL, full_hook_state, trace_vit_params
    g = golden("srflexmatch_full_trace")
    tr = FULL
    C, Bl, Bu = tr["C"], tr["Bl"], tr["Bu"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_SMALL_P2_32)
    T_ = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}   # noqa: E731
    P0 = trace_vit_params(cfg, tr["seed"], tr["head_gain"])
    bseed = int(g["meta/bseed"])
    b = synth.synth_batch(bseed, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
    sel0, acc0 = full_hook_state(b["idx_ulb"])
    for it in [int(i) for i in g["meta/its"]]:
        p = f"it{it}"
        K = int(g[f"{p}/K"])
        alg = get_algorithm(make_args(algorithm="srflexmatch", num_classes=C, num_train_iter=tr["num_train_iter"], ulb_dest_len=tr["ulb_dest_len"],
                                      start_timing=tr["start_timing"], feature_dim=cfg.embed_dim, num_warmup_iter=tr["num_warmup_iter"],
                                      p_cutoff=tr["p_cutoff"], N_k=tr["N_k"], lr=tr["lr"]), vit.vit_small_patch2_32)
        alg.model.load_state_dict(T_(P0))
        alg.rewarder.load_state_dict(T_(synth.synth_params(S.rewarder_shapes(cfg.embed_dim, C), tr["seed"] + 1)))
        alg.generator.load_state_dict(T_(synth.synth_params(S.generator_shapes(cfg.embed_dim), tr["seed"] + 2)))
        h = alg.hooks_dict["MaskingHook"]
        h.selected_label = torch.from_numpy(sel0.copy())
        h.classwise_acc = torch.from_numpy(acc0.copy()).to(DEV)
        alg.it = it
        alg.optimizer.sched_step = it
        alg.inject_droppath = [torch.from_numpy(synth.synth_droppath(int(g[f"{p}/dp_seed0"]) + k, V.drop_path_probs(cfg), Bl + 2 * Bu))
                               for k in range(K + 1)]
        alg.trace = {}
        rbefore = alg.rewarder.flat.clone()
        out, log = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))
        torch.cuda.synchronize()
        assert alg.trace["K"] == K and alg.optimizer.lr_factor() == pytest.approx(float(g[f"{p}/lr_factor"]), rel=1e-9, abs=1e-12)
        # ---- the score filter: every pass's decisions are the reference's, with room
        want = g[f"{p}/masks"]
        masks = np.stack([m.cpu().numpy() for m in alg.trace["masks"]])
        mpv = alg.trace["max_probs"].cpu().numpy().reshape(want.shape)
        refp = g[f"{p}/mask_probs"]
        devs = np.abs(mpv - refp)
        # room of a row's decisions: distance of the reference's max-prob from the nearer of its two thresholds, and from the runner-up class.
        # (A max-prob p moves by ~p (1 - p) x the deviation of its logit gap: 0.04-0.07 at p ~ 0.5-0.7 with this classifier gain, 1e-3 at 0.99;
        # the fixture's batch was chosen so that every row keeps its room -- gen_golden.run_full_step, tools/full_trace_diag.py prints the rows.)
        room = np.minimum(np.minimum(np.abs(refp - g[f"{p}/mask_thr"]), np.abs(refp - tr["p_cutoff"])), g[f"{p}/label_gap"])
        assert float(devs.max()) < 0.12 and (devs < room).all(), (p, float(devs.max()), float((room - devs).min()))
        assert np.array_equal(alg.trace["pseudo"].cpu().numpy().reshape(want.shape), g[f"{p}/pseudo_label"]), p
        assert np.array_equal(masks, want), p
        assert 0.0 < want.mean() < 1.0                                     # rows selected and rows rejected at full size
        assert np.array_equal(h.selected_label.cpu().numpy()[b["idx_ulb"]], g[f"{p}/sel_after_batch"])
        assert int((h.selected_label != -1).sum()) == int(g[f"{p}/n_selected_after"])
        assert np.array_equal(h.classwise_acc.cpu().numpy().view(np.uint32), g[f"{p}/accs"][-1].view(np.uint32))
        if K:
            r = alg.trace["reward"].cpu().numpy().reshape(K, Bu)
            rg = g[f"{p}/reward"]
            np.testing.assert_allclose(r, rg, rtol=0, atol=3e-2)       # (x 24 classifier: the features' bf16 noise reaches the rewarder's batch softmax)
            clear = np.abs(rg - rg.mean(axis=1, keepdims=True)) > 2.0 * np.abs(r - rg).max(axis=1, keepdims=True) + 1e-6
            m2 = alg.trace["mask2"].cpu().numpy().reshape(K, Bu)
            assert clear.mean() > 0.5 and np.array_equal(m2[clear], g[f"{p}/mask2"][clear]), (p, float(clear.mean()))
            assert np.array_equal(m2, (r >= r.mean(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)), p
        # ---- losses, features, the step gradient (before the optimizer consumes it), the rewarder update
        for k_ in ("sup_loss", "unsup_loss", "total_loss"):
            # (measured: 0.1-0.3 % -- sup_loss 31.90 vs 31.94, unsup_loss 5.175 vs 5.184; bound = 3 x that)
            assert float(log["train/" + k_]) == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=1e-2, abs=5e-3), (p, k_)
        assert float(log["train/util_ratio"]) == pytest.approx(float(g[f"{p}/log/util_ratio"]), abs=1e-6)
        for k_ in ("x_lb", "x_ulb_w", "x_ulb_s"):
            assert rel(out["feat"][k_].cpu().numpy(), g[f"{p}/feat/{k_}"]) < 2e-2, (p, k_)
        num = den = 0.0
        worst = (0.0, None)
        for nme, gv in alg.model.named_grads():
            gs = g.samp(f"{p}/grad/{nme}")
            a = samp_of(gv.cpu().numpy(), gs).astype(np.float64)
            e2, n2 = float(((a - gs["sample"]) ** 2).sum()), float((gs["sample"].astype(np.float64) ** 2).sum())
            num += e2; den += n2
            if n2 > 0 and not nme.endswith("attn.qkv.bias") and nme != "cls_token":       # (K third of the qkv bias: analytic gradient 0, see DESIGN)
                worst = max(worst, ((e2 / n2) ** 0.5, nme))
        # (0.020 / 0.037 measured: 12 bf16-operand layers under a x 24 classifier; the tiny traces sit at 0.005-0.010, the documented bound of
        # backbone gradients against an fp32 reference is 6e-2)
        print("full trace %s: pooled step-gradient rel-L2 %.4f, worst tensor %.4f (%s)" % (p, (num / den) ** 0.5, worst[0], worst[1]))
        assert (num / den) ** 0.5 < 5e-2, (p, (num / den) ** 0.5)
        assert worst[0] < WORST_TENSOR_REL, (p, worst)
        alg.out_dict, alg.log_dict = out, log
        alg.call_hook("after_train_step")
        assert int(not torch.equal(rbefore, alg.rewarder.flat)) == int(g[f"{p}/rewarder_updated"])
        for k_, v in alg.rewarder.named_parameters():
            gs = g.samp(f"{p}/rewarder/{k_}")
            np.testing.assert_allclose(samp_of(v.detach().cpu().numpy(), gs), gs["sample"], rtol=0, atol=2e-3, err_msg=p + k_)


def measure_mask_identity(g, gain, batches=None, cmp=None, aux=None):
    """The engine's score-filter decisions on every (batch, it) step of the reference sweep at one classifier gain (fixture
    srflexmatch_full_sweep.npz, oracle/gen_golden.py gen_sweep_full).  Returns a dict of counts.  Within a step the FlexMatch state is order
    dependent (a differing row of pass k moves selected_label -> classwise_acc -> the thresholds of every later pass), so rows are judged in
    pass order up to the FIRST pass with a differing row; its differing rows are classified:
      label   the engine's pseudo label is another class.  Then the reference's top-two probabilities have crossed: with gap = p1 - p2 of the
              reference, the two class probabilities moved by >= gap between them -- the row's gap is recorded (flips must stay confined to
              rows whose gap is within twice the engine's class-probability deviation);
      mask    same label (so the engine's max-prob is the SAME class's probability and its deviation is observable), other mask: the deviation
              must have reached the row's room = distance of the reference's max-prob from the nearer of its two thresholds.
    Rows of later passes are counted as downstream.
    cmp: another fixture with the same keys to compare the decisions with instead of ``g``; aux: a third set of max-probs (the rounding
    model's, gen_sweep_full_emu) whose squared distances from the engine's and from cmp's are accumulated (sq_engine_aux, sq_aux_cmp)."""
    from oracle.gen_golden import FULL, full_hook_state, trace_vit_params
    cmp = g if cmp is None else cmp
    tr = dict(FULL, head_gain=gain)
    C, Bl, Bu = tr["C"], tr["Bl"], tr["Bu"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_SMALL_P2_32)
    T_ = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}   # noqa: E731
    alg = get_algorithm(make_args(algorithm="srflexmatch", num_classes=C, num_train_iter=tr["num_train_iter"], ulb_dest_len=tr["ulb_dest_len"],
                                  start_timing=tr["start_timing"], feature_dim=cfg.embed_dim, num_warmup_iter=tr["num_warmup_iter"],
                                  p_cutoff=tr["p_cutoff"], N_k=tr["N_k"], lr=tr["lr"]), vit.vit_small_patch2_32)
    alg.model.load_state_dict(T_(trace_vit_params(cfg, tr["seed"], gain)))
    rew0 = T_(synth.synth_params(S.rewarder_shapes(cfg.embed_dim, C), tr["seed"] + 1))
    alg.generator.load_state_dict(T_(synth.synth_params(S.generator_shapes(cfg.embed_dim), tr["seed"] + 2)))
    h = alg.hooks_dict["MaskingHook"]
    st = dict(steps=0, rows=0, flipped_rows=0, label_mismatch_rows=0, steps_with_a_difference=0, first_label_flips=0, first_label_flip_max_gap=0.0,
              first_mask_flips=0, first_mask_flips_inside_their_room=0, rows_at_risk=0, downstream_rows=0, max_dev_same_label=0.0, table_entries=0,
              table_mismatches=0, mask2_rows=0, mask2_flips=0, mask2_flips_clear=0, sq_engine_cmp=0.0, sq_engine_aux=0.0, sq_aux_cmp=0.0, first=[])
    for bseed in (batches if batches is not None else [int(x) for x in g["meta/batches"]]):
        b = synth.synth_batch(bseed, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
        sel0, acc0 = full_hook_state(b["idx_ulb"])
        batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
        for it in [int(i) for i in g["meta/its"]]:
            p = "g%g/b%d/it%d" % (gain, bseed, it)
            K = int(g[p + "/K"])
            alg.rewarder.load_state_dict(rew0)
            alg.rewarder_optimizer.load_state_dict(FlatAdamFresh(alg))
            alg.max_reward.fill_(-float("inf"))
            h.selected_label = torch.from_numpy(sel0.copy())
            h.classwise_acc = torch.from_numpy(acc0.copy()).to(DEV)
            alg.model.grad.zero_()
            alg.it = it
            alg.optimizer.sched_step = it
            alg.inject_droppath = [torch.from_numpy(synth.synth_droppath(int(g[p + "/dp_seed0"]) + k, V.drop_path_probs(cfg), Bl + 2 * Bu))
                                   for k in range(K + 1)]
            alg.trace = {}
            alg.train_step(**batch)
            torch.cuda.synchronize()
            want, refp, thr, gap, wl = (cmp[p + "/" + k_] for k_ in ("masks", "mask_probs", "mask_thr", "label_gap", "pseudo_label"))
            masks = np.stack([m.cpu().numpy() for m in alg.trace["masks"]])
            mpv = alg.trace["max_probs"].cpu().numpy().reshape(want.shape)
            lab = alg.trace["pseudo"].cpu().numpy().reshape(want.shape)
            same = lab == wl
            devs = np.abs(mpv - refp)
            st["sq_engine_cmp"] += float(((mpv - refp).astype(np.float64) ** 2).sum())
            if aux is not None:
                ap = aux[p + "/mask_probs"]
                st["sq_engine_aux"] += float(((mpv - ap).astype(np.float64) ** 2).sum())
                st["sq_aux_cmp"] += float(((ap - refp).astype(np.float64) ** 2).sum())
            room = np.minimum(np.abs(refp - thr), np.abs(refp - tr["p_cutoff"]))
            diff = (masks != want) | ~same
            st["steps"] += 1; st["rows"] += want.size
            st["flipped_rows"] += int((masks != want).sum()); st["label_mismatch_rows"] += int((~same).sum())
            if same.any():
                st["max_dev_same_label"] = max(st["max_dev_same_label"], float(devs[same].max()))
            st["rows_at_risk"] += int(((devs >= room) | (devs >= 0.5 * gap)).sum())
            if diff.any():
                st["steps_with_a_difference"] += 1
                k0 = int(np.argmax(diff.any(axis=1)))                 # first pass with a differing row: its state was still the reference's
                for r in np.nonzero(diff[k0])[0]:
                    if not same[k0, r]:
                        st["first_label_flips"] += 1
                        st["first_label_flip_max_gap"] = max(st["first_label_flip_max_gap"], float(gap[k0, r]))
                    else:
                        st["first_mask_flips"] += 1
                        st["first_mask_flips_inside_their_room"] += int(devs[k0, r] < room[k0, r])
                    st["first"].append((bseed, it, k0, int(r), bool(same[k0, r]), float(refp[k0, r]), float(mpv[k0, r]), float(thr[k0, r]), float(gap[k0, r])))
                st["downstream_rows"] += int(diff[k0 + 1:].sum())
            sel = h.selected_label.cpu().numpy()[b["idx_ulb"]]
            st["table_entries"] += sel.size; st["table_mismatches"] += int((sel != cmp[p + "/sel_after_batch"]).sum())
            if K and cmp is g:
                r = alg.trace["reward"].cpu().numpy().reshape(K, Bu)
                rg, m2g = g[p + "/reward"], g[p + "/mask2"]
                m2 = alg.trace["mask2"].cpu().numpy().reshape(K, Bu)
                # a row's mask2 = reward >= pass mean is CLEAR when the reference's reward is further from its pass mean than twice the largest
                # reward deviation of that pass; only passes whose pseudo labels (the rewarder's input) are the reference's are compared
                same_labels = same[1:].all(axis=1)
                clear = (np.abs(rg - rg.mean(axis=1, keepdims=True)) > 2.0 * np.abs(r - rg).max(axis=1, keepdims=True) + 1e-6) & same_labels[:, None]
                st["mask2_rows"] += int(same_labels.sum()) * Bu
                st["mask2_flips"] += int(((m2 != m2g) & same_labels[:, None]).sum())
                st["mask2_flips_clear"] += int(((m2 != m2g) & clear).sum())
    st["flip_rate"] = st["flipped_rows"] / max(st["rows"], 1)
    st["label_mismatch_rate"] = st["label_mismatch_rows"] / max(st["rows"], 1)
    return st


def FlatAdamFresh(alg):
    """A zeroed state dict of the rewarder's Adam (the stage-1 step at it = 1000 advances it)."""
    sd = alg.rewarder_optimizer.state_dict()
    return dict(m=torch.zeros_like(sd["m"]), v=torch.zeros_like(sd["v"]), steps=0)


# What the sweep measured on MI355X (profiles/r06_mask_identity.txt) and the bounds asserted (about twice the measurement).  gain 24: 33 of 3 840
# mask rows differ (0.86 %), 34 pseudo labels; gain 1 (stock classifier: near-uniform probabilities, top-two gaps of 1e-4): 25 rows (0.65 %), 41
# labels -- an argmax over 100 probabilities of ~0.03 each is decided by gaps the bf16 operands reach; the thresholds themselves are never the
# reason there (max-prob deviation 4.4e-4).
MASK_IDENTITY_BOUNDS = {24.0: dict(flip_rate=0.018, label_mismatch_rate=0.018, max_dev_same_label=0.12, label_gap=0.30),
                        1.0: dict(flip_rate=0.014, label_mismatch_rate=0.022, max_dev_same_label=1e-3, label_gap=6e-4)}


@pytest.mark.parametrize("gain", [24.0, 1.0])
def test_end_to_end_mask_identity_over_the_reference_sweep(golden, gain):
    """north_star: "identical pseudo-label selection masks".  The full-size trace test pins ONE batch that was screened for keeping every row
    inside its room; this test MEASURES the end-to-end decision flip rate of the bf16-operand engine against the fp32 reference over all 48
    batches of the sweep that batch came from (96 steps, 3 840 thresholded rows per gain) and asserts (i) flip and label-mismatch rates under
    the bounds written above, (ii) in every step the FIRST differing rows are explained by operand rounding: a same-label mask flip only where
    the max-prob deviation reached the row's distance to its threshold, a label flip only where the reference's top-two gap is within the
    engine's class-probability deviation, (iii) the table differs only in steps with such a row, (iv) mask2 never differs on a row whose
    reward is clear of its pass mean."""
    import json
    import os
    g = golden("srflexmatch_full_sweep")
    st = measure_mask_identity(g, gain)
    line = "MASK_IDENTITY gain %g: %s" % (gain, json.dumps({k: v for k, v in st.items() if k != "first"}))
    print(line)
    try:
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r06_mask_identity_g%g.json" % gain), "w") as f:
            json.dump(st, f)
    except OSError:
        pass
    bd = MASK_IDENTITY_BOUNDS[gain]
    assert st["steps"] == 96 and st["rows"] == 48 * (8 + 72)
    assert st["first_mask_flips_inside_their_room"] == 0, [x for x in st["first"] if x[4]]
    assert st["first_label_flip_max_gap"] <= bd["label_gap"], line
    assert st["mask2_flips_clear"] == 0
    if st["steps_with_a_difference"] == 0:
        assert st["table_mismatches"] == 0
    assert st["table_mismatches"] <= st["flipped_rows"] + st["label_mismatch_rows"]
    assert st["flip_rate"] <= bd["flip_rate"] and st["label_mismatch_rate"] <= bd["label_mismatch_rate"], line
    assert st["max_dev_same_label"] <= bd["max_dev_same_label"], line


def test_engine_deviates_from_the_reference_as_a_cpu_model_of_its_rounding_does(golden):
    """Is the 0.9 % above operand ROUNDING or kernel ERROR?  oracle.vit_ref.vit_forward_engine_rounding is a CPU model of the engine's rounding
    points (bf16 GEMM operands, q / k / v, probabilities, branch outputs; fp32 everything else) that shares none of the engine's code;
    srflexmatch_full_sweep_emu.npz holds ITS max-probs and decisions on the same 96 steps (gain 24).  The model cannot reproduce the engine's
    numbers row by row -- which way an operand rounds depends on its value to a few 1e-4 relative, so from the second block on two
    implementations of the same rounding points draw (nearly) independent samples of the same rounding noise (tools/rounding_model_probe.py:
    logits rel-L2 engine vs fp32 6.4e-3, model vs fp32 6.4e-3, engine vs model 4.7e-3) -- but it predicts its STATISTICS: an engine with
    arithmetic errors beyond its rounding points would sit further from the reference than the model does.  Asserted: the engine's rms
    max-prob deviation from the reference is within 25 % of the model's, its mask / label differences are no more frequent than the model's
    (x 1.5 + 5: both are counts of ~30), and engine and model are closer to each other than either is to the reference (the shared part)."""
    import json
    g, e = golden("srflexmatch_full_sweep"), golden("srflexmatch_full_sweep_emu")
    flips_m, labels_m, dev_m, rows = 0, 0, 0.0, 0
    for b in g["meta/batches"]:
        for it in g["meta/its"]:
            p = "g24/b%d/it%d/" % (b, it)
            dev_m = max(dev_m, float(np.abs(e[p + "mask_probs"] - g[p + "mask_probs"]).max()))
            flips_m += int((e[p + "masks"] != g[p + "masks"]).sum()); labels_m += int((e[p + "pseudo_label"] != g[p + "pseudo_label"]).sum())
            rows += g[p + "masks"].size
    st = measure_mask_identity(g, 24.0, aux=e)
    rms = {k: (st[k] / rows) ** 0.5 for k in ("sq_engine_cmp", "sq_engine_aux", "sq_aux_cmp")}
    line = "ROUNDING_MODEL gain 24: rms max-prob deviation engine-reference %.4f, model-reference %.4f, engine-model %.4f; mask rows differing from the " \
           "reference: engine %d, model %d; labels: engine %d, model %d; max deviation engine %.3f (same label), model %.3f" % (
               rms["sq_engine_cmp"], rms["sq_aux_cmp"], rms["sq_engine_aux"], st["flipped_rows"], flips_m, st["label_mismatch_rows"], labels_m,
               st["max_dev_same_label"], dev_m)
    print(line)
    try:
        import os
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r06_rounding_model.txt"), "w") as f:
            f.write(line + "\n" + json.dumps({k: v for k, v in st.items() if k != "first"}) + "\n")
    except OSError:
        pass
    assert rows == 3840 and st["rows"] == rows
    assert rms["sq_engine_cmp"] <= 1.25 * rms["sq_aux_cmp"], line
    assert st["flipped_rows"] <= 1.5 * flips_m + 5 and st["label_mismatch_rows"] <= 1.5 * labels_m + 5, line
    assert rms["sq_engine_aux"] <= max(rms["sq_engine_cmp"], rms["sq_aux_cmp"]), line


def test_elide_unread_rows_changes_no_result():
    """Opt-in ``elide_unread_rows`` (never the default): the (pass, image) rows nothing reads are not computed.  Rows are independent in the
    ViT engine, so every mask, every logit that IS read, the losses and the updated parameters must equal those of the full step."""
    NSa = dict(algorithm="srflexmatch", num_classes=100, num_train_iter=204800, ulb_dest_len=50000, start_timing=20000, feature_dim=384,
               num_warmup_iter=5120)
    b = synth.synth_batch(101, 8, 8, 32, 100, 50000)
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    dps = [torch.from_numpy(synth.synth_droppath(700 + k, V.drop_path_probs(cfg), 24)) for k in range(9)]
    res = []
    for elide in (False, True):
        alg = get_algorithm(make_args(**NSa), vit.vit_small_patch2_32)
        alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
        alg.it = 30000
        alg.optimizer.sched_step = alg.it
        alg.elide_unread_rows = elide
        alg.inject_droppath = dps
        alg.trace = {}
        p_init = alg.model.flat.clone()
        out, log = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))
        alg.call_hook("after_train_step")
        torch.cuda.synchronize()
        res.append((alg, log, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in alg.trace.items()}, alg.model.flat.clone()))
    (a0, l0, t0, p0), (a1, l1, t1, p1) = res
    K, nl, nu = t0["K"], 8, 8
    assert K == 8 and t1["K"] == 8
    plan = a1._plans[(nl, nu, K, True, True)]
    assert plan.inf_cols.numel() + plan.rest_cols.numel() + plan.grad_cols.numel() == 216 - K * nl - (K - 1) * nu
    assert all(torch.equal(x, y) for x, y in zip(t0["masks"], t1["masks"]))
    assert torch.equal(t0["pseudo"], t1["pseudo"]) and torch.equal(t0["max_probs"], t1["max_probs"])
    assert torch.equal(t0["reward"], t1["reward"]) and torch.equal(t0["mask2"], t1["mask2"])
    L0, L1 = t0["logits"], t1["logits"]
    assert torch.equal(L0[:, nl:nl + nu], L1[:, nl:nl + nu])                 # weak rows of every pass
    assert torch.equal(L0[0], L1[0]) and torch.equal(L0[K, nl + nu:], L1[K, nl + nu:])     # pass 0 complete; strong rows of the last pass
    assert torch.equal(t0["feats"][0], t1["feats"][0])
    for k_ in ("sup_loss", "unsup_loss", "total_loss", "util_ratio"):
        assert float(l0["train/" + k_]) == float(l1["train/" + k_]), k_
    # fp32 atomics in the weight-gradient sums are unordered: a first-moment-free AdamW step is ~lr * sign(g), so a gradient within round-off
    # of zero may flip -- at most twice the largest update of the step
    assert float((p0 - p1).abs().max()) <= 2.1 * float((p0 - p_init).abs().max())
    assert float((p0 - p1).abs().mean()) <= 1e-3 * float((p0 - p_init).abs().mean())
    h0, h1 = a0.hooks_dict["MaskingHook"], a1.hooks_dict["MaskingHook"]
    assert torch.equal(h0.selected_label, h1.selected_label) and torch.equal(h0.classwise_acc, h1.classwise_acc)


def test_every_pass_image_row_is_computed(monkeypatch):
    """Every (pass, image) row of the reference's 1 + K forwards is computed -- the rows nothing reads included (the deferred launch train on
    the second stream; they were once dropped by accident): the step's logits table is finite everywhere and the unread rows hold what a direct
    forward of those (pass, image) columns gives, bit for bit."""
    NSa = dict(algorithm="srflexmatch", num_classes=100, num_train_iter=204800, ulb_dest_len=50000, start_timing=20000, feature_dim=384,
               num_warmup_iter=5120)
    b = synth.synth_batch(103, 8, 8, 32, 100, 50000)
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    dps = [torch.from_numpy(synth.synth_droppath(500 + k, V.drop_path_probs(cfg), 24)) for k in range(9)]
    monkeypatch.setattr(vit, "_FUSED_MLP_MIN_ROWS", 1024)   # the direct 8-image forwards below on the kernels of the big launches
    alg = get_algorithm(make_args(**NSa), vit.vit_small_patch2_32)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.it = 30000
    alg.optimizer.sched_step = alg.it
    alg.inject_droppath = dps
    alg.trace = {}
    alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))
    torch.cuda.synchronize()
    K, nl, nu = alg.trace["K"], 8, 8
    L0 = alg.trace["logits"].clone()
    assert torch.isfinite(L0).all() and L0.shape == (K + 1, 24, 100)
    # the unread rows: strong rows of pass 3, labelled rows of pass 5
    pb = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
    imgs = torch.cat((pb["x_lb"], pb["x_ulb_w"], pb["x_ulb_s"])).contiguous()
    for k, rows in ((3, list(range(nl + nu, 24))), (5, list(range(nl)))):
        idx = torch.tensor(rows, dtype=torch.int32, device=DEV)
        dpk = dps[k][:, :, rows].contiguous().to(DEV)
        lg, _, _ = alg.model.forward_features(imgs, idx, dpk, save=False)
        torch.cuda.synchronize()
        assert torch.equal(lg.cpu(), L0[k, rows].cpu()), k


def test_deferred_share_autotune_picks_a_candidate_and_changes_no_result():
    """srflexmatch._DeferTuner: in the first steps of a regime every candidate share of deferred inference rows runs WARM + TIMED real training
    steps; afterwards the median-fastest one is kept.  The split is pure scheduling: the logits / masks / losses of a step under ANY candidate equal
    those under the untuned rule bit for bit (same kernels, rows independent)."""
    from semireward_amd.algorithms import srflexmatch as SF
    assert SF._DEFER_AUTOTUNE
    NSa = dict(algorithm="srflexmatch", num_classes=100, num_train_iter=204800, ulb_dest_len=50000, start_timing=20000, feature_dim=384,
               num_warmup_iter=5120)
    b = synth.synth_batch(104, 8, 8, 32, 100, 50000)
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    dps = [torch.from_numpy(synth.synth_droppath(300 + k, V.drop_path_probs(cfg), 24)) for k in range(9)]

    def fresh():
        alg = get_algorithm(make_args(**NSa), vit.vit_small_patch2_32)
        alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
        alg.it = 30000
        alg.optimizer.sched_step = alg.it
        alg.inject_droppath = dps
        return alg
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    key = (8, 8, 8, True, False)
    # (a) every candidate split gives the bits of the untuned rule
    ref = None
    for frac in (None,) + SF._DeferTuner.CANDIDATES:
        alg = fresh()
        alg._plans[key] = alg._make_plan(8, 8, 8, defer_fraction=frac)       # pinned plan: no tuner is created for an existing key
        alg.trace = {}
        out, log = alg.train_step(**alg.process_batch(**batch))
        torch.cuda.synchronize()
        assert key not in alg._tuners
        got = (alg.trace["logits"].clone(), [m.clone() for m in alg.trace["masks"]], alg.trace["reward"].clone(), float(log["train/total_loss"]),
               int(alg._plans[key].rest_cols.numel()))
        if ref is None:
            ref = got
        else:
            assert torch.equal(got[0], ref[0]) and all(torch.equal(x, y) for x, y in zip(got[1], ref[1])) and torch.equal(got[2], ref[2])
            assert got[3] == ref[3]
    # (b) the tuner runs its schedule over real steps and settles
    alg = fresh()
    alg.inject_droppath = None
    per = SF._DeferTuner.WARM + SF._DeferTuner.TIMED
    seen = []
    for i in range(per * (len(SF._DeferTuner.CANDIDATES) + 2) + 3):
        alg.out_dict, alg.log_dict = alg.train_step(**alg.process_batch(**batch))
        alg.call_hook("after_train_step")
        alg.it += 1
        seen.append(int(alg._plans[key].rest_cols.numel()))
        if i == 0:
            ncand = len(alg._tuners[key][1])
            assert ncand >= 4
    assert key not in alg._tuners and key in alg.defer_report
    rep = alg.defer_report[key]
    assert rep["deferred_images"] == seen[-1] == seen[-2] and len(set(seen[:per * ncand])) == ncand       # coarse pass: every candidate ran
    assert ncand <= len(rep["ms_per_step"]) <= ncand + 2 and all(0.5 < v < 1000.0 for v in rep["ms_per_step"].values())
    best = min(rep["ms_per_step"], key=rep["ms_per_step"].get)
    assert abs(float(best) - rep["chosen"]) < 1e-3
    assert np.isfinite(float(alg.log_dict["train/total_loss"]))
