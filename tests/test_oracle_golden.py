"""Pins the CPU oracle (oracle/) against outputs of the reference itself (tests/golden/*.npz,
written by oracle/gen_golden.py in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import check_samp
from oracle import hooks_ref as H
from oracle import optim_ref as O
from oracle import semireward_ref as S
from oracle import srflexmatch_ref as SF
from oracle import vit_ref as V
from oracle.gen_golden import TRACE, TRACE_C100, TRACE_FIX
from semireward_amd.utils import synth

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
TP = lambda d: {k: T(v) for k, v in d.items()}  # noqa: E731
# cross_attention_fc.bias cancels inside softmax(dim=0): its analytic gradient is exactly 0, the
# reference's autograd leaves ~1e-10 round-off there and Adam (m/sqrt(v)) turns that noise into
# +-lr steps.  The value never influences any output, so it is excluded from parameter parity.
NOISE_FREE_KEYS = tuple(k for k in S.REWARDER_KEYS if k != "cross_attention_fc.bias")


@pytest.mark.parametrize("tag", ["f384_b8", "f128_b64", "f768_b16_c200"])
def test_rewarder_generator(golden, tag):
    g = golden("rewarder")
    Fd, C, B, seed = [int(v) for v in g[f"{tag}/meta"]]
    rp = TP(synth.synth_params(S.rewarder_shapes(Fd, C), seed))
    gp = synth.synth_params(S.generator_shapes(Fd), seed + 100)
    gp["fc_layers.6.bias"] = gp["fc_layers.6.bias"] + np.float32(2.6)
    gp = TP(gp)
    rng = np.random.Generator(np.random.PCG64(seed + 200))
    feats = T(rng.standard_normal((B, Fd)).astype(np.float32))
    labels = T(rng.integers(0, C, size=(B,), dtype=np.int64))
    r = S.rewarder_forward(rp, feats, labels)
    np.testing.assert_allclose(r.numpy(), g[f"{tag}/reward"], rtol=2e-6, atol=2e-7)
    # mask2 on the reference's own rewards must be bit-exact
    assert np.array_equal(S.reward_mask2(T(g[f"{tag}/reward"])).numpy(), g[f"{tag}/mask2"])
    go = S.generator_forward(gp, feats)
    np.testing.assert_allclose(go.numpy(), g[f"{tag}/gen_out"], rtol=2e-6, atol=1e-6)
    gl = S.generated_labels(gp, feats)
    assert np.array_equal(gl.numpy(), g[f"{tag}/gen_label"][:, 0])
    assert gl.max() > 0          # fixture exercises non-zero generated labels
    # SR update: target, losses, grads, two Adam steps
    tgt = S.cosine_target(gl, labels, C)
    assert np.array_equal(tgt.numpy(), g[f"{tag}/upd_target"])
    m = {k: torch.zeros_like(v) for k, v in rp.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in rp.items()}
    for step in (1, 2):
        reward, grads, lg, lr_ = S.rewarder_update_grads(rp, feats, gl, tgt)
        if step == 1:
            np.testing.assert_allclose(reward.numpy(), g[f"{tag}/upd_reward"], rtol=2e-6, atol=2e-7)
            assert abs(lg - float(g[f"{tag}/generator_loss"])) < 1e-6
            assert abs(lr_ - float(g[f"{tag}/rewarder_loss"])) < 1e-6
            for k in S.REWARDER_KEYS:
                check_samp(grads[k].numpy(), g.samp(f"{tag}/grad/{k}"), 2e-4, 2e-8, k)
        S.adam_step(rp, grads, m, v, step, 5e-4)
        for k in NOISE_FREE_KEYS:
            # Adam's first steps move every weight by ~lr regardless of |g| -> atol ~ 1e-6 * lr scale
            check_samp(rp[k].numpy(), g.samp(f"{tag}/after{step}/{k}"), 1e-5, 2e-5, k)


@pytest.mark.parametrize("tag", ["c10_w", "c100_w", "c10_nw", "c100_b256", "c100_rej", "c100_rej_nw"])
def test_flexmatch_hook_bit_exact(golden, tag):
    g = golden("hooks")
    C, U, Bu, steps, warm, seed = [int(v) for v in g[f"{tag}/meta"]]
    st = H.FlexMatchState(U, C, bool(warm))
    for t in range(steps):
        probs = g[f"{tag}/probs"][t]
        m = st.masking(probs, g[f"{tag}/idx"][t], 0.95)
        assert np.array_equal(m, g[f"{tag}/mask"][t]), (tag, t)
        assert np.array_equal(st.classwise_acc.view(np.uint32), g[f"{tag}/classwise_acc"][t].view(np.uint32)), (tag, t)
        assert np.array_equal(H.pseudo_label_hard(probs), g[f"{tag}/pseudo_label"][t])
        assert np.array_equal(H.fixed_threshold_mask(probs, 0.95), g[f"{tag}/fixed_mask"][t])
    nz = np.nonzero(st.selected_label != -1)[0]
    assert np.array_equal(nz, g[f"{tag}/sel_idx"]) and np.array_equal(st.selected_label[nz], g[f"{tag}/sel_val"])
    if tag.startswith("c10_") or tag.startswith("c100_rej"):     # these fixtures exercise both mask outcomes (c100_rej*: at the headline class count)
        assert g[f"{tag}/mask"].min() == 0.0 and g[f"{tag}/mask"].max() == 1.0
    if tag.startswith("c100_rej"):
        assert 0.2 < g[f"{tag}/mask"].mean() < 0.9 and g[f"{tag}/classwise_acc"][-1].max() == 1.0


@pytest.mark.parametrize("tag", ["b8_c100", "b64_c10", "b256_c100"])
def test_losses(golden, tag):
    g = golden("losses")
    lg, y = T(g[f"{tag}/logits"]), T(g[f"{tag}/y"])
    mask, mask2 = T(g[f"{tag}/mask"]), T(g[f"{tag}/mask2"])
    assert abs(float(H.ce_loss_mean(lg, y)) - float(g[f"{tag}/sup"])) < 2e-6
    assert abs(float(H.consistency_loss(lg, y, mask, mask2)) - float(g[f"{tag}/unsup"])) < 2e-6
    assert abs(float(H.consistency_loss(lg, y, mask)) - float(g[f"{tag}/unsup_mask1"])) < 2e-6
    l2 = lg.clone().requires_grad_(True)
    H.consistency_loss(l2, y, mask, mask2).backward()
    np.testing.assert_allclose(l2.grad.numpy(), g[f"{tag}/unsup_grad"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("tag,cfgd", [("tiny", V.VIT_TINY_TEST), ("small_p2_32", V.VIT_SMALL_P2_32), ("base_p16_96", V.VIT_BASE_P16_96)])
def test_vit_forward_backward(golden, tag, cfgd):
    g = golden("vit_b16_96" if tag == "base_p16_96" else "vit")
    C, B, seed = [int(v) for v in g[f"{tag}/meta"]]
    cfg = V.VitCfg(num_classes=C, **cfgd)
    P = TP(synth.synth_params(V.param_shapes(cfg), seed))
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = T(rng.standard_normal((B, 3, cfg.img_size, cfg.img_size)).astype(np.float32))
    y = T(rng.integers(0, C, size=(B,), dtype=np.int64))
    w = T(rng.random(B).astype(np.float32))
    dp = T(synth.synth_droppath(seed + 2, V.drop_path_probs(cfg), B))
    with torch.no_grad():
        o = V.vit_forward(P, x, cfg, None)
    np.testing.assert_allclose(o["logits"].numpy(), g[f"{tag}/eval_logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(o["feat"].numpy(), g[f"{tag}/eval_feat"], rtol=1e-4, atol=2e-5)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    o = V.vit_forward(Pg, x, cfg, dp)
    np.testing.assert_allclose(o["logits"].detach().numpy(), g[f"{tag}/train_logits"], rtol=1e-4, atol=2e-5)
    loss = (H.ce_loss_rows(o["logits"], y) * w).mean()
    assert abs(float(loss.detach()) - float(g[f"{tag}/loss"])) < 1e-5
    loss.backward()
    for n, _ in V.param_shapes(cfg):
        check_samp(Pg[n].grad.numpy(), g.samp(f"{tag}/grad/{n}"), 2e-3, 2e-6, n)


def test_optimizer_semantics(golden):
    g = golden("optim")
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    hp = O.vit_param_hparams(V.param_shapes(cfg), cfg.depth, 5e-4, 5e-4, 0.5)
    assert int(g["small/num_groups"]) == 28 and len(g["small/names"]) == 152
    for n, lr, wd in zip(g["small/names"], g["small/lr"], g["small/wd"]):
        assert hp[str(n)][0] == pytest.approx(float(lr), rel=1e-12) and hp[str(n)][1] == float(wd), n
    for s, f in zip(g["sched/steps"], g["sched/factor"]):
        assert O.cosine_warmup_factor(int(s), 204800, 5120) == pytest.approx(float(f), rel=1e-12, abs=1e-15)
    cfgt = V.VitCfg(num_classes=10, **V.VIT_TINY_TEST)
    P = TP(synth.synth_params(V.param_shapes(cfgt), 61))
    hpt = O.vit_param_hparams(V.param_shapes(cfgt), cfgt.depth, 5e-4, 5e-4, 0.5)
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in P.items()}
    for step in range(4):
        gr = synth.synth_params(V.param_shapes(cfgt), 70 + step)
        fac = O.cosine_warmup_factor(step, 10, 2)
        for k in P:
            O.adamw_step(P[k], T(gr[k]) * 0.1, m[k], v[k], step + 1, hpt[k][0] * fac, hpt[k][1])
    for k in P:
        check_samp(P[k].numpy(), g.samp(f"adamw_tiny/{k}"), 1e-5, 1e-7, k)


def k_bias_rows(cfg):
    """The K third of attn.qkv.bias: q.(k+b) shifts every score of a query equally, softmax cancels it,
    so its analytic gradient is exactly 0; autograd leaves round-off there and AdamW turns that into
    +-lr random steps (reference and oracle alike).  No output depends on it -> excluded from parity."""
    D = cfg.embed_dim
    return lambda idx: (idx >= D) & (idx < 2 * D)


@pytest.mark.parametrize("name,tr", [("srflexmatch_trace", TRACE), ("srflexmatch_c100_trace", TRACE_C100), ("srfixmatch_trace", TRACE_FIX)])
def test_sr_train_step_trace(golden, name, tr):
    """Whole-step control flow (SURVEY A.1-A.7, A.5 boundary iterations) against the reference traces
    (SRFlexMatch: srflexmatch.py:107-217, at 10 and at 100 classes; SRFixMatch: srfixmatch/fixmatch.py:96-205)."""
    from oracle.gen_golden import trace_vit_params
    g = golden(name)
    fix = tr["algorithm"] == "srfixmatch"
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_TINY_TEST)
    Fd = cfg.embed_dim
    orc = SF.SRFlexMatchOracle(
        cfg, TP(trace_vit_params(cfg, seed, tr.get("head_gain", 1.0), tr.get("hot_classes", 0), tr.get("cold_scale", 0.25))),
        TP(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)),
        TP(synth.synth_params(S.generator_shapes(Fd), seed + 2)),
        num_train_iter=tr["num_train_iter"], start_timing=tr["start_timing"], N_k=tr["N_k"],
        ulb_dest_len=tr["ulb_dest_len"], num_warmup_iter=tr["num_warmup_iter"], p_cutoff=tr["p_cutoff"],
        algorithm=tr["algorithm"], lr=tr.get("lr", 5e-4))
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        orc.it = it
        K = int(g[f"{p}/K"])
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
        dps = [T(synth.synth_droppath(seed + 1000 * (n + 1) + k, V.drop_path_probs(cfg), Bl + 2 * Bu)) for k in range(K + 1)]
        t = orc.train_step(T(b["x_lb"]), T(b["y_lb"]), T(b["idx_ulb"]), T(b["x_ulb_w"]), T(b["x_ulb_s"]), dps)
        assert t["K"] == K
        masks = np.stack([q["mask"].numpy() for q in t["passes"]])
        assert np.array_equal(masks, g[f"{p}/masks"]), p
        if not fix:
            accs = np.stack([q["classwise_acc"] for q in t["passes"]])
            assert np.array_equal(accs.view(np.uint32), g[f"{p}/accs"].view(np.uint32)), p
            # per pass: the pseudo labels the hooks and the rewarder saw, the rewards of the K scoring calls (:99) and the mask2 the loss received (:100-102)
            assert np.array_equal(np.stack([q["pseudo_label"].numpy() for q in t["passes"]]), g[f"{p}/pseudo_label"]), p
            if K:
                np.testing.assert_allclose(np.stack([q["reward"].numpy().reshape(-1) for q in t["passes"][1:]]), g[f"{p}/reward"], rtol=2e-5, atol=2e-6)
                assert np.array_equal(np.stack([q["mask2"].numpy() for q in t["passes"][1:]]), g[f"{p}/mask2"]), p
        for k_ in ("sup_loss", "unsup_loss", "total_loss", "util_ratio"):
            assert t[k_] == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=2e-5, abs=2e-6), (p, k_)
        assert t["lr_factor"] == pytest.approx(float(g[f"{p}/lr_factor"]), rel=1e-9, abs=1e-12), p
        assert int("sr_stage" in t) == int(g[f"{p}/rewarder_updated"]), p
        for k_ in ("x_lb", "x_ulb_w", "x_ulb_s"):
            np.testing.assert_allclose(t["feat"][k_].numpy(), g[f"{p}/feat/{k_}"], rtol=1e-4, atol=2e-5)
        for nme, _ in V.param_shapes(cfg):
            check_samp(t["grads"][nme].numpy(), g.samp(f"{p}/grad/{nme}"), 2e-3, 2e-6, f"{p} grad {nme}")
            check_samp(orc.P[nme].numpy(), g.samp(f"{p}/param/{nme}"), 1e-4, 2e-6, f"{p} param {nme}",
                       exclude=k_bias_rows(cfg) if nme.endswith("attn.qkv.bias") else None)
        for k_ in NOISE_FREE_KEYS:
            check_samp(orc.R[k_].numpy(), g.samp(f"{p}/rewarder/{k_}"), 1e-4, 3e-5, f"{p} rewarder {k_}")
        mr = float(g[f"{p}/max_reward"])
        assert (np.isinf(mr) and np.isinf(orc.max_reward)) or orc.max_reward == pytest.approx(mr, rel=1e-5), p
        if not fix:
            nz = np.nonzero(orc.hook.selected_label != -1)[0]
            assert np.array_equal(nz, g[f"{p}/sel_idx"]) and np.array_equal(orc.hook.selected_label[nz], g[f"{p}/sel_val"])
    allm = np.concatenate([g[f"it{it}/masks"].ravel() for it in tr["its"]])
    if fix:      # fixture exercises both mask outcomes
        assert 0.05 < allm.mean() < 0.95
    else:
        # the FlexMatch fixtures are NOT degenerate: rows are rejected (mask 0) in several iterations, labels are selected from the third
        # iteration on, classwise_acc is non-zero, the convex threshold p_cutoff * acc / (2 - acc) is non-zero where it rejects, mask2 has both
        # values, and no max-prob the reference thresholds sits closer than tr['min_margin'] to the threshold it is compared with
        assert 0.2 < allm.mean() < 0.9
        third = tr["its"][2]
        assert len(g[f"it{third}/sel_idx"]) > 0 and g[f"it{third}/accs"].max() > 0
        assert sum(1 for it in tr["its"] if g[f"it{it}/masks"].min() == 0.0) >= 4
        m2 = np.concatenate([g[f"it{it}/mask2"].ravel() for it in tr["its"] if int(g[f"it{it}/K"])])
        assert 0.1 < m2.mean() < 0.9
        margin = min(min(float(np.abs(g[f"it{it}/mask_probs"] - g[f"it{it}/mask_thr"]).min()), float(np.abs(g[f"it{it}/mask_probs"] - tr["p_cutoff"]).min()))
                     for it in tr["its"])
        assert margin >= tr["min_margin"], margin
        rej = [float(g[f"it{it}/mask_thr"][g[f"it{it}/masks"] == 0.0].min()) for it in tr["its"] if g[f"it{it}/masks"].min() == 0.0]
        assert min(rej) > 0.0


def test_sr_decay_schedule():
    """NS config K = 11 / 10 / 9 / 8 bands."""
    N = 204800
    assert H.sr_decay(N, 20001) == 11 and H.sr_decay(N, 20480) == 11
    assert H.sr_decay(N, 22755) == 10 and H.sr_decay(N, 25600) == 9
    assert H.sr_decay(N, 25601) == 8 and H.sr_decay(N, 204799) == 8


def test_srpseudolabel_trace(golden):
    """SRPseudoLabel (srpseudolabel.py:59-201) against the reference trace: separate lb / ulb forwards, logits-fed fixed
    threshold, self-training loss on the weak view, unsup warm-up factor."""
    from oracle.gen_golden import TRACE_PL as tr
    from oracle.srpseudolabel_ref import SRPseudoLabelOracle
    g = golden("srpseudolabel_trace")
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_TINY_TEST)
    Fd = cfg.embed_dim
    orc = SRPseudoLabelOracle(
        cfg, TP(synth.synth_params(V.param_shapes(cfg), seed)), TP(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)),
        TP(synth.synth_params(S.generator_shapes(Fd), seed + 2)), num_train_iter=tr["num_train_iter"],
        start_timing=tr["start_timing"], N_k=tr["N_k"], ulb_dest_len=tr["ulb_dest_len"], num_warmup_iter=tr["num_warmup_iter"],
        p_cutoff=tr["p_cutoff"], unsup_warm_up=tr["unsup_warm_up"])
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        orc.it = it
        K = int(g[f"{p}/K"])
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
        dpl = T(synth.synth_droppath(seed + 1000 * (n + 1), V.drop_path_probs(cfg), Bl))
        dpu = [T(synth.synth_droppath(seed + 1000 * (n + 1) + 1 + k, V.drop_path_probs(cfg), Bu)) for k in range(K + 1)]
        t = orc.train_step(T(b["x_lb"]), T(b["y_lb"]), T(b["x_ulb_w"]), [(dpl, dpu[0])] + dpu[1:])
        assert t["K"] == K
        assert np.array_equal(np.stack([q["mask"].numpy() for q in t["passes"]]), g[f"{p}/masks"]), p
        for k_ in ("sup_loss", "unsup_loss", "total_loss", "util_ratio"):
            assert t[k_] == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=2e-5, abs=2e-6), (p, k_)
        assert int("sr_stage" in t) == int(g[f"{p}/rewarder_updated"]), p
        for nme, _ in V.param_shapes(cfg):
            check_samp(t["grads"][nme].numpy(), g.samp(f"{p}/grad/{nme}"), 2e-3, 2e-6, f"{p} grad {nme}")
            check_samp(orc.P[nme].numpy(), g.samp(f"{p}/param/{nme}"), 1e-4, 2e-6, f"{p} param {nme}",
                       exclude=k_bias_rows(cfg) if nme.endswith("attn.qkv.bias") else None)
        for k_ in NOISE_FREE_KEYS:
            check_samp(orc.R[k_].numpy(), g.samp(f"{p}/rewarder/{k_}"), 1e-4, 3e-5, f"{p} rewarder {k_}")
    allm = np.concatenate([g[f"it{it}/masks"].ravel() for it in tr["its"]])
    assert 0.05 < allm.mean() < 0.95


@pytest.mark.parametrize("tag", ["c10_q", "c100_mean", "c100_q"])
def test_freematch_hook_and_entropy(golden, tag):
    """FreeMatchThresholdingHook state (freematch/utils.py:24-66) bit for bit + entropy_loss (srfreematch.py:16-44) value / gradient."""
    g = golden("freematch_hook")
    C, Bu, steps, uq, clip, seed = [int(v) for v in g[f"{tag}/meta"]]
    st = H.FreeMatchState(C, float(g[f"{tag}/momentum"]), bool(uq), bool(clip))
    for t in range(steps):
        m = st.masking(T(g[f"{tag}/probs"][t]))
        assert np.array_equal(m.numpy(), g[f"{tag}/mask"][t]), (tag, t)
        assert np.float32(st.time_p).view(np.uint32) == g[f"{tag}/time_p"][t].view(np.uint32), (tag, t)
        assert np.array_equal(st.p_model.numpy().view(np.uint32), g[f"{tag}/p_model"][t].view(np.uint32))
        assert np.array_equal(st.label_hist.numpy().view(np.uint32), g[f"{tag}/label_hist"][t].view(np.uint32))
        lg = T(g[f"{tag}/logits_s"][t]).requires_grad_(True)
        if float(m.sum()) > 0:
            e = H.freematch_entropy_loss(m, lg, st.p_model, st.label_hist)
            e.backward()
            assert float(e.detach()) == pytest.approx(float(g[f"{tag}/ent"][t]), rel=1e-6, abs=1e-7)
            np.testing.assert_allclose(lg.grad.numpy(), g[f"{tag}/ent_grad"][t], rtol=1e-5, atol=1e-9)


def test_srfreematch_trace(golden):
    from oracle.gen_golden import TRACE_FREE as tr
    from oracle.srfreematch_ref import SRFreeMatchOracle
    g = golden("srfreematch_trace")
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_TINY_TEST)
    Fd = cfg.embed_dim
    orc = SRFreeMatchOracle(
        cfg, TP(synth.synth_params(V.param_shapes(cfg), seed)), TP(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)),
        TP(synth.synth_params(S.generator_shapes(Fd), seed + 2)), num_train_iter=tr["num_train_iter"],
        start_timing=tr["start_timing"], N_k=tr["N_k"], ulb_dest_len=tr["ulb_dest_len"], num_warmup_iter=tr["num_warmup_iter"],
        ema_p=tr["ema_p"], use_quantile=tr["use_quantile"], clip_thresh=tr["clip_thresh"], lambda_e=tr["ent_loss_ratio"])
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        orc.it = it
        K = int(g[f"{p}/K"])
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
        dps = [T(synth.synth_droppath(seed + 1000 * (n + 1) + k, V.drop_path_probs(cfg), Bl + 2 * Bu)) for k in range(K + 1)]
        t = orc.train_step(T(b["x_lb"]), T(b["y_lb"]), T(b["x_ulb_w"]), T(b["x_ulb_s"]), dps)
        assert t["K"] == K
        assert np.array_equal(np.stack([q["mask"].numpy() for q in t["passes"]]), g[f"{p}/masks"]), p
        for k_ in ("sup_loss", "unsup_loss", "total_loss", "util_ratio"):
            assert t[k_] == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=2e-5, abs=2e-6), (p, k_)
        assert float(orc.fm.time_p) == pytest.approx(float(g[f"{p}/time_p"]), rel=1e-6)
        np.testing.assert_allclose(orc.fm.p_model.numpy(), g[f"{p}/p_model"], rtol=1e-6)
        np.testing.assert_allclose(orc.fm.label_hist.numpy(), g[f"{p}/label_hist"], rtol=1e-6)
        for nme, _ in V.param_shapes(cfg):
            check_samp(t["grads"][nme].numpy(), g.samp(f"{p}/grad/{nme}"), 2e-3, 2e-6, f"{p} grad {nme}")


@pytest.mark.parametrize("tag", ["c10_uniform", "c100_model", "c10_model_s3"])
def test_softmatch_and_distalign_hooks(golden, tag):
    """DistAlignEMAHook (dist_align.py:26-56) and SoftMatchWeightingHook (srsoftmatch/utils.py:32-76) sequences of the reference:
    EMA state bit for bit, aligned probabilities / weights to fp32 round-off."""
    g = golden("softmatch_hook")
    C, Bu, Bl, steps, ns, model_t, seed = [int(v) for v in g[f"{tag}/meta"]]
    m = float(g[f"{tag}/momentum"])
    da, sm = H.DistAlignState(C, m, "model" if model_t else "uniform"), H.SoftMatchState(C, ns, m)
    for t in range(steps):
        pu, plb = torch.softmax(T(g[f"{tag}/logits_ulb"][t]), dim=-1), torch.softmax(T(g[f"{tag}/logits_lb"][t]), dim=-1)
        al = da.dist_align(pu, plb)
        ma = sm.masking(al)
        mp_ = sm.masking(pu)
        assert np.array_equal(da.p_model.numpy().view(np.uint32), g[f"{tag}/p_model"][t].view(np.uint32)), (tag, t)
        assert np.array_equal(da.p_target.numpy().view(np.uint32), g[f"{tag}/p_target"][t].view(np.uint32)), (tag, t)
        assert np.float32(sm.mu).view(np.uint32) == g[f"{tag}/mu"][t].view(np.uint32), (tag, t)
        assert np.float32(sm.var).view(np.uint32) == g[f"{tag}/var"][t].view(np.uint32), (tag, t)
        assert np.array_equal(al.numpy(), g[f"{tag}/aligned"][t]) and np.array_equal(ma.numpy(), g[f"{tag}/mask_a"][t])
        assert np.array_equal(mp_.numpy(), g[f"{tag}/mask_p"][t])
    allm = np.concatenate([g[f"{tag}/mask_a"].ravel(), g[f"{tag}/mask_p"].ravel()])
    assert allm.min() >= 0.0 and allm.max() <= 1.0 and 0.05 < allm.mean() < 0.999 and (allm < 0.9).any()     # the weight is exercised


def test_srsoftmatch_trace(golden):
    from oracle.gen_golden import TRACE_SOFT as tr
    from oracle.srsoftmatch_ref import SRSoftMatchOracle
    g = golden("srsoftmatch_trace")
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    cfg = V.VitCfg(num_classes=C, **V.VIT_TINY_TEST)
    Fd = cfg.embed_dim
    orc = SRSoftMatchOracle(
        cfg, TP(synth.synth_params(V.param_shapes(cfg), seed)), TP(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)),
        TP(synth.synth_params(S.generator_shapes(Fd), seed + 2)), num_train_iter=tr["num_train_iter"],
        start_timing=tr["start_timing"], N_k=tr["N_k"], ulb_dest_len=tr["ulb_dest_len"], num_warmup_iter=tr["num_warmup_iter"],
        ema_p=tr["ema_p"], n_sigma=tr["n_sigma"], dist_uniform=tr["dist_uniform"])
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        orc.it = it
        K = int(g[f"{p}/K"])
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
        dps = [T(synth.synth_droppath(seed + 1000 * (n + 1) + k, V.drop_path_probs(cfg), Bl + 2 * Bu)) for k in range(K + 1)]
        t = orc.train_step(T(b["x_lb"]), T(b["y_lb"]), T(b["x_ulb_w"]), T(b["x_ulb_s"]), dps)
        assert t["K"] == K
        np.testing.assert_allclose(np.stack([q["mask"].numpy() for q in t["passes"]]), g[f"{p}/masks"], rtol=1e-5, atol=1e-7)
        for k_ in ("sup_loss", "unsup_loss", "total_loss", "util_ratio"):
            assert t[k_] == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=2e-5, abs=2e-6), (p, k_)
        assert float(orc.sm.mu) == pytest.approx(float(g[f"{p}/mu"]), rel=1e-6) and float(orc.sm.var) == pytest.approx(float(g[f"{p}/var"]), rel=1e-6)
        np.testing.assert_allclose(orc.da.p_model.numpy(), g[f"{p}/p_model"], rtol=1e-6)
        np.testing.assert_allclose(orc.da.p_target.numpy(), g[f"{p}/p_target"], rtol=1e-6)
        for nme, _ in V.param_shapes(cfg):
            check_samp(t["grads"][nme].numpy(), g.samp(f"{p}/grad/{nme}"), 2e-3, 2e-6, f"{p} grad {nme}")


@pytest.mark.parametrize("tag", ["tiny", "wrn_28_2"])
def test_wrn_oracle_matches_reference(golden, tag):
    """oracle/wrn_ref.py against the reference WideResNet (wrn.py): eval / train / frozen-BN forwards, running statistics, gradients."""
    from oracle import wrn_ref as W
    from oracle.gen_golden import synth_wrn_params
    g = golden("wrn")
    C, B, HW, seed = [int(v) for v in g[f"{tag}/meta"]]
    cfg = W.WrnCfg(num_classes=C, **(W.WRN_TINY_TEST if tag == "tiny" else W.WRN_28_2))
    P = {k: T(v).requires_grad_(True) for k, v in synth_wrn_params(cfg, seed).items()}
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = T(rng.standard_normal((B, 3, HW, HW)).astype(np.float32)); x2 = T(rng.standard_normal((B, 3, HW, HW)).astype(np.float32))
    y = T(rng.integers(0, C, size=(B,), dtype=np.int64)); w = T(rng.random(B).astype(np.float32))
    BUF = {k[len(tag) + 6:]: T(g[k]).clone() for k in g.keys() if k.startswith(f"{tag}/buf0/")}
    with torch.no_grad():
        o = W.wrn_forward(P, BUF, x, cfg, train=False)
    np.testing.assert_allclose(o["logits"].numpy(), g[f"{tag}/eval_logits"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(o["feat"].numpy(), g[f"{tag}/eval_feat"], rtol=2e-4, atol=2e-5)
    o = W.wrn_forward(P, BUF, x, cfg, train=True, update_stats=True)
    np.testing.assert_allclose(o["logits"].detach().numpy(), g[f"{tag}/train_logits"], rtol=2e-4, atol=2e-5)
    for k in BUF:
        np.testing.assert_allclose(BUF[k].numpy(), g[f"{tag}/buf1/{k}"], rtol=1e-6, atol=1e-7)
    o2 = W.wrn_forward(P, BUF, x2, cfg, train=True, update_stats=False)
    np.testing.assert_allclose(o2["logits"].detach().numpy(), g[f"{tag}/frozen_logits"], rtol=2e-4, atol=2e-5)
    for k in BUF:
        np.testing.assert_allclose(BUF[k].numpy(), g[f"{tag}/buf1/{k}"], rtol=1e-6, atol=1e-7)          # untouched by the frozen forward
    ce = torch.nn.functional.cross_entropy
    loss = (ce(o["logits"], y, reduction="none") * w).mean() + 0.5 * (ce(o2["logits"], y, reduction="none") * w).mean()
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g[f"{tag}/loss"]), rel=1e-5)
    for n, _ in W.param_shapes(cfg):
        gr = P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])
        check_samp(gr.numpy(), g.samp(f"{tag}/grad/{n}"), 2e-3, 1e-6, f"{tag} grad {n}")


def test_srpseudolabel_wrn_trace(golden):
    """SRPseudoLabel on the WideResNet backbone with SGD (classic_cv, BASELINE.json configs[0]) against a trace of the reference itself:
    masks of every pass, losses, BatchNorm running statistics (labelled forward only), gradients, parameters after the SGD steps."""
    from oracle import wrn_ref as W
    from oracle.gen_golden import TRACE_PL_WRN as tr, synth_wrn_params
    from oracle.srpseudolabel_ref import SRPseudoLabelWrnOracle
    g = golden("srpseudolabel_wrn_trace")
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    wcfg = W.WrnCfg(num_classes=C, **W.WRN_TINY_TEST)
    Fd = W.channels(wcfg)[3]
    orc = SRPseudoLabelWrnOracle(
        wcfg, TP(synth_wrn_params(wcfg, seed)), W.init_buffers(wcfg), TP(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)),
        TP(synth.synth_params(S.generator_shapes(Fd), seed + 2)), num_train_iter=tr["num_train_iter"], start_timing=tr["start_timing"],
        N_k=tr["N_k"], p_cutoff=tr["p_cutoff"], num_warmup_iter=tr["num_warmup_iter"], unsup_warm_up=tr["unsup_warm_up"], lr=tr["lr"],
        momentum=tr["momentum"], weight_decay=tr["weight_decay"])
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        orc.it = it
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, tr["img"], C, tr["ulb_dest_len"])
        t = orc.train_step(T(b["x_lb"]), T(b["y_lb"]), T(b["x_ulb_w"]))
        assert t["K"] == int(g[f"{p}/K"])
        assert np.array_equal(np.stack([q["mask"].numpy() for q in t["passes"]]), g[f"{p}/masks"]), p
        for k_ in ("sup_loss", "unsup_loss", "total_loss", "util_ratio"):
            assert t[k_] == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=3e-5, abs=3e-6), (p, k_)
        assert t["lr_factor"] == pytest.approx(float(g[f"{p}/lr_factor"]), rel=1e-9)
        for k_, v in orc.BUF.items():
            np.testing.assert_allclose(v.numpy(), g[f"{p}/buf/{k_}"], rtol=2e-6, atol=1e-7, err_msg=f"{p} {k_}")
        for nme, _ in W.param_shapes(wcfg):
            check_samp(t["grads"][nme].numpy(), g.samp(f"{p}/grad/{nme}"), 3e-3, 3e-6, f"{p} grad {nme}")
            check_samp(orc.P[nme].numpy(), g.samp(f"{p}/param/{nme}"), 2e-4, 2e-5, f"{p} param {nme}")
    allm = np.concatenate([g[f"it{it}/masks"].ravel() for it in tr["its"]])
    assert 0.05 < allm.mean() < 0.95


@pytest.mark.parametrize("tag", ["tiny", "base"])
def test_bert_oracle_matches_reference(golden, tag):
    """oracle/bert_ref.py against the reference ClassificationBert on a random-init HF BertModel (third-party arithmetic pinned by these
    vectors): eval forward, train forward with the shared counter-based dropout masks, gradients of a weighted CE."""
    from oracle import bert_ref as BR
    g = golden("bert")
    C, B, L, seed, dseed = [int(v) for v in g[f"{tag}/meta"]]
    cfg = BR.BertCfg(num_classes=C, **(BR.BERT_TINY_TEST if tag == "tiny" else BR.BERT_BASE))
    P = {k: T(v).requires_grad_(True) for k, v in BR.synth_params(cfg, seed).items()}
    ids, mask = (T(a) for a in BR.synth_tokens(seed + 1, B, L, cfg.vocab))
    assert int(mask.sum(1).min()) < L and int(mask.sum(1).max()) == L            # ragged batch, longest row fills L
    rng = np.random.Generator(np.random.PCG64(seed + 2))
    y, w = T(rng.integers(0, C, size=(B,), dtype=np.int64)), T(rng.random(B).astype(np.float32))
    with torch.no_grad():
        o = BR.bert_forward(P, ids, mask, cfg)
    np.testing.assert_allclose(o["logits"].numpy(), g[f"{tag}/eval_logits"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(o["feat"].numpy(), g[f"{tag}/eval_feat"], rtol=2e-4, atol=2e-5)
    o = BR.bert_forward(P, ids, mask, cfg, seed=dseed)
    np.testing.assert_allclose(o["logits"].detach().numpy(), g[f"{tag}/train_logits"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(o["feat"].detach().numpy(), g[f"{tag}/train_feat"], rtol=2e-4, atol=2e-5)
    loss = (torch.nn.functional.cross_entropy(o["logits"], y, reduction="none") * w).mean()
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g[f"{tag}/loss"]), rel=1e-5)
    for n, _ in BR.param_shapes(cfg):
        gr = P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])
        check_samp(gr.numpy(), g.samp(f"{tag}/grad/{n}"), 2e-3, 1e-6, f"{tag} grad {n}")
    keep = BR.keep_mask(dseed, 7, (1 << 20,), 0.1)
    assert abs(keep.mean() - 0.9) < 2e-3                                          # the shared generator is a fair Bernoulli(0.9)
    # ... whose two decisions per hash (low / high 16 bits) are independent of each other: P(both kept) = 0.81, P(even) = P(odd) = 0.9
    assert abs((keep[0::2] & keep[1::2]).mean() - 0.81) < 3e-3 and abs(keep[0::2].mean() - 0.9) < 3e-3 and abs(keep[1::2].mean() - 0.9) < 3e-3
    k4 = BR.keep_mask(dseed, 9, (2, 3, 64, 65), 0.1)                               # attention probabilities: pairs within a row (odd row length)
    assert abs(k4.mean() - 0.9) < 6e-3 and abs((k4[..., 0:64:2] & k4[..., 1:64:2]).mean() - 0.81) < 1e-2


def test_srsoftmatch_bert_trace(golden):
    """SRSoftMatch on the BERT backbone (usb_nlp, BASELINE.json configs[3]: dict batches of different padded lengths, use_cat False, AdamW
    with layer decay through ClassificationBert.group_matcher) against a trace of the reference itself."""
    from oracle import bert_ref as BR
    from oracle.gen_golden import TRACE_SOFT_BERT as tr, synth_token_step, trace_bert_params
    from oracle.srsoftmatch_ref import SRSoftMatchBertOracle
    g = golden("srsoftmatch_bert_trace")
    C, seed = tr["C"], tr["seed"]
    cfg = BR.BertCfg(num_classes=C, p_drop=0.0, **BR.BERT_TINY_TEST)
    Fd = cfg.hidden
    orc = SRSoftMatchBertOracle(
        cfg, TP(trace_bert_params(cfg, seed, tr["head_gain"])), TP(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)),
        TP(synth.synth_params(S.generator_shapes(Fd), seed + 2)), num_train_iter=tr["num_train_iter"], start_timing=tr["start_timing"],
        N_k=tr["N_k"], ulb_dest_len=tr["ulb_dest_len"], num_warmup_iter=tr["num_warmup_iter"], ema_p=tr["ema_p"], n_sigma=tr["n_sigma"],
        dist_uniform=tr["dist_uniform"], lr=tr["lr"], weight_decay=tr["weight_decay"], layer_decay=tr["layer_decay"])
    tt = lambda b: (T(b[0]), T(b[1]))   # noqa: E731
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        orc.it = it
        K = int(g[f"{p}/K"])
        lb, y, w, s_ = synth_token_step(tr, cfg, n)
        assert len({lb[0].shape[1], w[0].shape[1], s_[0].shape[1]}) == 3          # three different padded lengths
        t = orc.train_step(tt(lb), T(y), tt(w), tt(s_))
        assert t["K"] == K
        np.testing.assert_allclose(np.stack([q["mask"].numpy() for q in t["passes"]]), g[f"{p}/masks"], rtol=2e-4, atol=1e-6)
        for k_ in ("sup_loss", "unsup_loss", "total_loss", "util_ratio"):
            assert t[k_] == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=5e-5, abs=2e-6), (p, k_)
        assert float(orc.sm.mu) == pytest.approx(float(g[f"{p}/mu"]), rel=1e-5) and float(orc.sm.var) == pytest.approx(float(g[f"{p}/var"]), rel=1e-4)
        np.testing.assert_allclose(orc.da.p_model.numpy(), g[f"{p}/p_model"], rtol=1e-5)
        assert t["lr_factor"] == pytest.approx(float(g[f"{p}/lr_factor"]), rel=1e-9, abs=1e-12)
        for nme, _ in BR.param_shapes(cfg):
            check_samp(t["grads"][nme].numpy(), g.samp(f"{p}/grad/{nme}"), 3e-3, 2e-6, f"{p} grad {nme}",
                       exclude=(lambda i: np.ones_like(i, bool)) if nme.endswith("key.bias") else None)     # analytically zero: round-off only
    for nme, _ in BR.param_shapes(cfg):                                              # after 6 AdamW steps with layer decay 0.65
        if not nme.endswith("key.bias"):
            check_samp(orc.P[nme].numpy(), g.samp(f"it{tr['its'][-1]}/param/{nme}"), 2e-3, 3e-5, f"param {nme}")


@pytest.mark.parametrize("tag", ["tiny", "tiny_skip", "base", "hubert_tiny"])
def test_w2v_oracle_matches_reference(golden, tag):
    """oracle/w2v2_ref.py against the reference ClassificationWave2Vec on a random-init HF Wav2Vec2Model (base-960h hyper-parameters):
    eval forward; train forward with the injected dropout masks, SpecAugment mask and LayerDrop decisions; gradients of a weighted CE."""
    from oracle import w2v2_ref as WR
    g = golden("w2v")
    C, B, S, seed, dseed = [int(v) for v in g[f"{tag}/meta"]]
    cfg = WR.W2vCfg(num_classes=C, **(WR.W2V_BASE if tag == "base" else WR.W2V_TINY_TEST))
    P = {k: T(v).requires_grad_(True) for k, v in WR.synth_params(cfg, seed).items()}
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    wave = T(rng.standard_normal((B, S)).astype(np.float32))
    y, w = T(rng.integers(0, C, size=(B,), dtype=np.int64)), T(rng.random(B).astype(np.float32))
    spec = WR.spec_augment_mask(seed + 2, B, WR.frames(cfg, S)[-1], cfg.mask_time_prob, cfg.mask_time_length, cfg.mask_time_min_masks)
    assert np.array_equal(spec, g[f"{tag}/spec_mask"]) and spec.any() and not spec.all()
    skip = [bool(v) for v in g[f"{tag}/skip"]]
    with torch.no_grad():
        o = WR.w2v_forward(P, wave, cfg)
    np.testing.assert_allclose(o["logits"].numpy(), g[f"{tag}/eval_logits"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(o["feat"].numpy(), g[f"{tag}/eval_feat"], rtol=3e-4, atol=3e-5)
    o = WR.w2v_forward(P, wave, cfg, seed=dseed, spec_mask=spec, skip=skip)
    np.testing.assert_allclose(o["logits"].detach().numpy(), g[f"{tag}/train_logits"], rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(o["feat"].detach().numpy(), g[f"{tag}/train_feat"], rtol=3e-4, atol=3e-5)
    loss = (torch.nn.functional.cross_entropy(o["logits"], y, reduction="none") * w).mean()
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g[f"{tag}/loss"]), rel=2e-5)
    for n, _ in WR.param_shapes(cfg):
        gr = P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])
        check_samp(gr.numpy(), g.samp(f"{tag}/grad/{n}"), 3e-3, 2e-6, f"{tag} grad {n}",
                   exclude=(lambda i: np.ones_like(i, bool)) if n.endswith("k_proj.bias") else None)        # analytically zero: round-off only


def test_srfreematch_w2v_trace(golden):
    """SRFreeMatch on the Wav2Vec2 backbone (usb_audio, BASELINE.json configs[4]: raw waveforms, use_cat False, AdamW with layer decay through
    ClassificationWave2Vec.group_matcher) against a trace of the reference itself."""
    from oracle import w2v2_ref as WR
    from oracle.gen_golden import TRACE_FREE_W2V as tr, W2V_QUIET, synth_wave_step, trace_w2v_params
    from oracle.srfreematch_ref import SRFreeMatchW2vOracle
    g = golden("srfreematch_w2v_trace")
    C, seed = tr["C"], tr["seed"]
    cfg = WR.W2vCfg(num_classes=C, **WR.W2V_TINY_TEST, **W2V_QUIET)
    Fd = cfg.hidden
    orc = SRFreeMatchW2vOracle(
        cfg, TP(trace_w2v_params(cfg, seed, tr["head_gain"])), TP(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)),
        TP(synth.synth_params(S.generator_shapes(Fd), seed + 2)), num_train_iter=tr["num_train_iter"], start_timing=tr["start_timing"],
        N_k=tr["N_k"], ulb_dest_len=tr["ulb_dest_len"], num_warmup_iter=tr["num_warmup_iter"], ema_p=tr["ema_p"],
        use_quantile=tr["use_quantile"], clip_thresh=tr["clip_thresh"], lambda_e=tr["ent_loss_ratio"], lr=tr["lr"],
        weight_decay=tr["weight_decay"], layer_decay=tr["layer_decay"])
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        orc.it = it
        K = int(g[f"{p}/K"])
        xl, y, xw, xs = synth_wave_step(tr, n)
        t = orc.train_step(T(xl), T(y), T(xw), T(xs))
        assert t["K"] == K
        assert np.array_equal(np.stack([q["mask"].numpy() for q in t["passes"]]), g[f"{p}/masks"]), p
        for k_ in ("sup_loss", "unsup_loss", "total_loss", "util_ratio"):
            assert t[k_] == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=1e-4, abs=5e-6), (p, k_)
        assert float(orc.fm.time_p) == pytest.approx(float(g[f"{p}/time_p"]), rel=1e-5)
        np.testing.assert_allclose(orc.fm.p_model.numpy(), g[f"{p}/p_model"], rtol=1e-5)
        assert t["lr_factor"] == pytest.approx(float(g[f"{p}/lr_factor"]), rel=1e-9, abs=1e-12)
        for nme, _ in WR.param_shapes(cfg):
            check_samp(t["grads"][nme].numpy(), g.samp(f"{p}/grad/{nme}"), 5e-3, 3e-6, f"{p} grad {nme}",
                       exclude=(lambda i: np.ones_like(i, bool)) if nme.endswith("k_proj.bias") else None)
    for nme, _ in WR.param_shapes(cfg):
        if not nme.endswith("k_proj.bias"):
            check_samp(orc.P[nme].numpy(), g.samp(f"it{tr['its'][-1]}/param/{nme}"), 2e-3, 5e-5, f"param {nme}")


def test_augment_oracle_matches_reference(golden):
    """oracle/augment_ref.py against the reference's RandAugment functions (Pillow underneath): every op, Cutout, whole chains -- bit for bit."""
    from oracle import augment_ref as A
    from oracle.gen_golden import synth_image
    g = golden("augment")
    for oi, name in enumerate(A.OPS):
        for t in range(4):
            seed, H, W, kind = [int(v) for v in g[f"op/{oi}/{t}/meta"]]
            got = A.apply_op(oi, synth_image(seed, H, W, kind), float(g[f"op/{oi}/{t}/v"]))
            assert np.array_equal(got, g[f"op/{oi}/{t}/out"]), (name, t)
    for t in range(6):
        seed, H, W, kind = [int(v) for v in g[f"chain/{t}/meta"]]
        x = synth_image(seed, H, W, kind)
        for o, v in zip(g[f"chain/{t}/ops"], g[f"chain/{t}/vals"]):
            x = A.apply_op(int(o), x, float(v))
        cv, ux, uy = [float(v) for v in g[f"chain/{t}/cut"]]
        assert np.array_equal(A.cutout(x, cv, ux, uy), g[f"chain/{t}/out"]), t


def test_augment_oracle_matches_the_whole_reference_transforms(golden):
    """transform_weak / transform_strong (cv_datasets/cifar.py:34-49) end to end: tests/golden/augment_tv.npz was produced by the op sequence of
    torchvision's PIL branch on Pillow images (np.pad 'reflect' -> Image.crop -> Image.transpose), the REFERENCE's RandAugment(3, 5) class and
    torch's ToTensor / Normalize arithmetic (oracle/gen_golden.py:gen_augment_tv says which call stands for which transform; torchvision itself
    is absent from the build container).  The numpy oracle reproduces bytes and fp32 tensors exactly."""
    from oracle import augment_ref as A
    from oracle.gen_golden import synth_image
    g = golden("augment_tv")
    mean, std = tuple(float(v) for v in g["meta/mean"]), tuple(float(v) for v in g["meta/std"])
    flips = 0
    for n in range(int(g["meta/n"])):
        seed, S, kind, pad, i, j, flip = [int(v) for v in g[f"case/{n}/meta"]]
        assert pad == int(S * (1 - 0.875))
        flips += flip
        src = synth_image(seed, S, S, kind)
        assert np.array_equal(A.weak(src, pad, S, i, j, bool(flip), mean, std), g[f"case/{n}/weak"]), n
        cv, ux, uy = [float(v) for v in g[f"case/{n}/cut"]]
        x = A.crop_flip(src, pad, S, i, j, bool(flip))
        for o, v in zip(g[f"case/{n}/ops"], g[f"case/{n}/vals"]):
            x = A.apply_op(int(o), x, float(v))
        assert np.array_equal(A.cutout(x, cv, ux, uy), g[f"case/{n}/strong_u8"]), n
        assert np.array_equal(A.strong(src, pad, S, i, j, bool(flip), g[f"case/{n}/ops"], g[f"case/{n}/vals"], cv, ux, uy, mean, std),
                              g[f"case/{n}/strong"]), n
    assert 0 < flips < int(g["meta/n"])


def test_full_size_trace_filter_decisions(golden):
    """tests/golden/srflexmatch_full_trace.npz (the reference's own train_step at BASELINE.json configs[1] size: ViT-S/2, 100 classes, 8 / 8 / 8,
    50 000-entry table, K = 0 and K = 8): the numpy FlexMatch hook replays every pass's decision from the max-probs / pseudo labels the REFERENCE
    hook saw -- masks, classwise_acc and the table entries of the batch bit for bit, from the mid-training state the fixture was generated from --
    and the reward filter is mask2 = reward >= mean(reward) per pass.  The fixture is non-trivial (rows selected AND rejected by both filters)
    and keeps the margin it was chosen for (the GPU test needs every engine max-prob closer to the reference's than to its threshold)."""
    from oracle.gen_golden import FULL, full_hook_state
    g = golden("srflexmatch_full_trace")
    b = synth.synth_batch(int(g["meta/bseed"]), FULL["Bl"], FULL["Bu"], 32, FULL["C"], FULL["ulb_dest_len"])
    assert float(g["meta/margin"]) >= 0.2 and int(g["meta/bseed"]) == FULL["batch_seeds"][0]      # "slack" of the kept batch (gen_golden.run_full_step)
    for it in [int(i) for i in g["meta/its"]]:
        p = f"it{it}"
        K = int(g[f"{p}/K"])
        assert K == (0 if it <= FULL["start_timing"] else 8)
        sel0, acc0 = full_hook_state(b["idx_ulb"])
        st = H.FlexMatchState(FULL["ulb_dest_len"], FULL["C"], True)
        st.selected_label[:] = sel0
        st.classwise_acc[:] = acc0
        mp, mi = g[f"{p}/mask_probs"], g[f"{p}/pseudo_label"]
        room = np.minimum(np.minimum(np.abs(mp - g[f"{p}/mask_thr"]), np.abs(mp - FULL["p_cutoff"])), g[f"{p}/label_gap"])
        assert float((room / (0.3 * mp * (1.0 - mp) + 2e-3)).min()) >= float(g["meta/margin"]) - 1e-6
        for k in range(K + 1):
            probs = np.zeros((FULL["Bu"], FULL["C"]), np.float32)
            probs[np.arange(FULL["Bu"]), mi[k]] = mp[k]                    # masking only looks at (max, argmax) of each row
            thr = np.float32(FULL["p_cutoff"]) * (st.classwise_acc[mi[k]] / (np.float32(2.0) - st.classwise_acc[mi[k]]))
            assert np.array_equal(thr.view(np.uint32), g[f"{p}/mask_thr"][k].view(np.uint32)), (p, k)
            want = st.masking(probs, b["idx_ulb"], FULL["p_cutoff"])
            assert np.array_equal(want, g[f"{p}/masks"][k]), (p, k)
            assert np.array_equal(st.classwise_acc.view(np.uint32), g[f"{p}/accs"][k].view(np.uint32)), (p, k)
        assert np.array_equal(st.selected_label[b["idx_ulb"]], g[f"{p}/sel_after_batch"])
        assert int((st.selected_label != -1).sum()) == int(g[f"{p}/n_selected_after"])
        assert 0.0 < g[f"{p}/masks"].mean() < 1.0
        if K:
            r = g[f"{p}/reward"]
            assert np.array_equal(g[f"{p}/mask2"], (r >= r.mean(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32))
            assert 0.0 < g[f"{p}/mask2"].mean() < 1.0
        assert float(g[f"{p}/log/util_ratio"]) == pytest.approx(float(g[f"{p}/masks"][0].mean()), abs=1e-7)


def test_rounding_model_fixture_is_what_the_oracle_function_gives(golden):
    """srflexmatch_full_sweep_emu.npz (gen_golden.gen_sweep_full_emu) = oracle.vit_ref.vit_forward_engine_rounding + the oracle's FlexMatch state
    machine: one step of the sweep recomputed here (K = 0: one pass of 8 weak images at full size) gives the fixture's max-probs (1e-6: CPU
    thread counts move fp32 sums), labels, masks and table entries."""
    import torch
    from oracle import hooks_ref as H
    from oracle import vit_ref as V
    from oracle.gen_golden import FULL, full_hook_state, trace_vit_params
    from semireward_amd.utils import synth
    e = golden("srflexmatch_full_sweep_emu")
    gain, bseed, it = float(e["meta/gain"]), int(e["meta/batches"][3]), 1000
    cfg = V.VitCfg(num_classes=FULL["C"], **V.VIT_SMALL_P2_32)
    P = {k: torch.from_numpy(v) for k, v in trace_vit_params(cfg, FULL["seed"], gain).items()}
    b = synth.synth_batch(bseed, FULL["Bl"], FULL["Bu"], cfg.img_size, FULL["C"], FULL["ulb_dest_len"])
    sel0, acc0 = full_hook_state(b["idx_ulb"])
    st = H.FlexMatchState(FULL["ulb_dest_len"], FULL["C"], True)
    st.selected_label[:] = sel0
    st.classwise_acc[:] = acc0
    dp = torch.from_numpy(synth.synth_droppath(900 + 16 * (bseed % 64), V.drop_path_probs(cfg), FULL["Bl"] + 2 * FULL["Bu"]))[:, :, FULL["Bl"]:FULL["Bl"] + FULL["Bu"]]
    with torch.no_grad():
        pr = torch.softmax(V.vit_forward_engine_rounding(P, torch.from_numpy(b["x_ulb_w"]), cfg, dp)["logits"], dim=-1)
    p = "g%g/b%d/it%d/" % (gain, bseed, it)
    np.testing.assert_allclose(pr.max(dim=-1).values.numpy(), e[p + "mask_probs"][0], rtol=0, atol=1e-5)
    assert np.array_equal(pr.argmax(dim=-1).numpy(), e[p + "pseudo_label"][0])
    assert np.array_equal(st.masking(pr.numpy(), b["idx_ulb"], FULL["p_cutoff"]), e[p + "masks"][0])
    assert np.array_equal(st.selected_label[b["idx_ulb"]], e[p + "sel_after_batch"])
