"""BERT engine (semireward_amd/nets/bert.py) on the HIP kernels: building blocks against fp64 torch restatements with the SAME dropout
masks (oracle/bert_ref.keep_mask), the whole backbone against vectors produced by the reference ClassificationBert on a random-init HF
BertModel (tests/golden/bert.npz): eval forward, train forward with injected dropout, gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bert_ref as BR                  # noqa: E402
from semireward_amd import ops                     # noqa: E402
from semireward_amd.nets import bert               # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def keep(seed, site, shape, p):
    return torch.from_numpy(BR.keep_mask(seed, site, shape, p))


def test_site_key_matches_oracle():
    for seed, site in [(0, 0), ((71 << 32) + 5, 3), (2 ** 64 - 1, BR.SITE_HEAD), (123456789012345, BR.SITE_EMB)]:
        assert ops.site_key(seed, site) == BR.site_key(seed, site)


@pytest.mark.parametrize("B,N,H,p", [(3, 24, 2, 0.0), (2, 80, 12, 0.1), (2, 200, 3, 0.1), (5, 257, 2, 0.1), (2, 512, 2, 0.1), (1, 512, 12, 0.0),
                                     (3, 400, 2, 0.1), (4, 300, 3, 0.0), (6, 512, 1, 0.1)])      # (N > 288: the paired-tile forward, odd tile counts, short key lengths)
def test_attention_masked_dropout_fwd_bwd(B, N, H, p):
    """softmax(q k^T / 8 + key mask) with dropout on the probabilities, forward and backward, incl. N = 512 (V fragments of the dQ pass
    from L2) -- against an fp64 restatement that uses the same counter-based keep mask."""
    D, seed, site = H * 64, (9 << 32) + 77, 6
    qkv = rnd(B * N, 3 * D, seed=7, scale=1.2).to(torch.bfloat16)
    rng = np.random.Generator(np.random.PCG64(N))
    klen = torch.from_numpy(rng.integers(max(1, N // 3), N + 1, size=B).astype(np.int32))
    klen[0] = N
    out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, N, device=DEV)
    dr = ops.Drop(seed, site, p) if p > 0 else None
    ops.attn_masked_fwd(qkv, out, lse, klen.to(DEV), B, N, H, 0.125, dr)
    x = qkv.double().cpu().requires_grad_(True)
    q, k, v = (x.view(B, N, 3, H, 64)[:, :, j].transpose(1, 2) for j in range(3))
    neg = torch.where(torch.arange(N)[None] < klen[:, None], 0.0, float("-inf"))[:, None, None, :].double()
    probs = torch.softmax(q @ k.transpose(-1, -2) * 0.125 + neg, -1)
    if p > 0:
        probs = probs * keep(seed, site, (B, H, N, N), p).double() / (1 - p)
    ro = (probs @ v).transpose(1, 2).reshape(B * N, D)
    assert rel(out, ro) < 6e-3
    d_out = rnd(B * N, D, seed=8).to(torch.bfloat16)
    ro.backward(d_out.double().cpu())
    dqkv = torch.zeros(B * N, 3 * D, dtype=torch.bfloat16, device=DEV)
    delta = torch.empty(B, H, N, device=DEV)
    ops.attn_masked_bwd(qkv, out, d_out, lse, dqkv, delta, klen.to(DEV), B, N, H, 0.125, dr)
    g = x.grad
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert rel(dqkv[:, sl], g[:, sl]) < 1.5e-2, name
    pad = (torch.arange(N)[None] >= klen[:, None]).reshape(-1)                   # padded positions: never attended -> dK = dV = 0 exactly
    assert float(dqkv[pad.to(DEV)][:, D:].float().abs().max() if pad.any() else 0.0) == 0.0


@pytest.mark.parametrize("D,p", [(128, 0.0), (768, 0.1)])
def test_embed_postln_meanpool(D, p):
    B, L, V, S, seed = 5, 19, 50, 3, (3 << 32) + 9
    rng = np.random.Generator(np.random.PCG64(D))
    ids = torch.from_numpy(rng.integers(0, V, size=(S, L), dtype=np.int64))
    ids[:, -3:] = 0                                                                    # [PAD] rows
    idx = torch.tensor([2, 0, 1, 1, 2], dtype=torch.int32)
    word, pos, typ = rnd(V, D, seed=1), rnd(L + 4, D, seed=2), rnd(2, D, seed=3)
    gam, bet = 1 + 0.1 * rnd(D, seed=4), 0.1 * rnd(D, seed=5)
    M = B * L
    x, xb = torch.empty(M, D, device=DEV), torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    dr = (lambda s: ops.Drop(seed, s, p)) if p > 0 else (lambda s: None)
    ops.embed_ln_fwd(ids.to(DEV), idx.to(DEV), word, pos, typ, gam, bet, 1e-12, x, xb, mean, rstd, B, L, D, dr(1))
    c = lambda t: t.double().cpu().requires_grad_(True)   # noqa: E731
    wc, pc, tc, gc, bc = c(word), c(pos), c(typ), c(gam), c(bet)
    e = torch.nn.functional.embedding(ids[idx.long()], wc, padding_idx=0) + pc[:L][None] + tc[0]
    r = torch.nn.functional.layer_norm(e, (D,), gc, bc, 1e-12)
    if p > 0:
        r = r * keep(seed, 1, (B, L, D), p).double() / (1 - p)
    assert rel(x, r.reshape(M, D)) < 2e-6 and rel(xb.float(), r.reshape(M, D)) < 4e-3
    dy = rnd(M, D, seed=6)
    r.backward(dy.double().cpu().view(B, L, D))
    dw, dp_, dt = torch.zeros(V, D, device=DEV), torch.zeros(L + 4, D, device=DEV), torch.zeros(2, D, device=DEV)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.embed_ln_bwd(dy, ids.to(DEV), idx.to(DEV), word, pos, typ, mean, rstd, gam, dw, dp_, dt[0], dg, db, B, L, D, 0, dr(1))
    assert rel(dw, wc.grad) < 1e-5 and rel(dp_, pc.grad) < 1e-5 and rel(dt, tc.grad) < 1e-5
    assert rel(dg, gc.grad) < 1e-5 and rel(db, bc.grad) < 1e-5
    assert float(dw[0].abs().max()) == 0.0
    # post-LN forward / backward (dropout mask on the bf16 branch gradient only)
    y = rnd(M, D, seed=7, scale=2.0) + 0.3
    ops.postln_fwd(y, gam, bet, 1e-12, x, xb, mean, rstd, M, D)
    yc = c(y); gc, bc = c(gam), c(bet)
    r = torch.nn.functional.layer_norm(yc, (D,), gc, bc, 1e-12)
    assert rel(x, r) < 2e-6 and rel(xb.float(), r) < 4e-3
    r.backward(dy.double().cpu())
    dx, dxb = torch.empty(M, D, device=DEV), torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    dg.zero_(); db.zero_()
    ops.postln_bwd(dy, y, mean, rstd, gam, dx, dxb, dg, db, M, D, dr(2))
    assert rel(dx, yc.grad) < 1e-5 and rel(dg, gc.grad) < 1e-5 and rel(db, bc.grad) < 1e-5
    masked = yc.grad * (keep(seed, 2, (M, D), p).double() / (1 - p) if p > 0 else 1.0)
    assert rel(dxb.float(), masked) < 4e-3
    # in-place variant (dx aliases dy), as the engine calls it
    dy2 = dy.clone()
    ops.postln_bwd(dy2, y, mean, rstd, gam, dy2, dxb, dg, db, M, D, dr(2))
    assert torch.equal(dy2, dx)
    # mean pool over ALL positions + its adjoint
    feat = torch.empty(B, D, device=DEV)
    xin = rnd(M, D, seed=8)
    ops.meanpool_fwd(xin, feat, B, L, D, dr(3))
    xc = c(xin)
    r = (xc.view(B, L, D) * (keep(seed, 3, (B, L, D), p).double() / (1 - p) if p > 0 else 1.0)).mean(1)
    assert rel(feat, r) < 2e-6
    df = rnd(B, D, seed=9)
    r.backward(df.double().cpu())
    ops.meanpool_bwd(df, dx, B, L, D, dr(3))
    assert rel(dx, xc.grad) < 2e-6
    # differently padded batches in one launch: rows >= seq_len[b] are filler (not averaged, zero gradient)
    sl = torch.tensor([L, L - 4, 1, L, 7], dtype=torch.int32)
    ops.meanpool_fwd(xin, feat, B, L, D, dr(3), sl.to(DEV))
    xc = c(xin)
    km = (keep(seed, 3, (B, L, D), p).double() / (1 - p) if p > 0 else torch.ones(B, L, D, dtype=torch.float64))
    valid = (torch.arange(L)[None] < sl[:, None]).double()[:, :, None]
    r = (xc.view(B, L, D) * km * valid).sum(1) / sl.double()[:, None]
    assert rel(feat, r) < 2e-6
    r.backward(df.double().cpu())
    ops.meanpool_bwd(df, dx, B, L, D, dr(3), sl.to(DEV))
    assert rel(dx, xc.grad) < 2e-6 and float(dx.view(B, L, D)[2, 1:].abs().max()) == 0.0
    # GELU (exact erf) forward / backward
    ops.gelu_f32(xin, dx, M * D)
    xc = c(xin); r = torch.nn.functional.gelu(xc); r.backward(dy.double().cpu())
    assert rel(dx, r) < 2e-6
    ops.gelu_bwd_f32(dy, xin, dx, M * D)
    assert rel(dx, xc.grad) < 2e-6
    # key lengths
    am = (torch.arange(L)[None] < torch.tensor([[3], [19], [7]])).long()
    kl = torch.empty(S, dtype=torch.int32, device=DEV)
    ops.mask_lengths(am.to(DEV), kl, S, L)
    assert kl.cpu().tolist() == [3, 19, 7]


def test_gemm_resid_dropout():
    M, N, K, seed = 300, 768, 256, (5 << 32) + 1
    A, Bm = rnd(M, K, seed=1).to(torch.bfloat16), rnd(N, K, seed=2, scale=0.1).to(torch.bfloat16)
    bias, R = rnd(N, seed=3), rnd(M, N, seed=4)
    for big in (False, True):
        Mx = 8192 + 37 if big else M            # 64x64-tile kernel (under-filled grid) and the 128x128 LDS-DMA kernel
        Ax = A.repeat(Mx // M + 1, 1)[:Mx].contiguous()
        Rx = R.repeat(Mx // M + 1, 1)[:Mx].contiguous()
        C = torch.empty(Mx, N, device=DEV)
        ops.gemm_nt_resid_dropout(Ax, Bm, C, Mx, N, K, bias, Rx, ops.Drop(seed, 4, 0.1))
        ref = Rx.double().cpu() + (Ax.double().cpu() @ Bm.double().cpu().t() + bias.double().cpu()) * keep(seed, 4, (Mx, N), 0.1).double() / 0.9
        assert rel(C, ref) < 1e-5
        C2 = Rx.clone()
        ops.gemm_nt_resid_dropout(Ax, Bm, C2, Mx, N, K, bias, None, None)                 # in place, no dropout
        assert rel(C2, Rx.double().cpu() + Ax.double().cpu() @ Bm.double().cpu().t() + bias.double().cpu()) < 1e-5


# Gradient bounds of the BERT golden test.  Against the fixture only 256 SAMPLED elements per tensor are compared: the sampled rel-L2 of a tensor
# moves between 0.04 and 0.10 with the realisation of the dropout masks / bf16 noise (0.059 with the per-element hash of rounds 1-5, 0.100 on
# classifier.0.weight with the pair hash -- other masks, same kernels).  Against the oracle's fp32 autograd over the FULL tensors
# (oracle.bert_ref.bert_forward, which test_oracle_golden pins to the reference) every tensor is within 0.034-0.038 for five mask seeds
# (tools/bert_grad_probe.py): that comparison carries the tight bound.
GRAD_TOL, GRAD_TOL_FULL = 0.12, 5e-2


@pytest.mark.parametrize("tag", ["tiny", "base"])
def test_bert_matches_reference_golden(golden, tag):
    g = golden("bert")
    C, B, L, seed, dseed = [int(v) for v in g[f"{tag}/meta"]]
    cfg = BR.BertCfg(num_classes=C, **(BR.BERT_TINY_TEST if tag == "tiny" else BR.BERT_BASE))
    model = bert.ClassificationBert(bert.BertConfig(num_classes=C, **(BR.BERT_TINY_TEST if tag == "tiny" else BR.BERT_BASE)), device=DEV)
    assert sorted(n for n, _ in model.names_shapes) == sorted(n for n, _ in BR.param_shapes(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in BR.synth_params(cfg, seed).items()})
    ids, mask = (torch.from_numpy(a) for a in BR.synth_tokens(seed + 1, B, L, cfg.vocab))
    rng = np.random.Generator(np.random.PCG64(seed + 2))
    y = torch.from_numpy(rng.integers(0, C, size=(B,), dtype=np.int64)).to(DEV)
    w = torch.from_numpy(rng.random(B).astype(np.float32)).to(DEV)
    x = {"input_ids": ids.to(DEV), "attention_mask": mask.to(DEV)}
    TOL, TOLF = 4e-2, 2e-2                           # bf16 GEMM / attention operands through up to 12 post-LN layers vs the fp32 reference
                                                     # (logits: small differences of pooled features through the 2-layer head)
    model.eval()
    o = model(x)
    assert rel(o["logits"], g[f"{tag}/eval_logits"]) < TOL and rel(o["feat"], g[f"{tag}/eval_feat"]) < TOLF
    model.train()
    model.inject_seed = dseed
    tok = bert.TokenBatch.from_dict(x, DEV)
    lg, ft, ctx = model.forward_features(tok, None, save=True)
    assert rel(lg, g[f"{tag}/train_logits"]) < TOL and rel(ft, g[f"{tag}/train_feat"]) < TOLF
    lg_i, ft_i, _ = model.forward_features(tok, None, save=False)                          # the no-save path computes the same thing
    assert rel(lg_i, lg) < 2e-3 and rel(ft_i, ft) < 2e-3
    loss, dl = torch.empty(1, device=DEV), torch.empty(B, C, device=DEV)
    ops.masked_ce(lg, y, w, None, 1.0, loss, dl, B, C)
    assert float(loss) == pytest.approx(float(g[f"{tag}/loss"]), rel=3e-2)
    model.zero_grad()
    model.backward(ctx, dl)
    worst = {}
    for n, gr in model.named_grads():
        gs = g.samp(f"{tag}/grad/{n}")
        a = gr.reshape(-1).cpu().numpy()[::gs["stride"]]
        if np.abs(gs["sample"]).max() == 0.0:
            assert np.abs(a).max() == 0.0, n                                               # pooler, [PAD] row
            continue
        if n.endswith("key.bias"):
            # analytically ZERO (a key bias shifts every score of a query row by the same q . b_k: softmax is invariant); the reference's
            # own value is fp32 round-off, ours is bf16 round-off -- both must vanish against the query-bias gradient
            qs = np.abs(g.samp(f"{tag}/grad/{n.replace('key', 'query')}")["sample"]).max()
            assert np.abs(gs["sample"]).max() < 1e-4 * qs and np.abs(a).max() < 3e-2 * qs, n
            continue
        worst[n] = rel(a, gs["sample"])
    print("bert %s: worst gradient tensors (rel-L2 of the sampled elements): %s" % (
        tag, ", ".join("%s %.3f" % (k.replace("bert.encoder.layer.", "L"), v) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6])))
    bad = {k: v for k, v in worst.items() if v > GRAD_TOL}
    assert not bad, bad
    # ... and every FULL gradient tensor against the oracle's fp32 autograd of the same step with the same masks
    Pq = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in BR.synth_params(cfg, seed).items()}
    oq = BR.bert_forward(Pq, ids, mask, cfg, seed=dseed)
    (torch.nn.functional.cross_entropy(oq["logits"], y.cpu(), reduction="none") * w.cpu()).mean().backward()
    full = {}
    for n, gr in model.named_grads():
        ref = Pq[n].grad
        if ref is None or float(ref.abs().max()) == 0.0 or n.endswith("key.bias"):
            continue
        full[n] = rel(gr.reshape(-1).cpu().numpy(), ref.reshape(-1).numpy())
    print("bert %s: worst FULL gradient tensors vs the oracle's autograd: %s" % (
        tag, ", ".join("%s %.3f" % (k.replace("bert.encoder.layer.", "L"), v) for k, v in sorted(full.items(), key=lambda kv: -kv[1])[:4])))
    assert max(full.values()) < GRAD_TOL_FULL, sorted(full.items(), key=lambda kv: -kv[1])[:4]
    # gathered sequences (seq_index) = the same rows
    idx = torch.tensor([B - 1, 0], dtype=torch.int32, device=DEV)
    model.eval()
    lg2, _, _ = model.forward_features(tok, idx, save=False)
    assert rel(lg2, torch.from_numpy(g[f"{tag}/eval_logits"])[[B - 1, 0]]) < TOL


def test_srsoftmatch_bert_trace(golden):
    """BASELINE.json configs[3] (usb_nlp: BERT + SoftMatch + SemiReward, use_cat False, three batches of different padded lengths, AdamW
    with layer decay 0.65) end to end on the HIP engine against a trace of the reference: SoftMatch weights of every pass, losses, features,
    DistAlign / Gaussian EMA state, rewarder updates, parameters after the fused AdamW steps."""
    import argparse
    from oracle import semireward_ref as S
    from oracle.gen_golden import TRACE_SOFT_BERT as tr, synth_token_step, trace_bert_params
    from semireward_amd.algorithms import get_algorithm
    from semireward_amd.utils import synth
    g = golden("srsoftmatch_bert_trace")
    C, seed = tr["C"], tr["seed"]
    cfg = BR.BertCfg(num_classes=C, p_drop=0.0, **BR.BERT_TINY_TEST)
    Fd = cfg.hidden
    args = argparse.Namespace(
        algorithm="srsoftmatch", num_classes=C, num_train_iter=tr["num_train_iter"], epoch=1, ema_m=0.0, ulb_loss_ratio=1.0, use_cat=False,
        amp=False, optim="AdamW", lr=tr["lr"], weight_decay=tr["weight_decay"], layer_decay=tr["layer_decay"],
        num_warmup_iter=tr["num_warmup_iter"], T=0.5, hard_label=True, ema_p=tr["ema_p"], n_sigma=tr["n_sigma"], dist_uniform=tr["dist_uniform"],
        dist_align=True, per_class=False, ulb_dest_len=tr["ulb_dest_len"], N_k=tr["N_k"], start_timing=tr["start_timing"], feature_dim=Fd,
        sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99, gpu=0, rank=0, world_size=1, distributed=False)
    builder = lambda num_classes, device: bert.ClassificationBert(   # noqa: E731
        bert.BertConfig(num_classes=num_classes, p_drop=0.0, **BR.BERT_TINY_TEST), device=device)
    alg = get_algorithm(args, builder)
    Tn = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}   # noqa: E731
    alg.model.load_state_dict(Tn(trace_bert_params(cfg, seed, tr["head_gain"])))
    alg.rewarder.load_state_dict(Tn(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)))
    alg.generator.load_state_dict(Tn(synth.synth_params(S.generator_shapes(Fd), seed + 2)))
    dx = lambda b: {"input_ids": torch.from_numpy(b[0]), "attention_mask": torch.from_numpy(b[1])}   # noqa: E731
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        alg.it = it
        alg.optimizer.sched_step = it
        K = int(g[f"{p}/K"])
        lb, y, w, s_ = synth_token_step(tr, cfg, n)
        alg.trace = {}
        before = alg.rewarder.flat.clone()
        out, log = alg.train_step(**alg.process_batch(x_lb=dx(lb), y_lb=torch.from_numpy(y), x_ulb_w=dx(w), x_ulb_s=dx(s_)))
        alg.out_dict, alg.log_dict = out, log
        assert alg.optimizer.lr_factor() == pytest.approx(float(g[f"{p}/lr_factor"]), rel=1e-9, abs=1e-12)
        alg.call_hook("after_train_step")
        assert alg.trace["K"] == K
        ntol = 1.0 + n                                 # AdamW trajectory noise grows with the steps taken (see test_gpu_srflexmatch.py)
        masks = np.stack([m.cpu().numpy() for m in alg.trace["masks"]])
        np.testing.assert_allclose(masks, g[f"{p}/masks"], rtol=0.0, atol=6e-2 * ntol)      # continuous SoftMatch weights
        for k_ in ("sup_loss", "unsup_loss", "total_loss"):
            assert float(log["train/" + k_]) == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=5e-2 * ntol, abs=5e-3), (p, k_)
        for k_ in ("x_lb", "x_ulb_w", "x_ulb_s"):
            assert rel(out["feat"][k_], g[f"{p}/feat/{k_}"]) < 2e-2 * ntol, (p, k_)
        assert int(not torch.equal(before, alg.rewarder.flat)) == int(g[f"{p}/rewarder_updated"]), p
        sm, da = alg.hooks_dict["MaskingHook"], alg.hooks_dict["DistAlignHook"]
        assert float(sm.prob_max_mu_t) == pytest.approx(float(g[f"{p}/mu"]), rel=2e-2 * ntol)
        assert float(sm.prob_max_var_t) == pytest.approx(float(g[f"{p}/var"]), rel=8e-2 * ntol)
        assert rel(da.p_model, g[f"{p}/p_model"]) < 2e-2 * ntol
    worst = 0.0
    for nme, v in alg.model.named_parameters():
        if nme.endswith("key.bias"):
            continue                                   # analytic gradient 0: Adam amplifies round-off (reference) / bf16 noise (here)
        gs = g.samp(f"it{tr['its'][-1]}/param/{nme}")
        a = v.reshape(-1).cpu().numpy()[::gs["stride"]]
        worst = max(worst, float(np.abs(a - gs["sample"]).max()))
    assert worst < 4e-3, worst                         # <= a handful of lr-sized (5e-4) steps
    pool = alg.model.view("bert.pooler.dense.weight").cpu().numpy()
    assert np.array_equal(pool, trace_bert_params(cfg, seed, tr["head_gain"])["bert.pooler.dense.weight"])   # never touched (grad None in the reference)


def test_full_size_step_properties_bert():
    """BASELINE.json configs[3] at FULL size (bert-base, 8/8/8 sequences of 512 tokens, K = sr_decay() = 8, train-mode dropout on): the CPU
    oracle cannot step this in seconds, so parity goes through size-independent properties --
      * the SoftMatch weights of all 9 passes equal the oracle's SoftMatchState fed with the engine's own max-probs of each pass (sequential
        EMA state), the reward mask2 == (reward >= per-pass mean), util_ratio = mean(mask0);
      * K = 8 passes, 8 + 9 * 16 sequence forwards, labelled rows only in pass 0 (use_cat False), finite losses;
      * the step is reproducible: same state + same inputs + same dropout seeds -> identical logits (deferred rows included);
      * the pooler never moves."""
    import argparse
    from oracle import hooks_ref as H
    from semireward_amd.algorithms import get_algorithm
    # the configuration itself comes from the authored yaml of BASELINE configs[3] (IMDB: 2 classes; configs/README.md) through the yaml loader
    import os
    from semireward_amd import config as srconfig
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = srconfig.get_config(os.path.join(root, "configs", "usb_nlp_srsoftmatch_aclImdb_20_bert_base.yaml"),
                            overrides=dict(gpu=0, rank=0, world_size=1, distributed=False, ulb_dest_len=50000))
    assert a.algorithm == "srsoftmatch" and a.net == "bert_base_uncased" and a.num_classes == 2 and a.use_cat is False and a.batch_size == 8
    args = vars(a)
    C, L, nl, nu = a.num_classes, a.max_length, a.batch_size, a.batch_size * a.uratio
    g = torch.Generator().manual_seed(5)
    def tok(n, full):
        lens = torch.full((n,), L) if full else torch.randint(L // 2, L + 1, (n,), generator=g)
        lens[0] = L
        mask = (torch.arange(L)[None] < lens[:, None]).long()
        return {"input_ids": torch.randint(1, 30522, (n, L), generator=g) * mask, "attention_mask": mask}
    batch = dict(x_lb=tok(nl, True), y_lb=torch.randint(0, C, (nl,), generator=g), x_ulb_w=tok(nu, False), x_ulb_s=tok(nu, False))
    runs = []
    for _ in range(2):
        alg = get_algorithm(argparse.Namespace(**args), bert.bert_base_uncased)
        alg.model.view("classifier.2.weight").mul_(8.0); alg.model.refresh_operands()          # spread the max-probs (random-init head is flat)
        alg.it = 90001
        alg.optimizer.sched_step = alg.it
        alg.model.seed, alg.model._rng_calls = 77, 0
        alg.trace = {}
        pool0 = alg.model.view("bert.pooler.dense.weight").clone()
        out, log = alg.train_step(**alg.process_batch(**batch))
        alg.out_dict, alg.log_dict = out, log
        alg.call_hook("after_train_step")
        torch.cuda.synchronize()
        runs.append((alg, out, log, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in alg.trace.items()}, pool0))
    alg, out, log, tr, pool0 = runs[0]
    K = tr["K"]
    assert K == 8 and tr["logits"].shape[:2] == (9, 24)
    st = H.SoftMatchState(C, 2, 0.999)
    da = H.DistAlignState(C, 0.999, "uniform")
    mp = tr["max_probs"].cpu().numpy().reshape(9, nu)
    Lw = tr["logits"][:, nl:nl + nu].float().cpu()
    for k in range(9):
        probs = torch.softmax(Lw[k], -1)
        if k == 0:                                                  # pass 0 weights the distribution-aligned probabilities
            probs = da.dist_align(probs, None)
        want = st.masking(probs).numpy()
        np.testing.assert_allclose(tr["masks"][k].cpu().numpy(), want, rtol=0, atol=2e-4)
    r = tr["reward"].cpu().numpy().reshape(K, nu)
    rm = r.mean(axis=1, keepdims=True, dtype=np.float32)
    far = np.abs(r - rm) > 1e-6           # rows with one pseudo label share one reward (the feature enters only through the batch context, A.4):
    m2 = tr["mask2"].cpu().numpy().reshape(K, nu)                          # exact ties with the mean are decided by the summation order
    assert np.array_equal(m2[far], (r >= rm).astype(np.float32)[far]) and set(np.unique(m2)) <= {0.0, 1.0}
    assert float(log["train/util_ratio"]) == pytest.approx(float(tr["masks"][0].mean()), abs=1e-6)
    assert all(np.isfinite(float(log["train/" + k_])) for k_ in ("sup_loss", "unsup_loss", "total_loss"))
    assert abs(mp.mean() - 0.5) < 0.45 and mp.std() > 1e-3          # the scoring saw non-degenerate probabilities
    tr2 = runs[1][3]
    assert torch.equal(tr["logits"][:, nl:], tr2["logits"][:, nl:]) and torch.equal(tr["logits"][0], tr2["logits"][0])
    assert torch.equal(alg.model.view("bert.pooler.dense.weight"), pool0)
    # ... and the AdamW step lands on the same parameters up to the fp32 atomics of the gradient column sums (order-dependent round-off that Adam
    # turns into at most a sign flip of one lr-sized step for gradients at the noise floor)
    assert float((alg.model.flat - runs[1][0].model.flat).abs().max()) <= 2.1 * 5e-5


def test_softmatch_weights_on_the_reference_probabilities_have_zero_deviation(golden):
    """Oracle-fed companion of the trace test: the SoftMatch weighting kernels on EXACTLY the probabilities every masking call of the
    reference received (srsoftmatch_bert_trace.npz it*/mask_probs; pass 0 after DistAlign) from the reference's own EMA state before the
    step (it*/pre/*) reproduce the reference's sample weights of all passes -- the backbone's bf16 arithmetic is out of the picture, so
    nothing may move: the binary part (weight == 1.0 above the mean) with 0 flips, the Gaussian tail to fp32 round-off."""
    import types
    from oracle.gen_golden import TRACE_SOFT_BERT as tr
    from semireward_amd.algorithms.hooks import SoftMatchWeightingHook
    g = golden("srsoftmatch_bert_trace")
    stub = types.SimpleNamespace(dp=None)
    total = 0
    for it in tr["its"]:
        p = f"it{it}"
        probs, want = g[f"{p}/mask_probs"], g[f"{p}/masks"]
        assert probs.shape[0] == want.shape[0] == int(g[f"{p}/K"]) + 1
        h = SoftMatchWeightingHook(tr["C"], n_sigma=tr["n_sigma"], momentum=tr["ema_p"], device=DEV)
        h.mu_var.copy_(torch.tensor([float(g[f"{p}/pre/mu"]), float(g[f"{p}/pre/var"])]))
        for k in range(probs.shape[0]):
            m = h.masking(stub, torch.from_numpy(probs[k]).to(DEV), softmax_x_ulb=False).cpu().numpy()
            assert np.array_equal(m == 1.0, want[k] == 1.0), (p, k)                     # which rows get the full weight: 0 flips
            np.testing.assert_allclose(m, want[k], rtol=2e-5, atol=1e-7, err_msg=f"{p} pass {k}")
            total += m.size
        assert float(h.prob_max_mu_t) == pytest.approx(float(g[f"{p}/mu"]), rel=1e-6)
        assert float(h.prob_max_var_t) == pytest.approx(float(g[f"{p}/var"]), rel=1e-5)
    assert total >= 200
