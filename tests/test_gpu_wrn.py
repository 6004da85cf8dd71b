"""WideResNet engine (semireward_amd/nets/wrn.py) on the HIP kernels against vectors produced by the reference WideResNet itself
(tests/golden/wrn.npz): eval / train / frozen-BN forwards, BatchNorm running statistics, gradients of two graphs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import wrn_ref as W                    # noqa: E402
from oracle.gen_golden import synth_wrn_params     # noqa: E402
from semireward_amd import ops                     # noqa: E402
from semireward_amd.nets import wrn                # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("tag", ["tiny", "wrn_28_2"])
def test_wrn_matches_reference_golden(golden, tag):
    g = golden("wrn")
    C, B, HW, seed = [int(v) for v in g[f"{tag}/meta"]]
    cfg = W.WrnCfg(num_classes=C, **(W.WRN_TINY_TEST if tag == "tiny" else W.WRN_28_2))
    model = (wrn.wrn_tiny_test if tag == "tiny" else wrn.wrn_28_2)(num_classes=C, device=DEV)
    assert [n for n, _ in model.names_shapes] == [n for n, _ in W.param_shapes(cfg)]
    sd = {k: torch.from_numpy(v) for k, v in synth_wrn_params(cfg, seed).items()}
    sd.update({k[len(tag) + 6:]: torch.from_numpy(g[k]) for k in g.keys() if k.startswith(f"{tag}/buf0/")})
    model.load_state_dict(sd)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = torch.from_numpy(rng.standard_normal((B, 3, HW, HW)).astype(np.float32)).to(DEV)
    x2 = torch.from_numpy(rng.standard_normal((B, 3, HW, HW)).astype(np.float32)).to(DEV)
    y = torch.from_numpy(rng.integers(0, C, size=(B,), dtype=np.int64)).to(DEV)
    w = torch.from_numpy(rng.random(B).astype(np.float32)).to(DEV)
    TOL = 2.5e-2                                   # bf16 conv operands through 25 convolutions + BatchNorm, fp32 reference
    model.eval()
    lg, ft, _ = model.forward_features(x)
    assert rel(lg.cpu(), g[f"{tag}/eval_logits"]) < TOL and rel(ft.cpu(), g[f"{tag}/eval_feat"]) < TOL
    model.train()
    lg1, ft1, ctx1 = model.forward_features(x, save=True, tag="g0")                       # labelled-style forward: statistics move
    assert rel(lg1.cpu(), g[f"{tag}/train_logits"]) < TOL and rel(ft1.cpu(), g[f"{tag}/train_feat"]) < TOL
    for k, v in model.buffers.items():
        if not k.endswith("num_batches_tracked"):
            assert rel(v.cpu(), g[f"{tag}/buf1/{k}"]) < 2e-4, k                               # momentum 0.001 keeps them near buf0
            d_ref = g[f"{tag}/buf1/{k}"] - g[f"{tag}/buf0/{k}"]
            assert rel(v.cpu().numpy() - g[f"{tag}/buf0/{k}"], d_ref) < 6e-2, k               # the UPDATE itself (batch statistics)
    snap = {k: v.clone() for k, v in model.buffers.items()}
    lg2, _, ctx2 = model.forward_features(x2, save=True, update_stats=False, tag="g1")    # Bn_Controller.freeze_bn
    assert rel(lg2.cpu(), g[f"{tag}/frozen_logits"]) < TOL
    assert all(torch.equal(v, snap[k]) for k, v in model.buffers.items())
    # backward of  mean(w * CE(lg1))  +  0.5 * mean(w * CE(lg2))
    loss1, loss2 = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    dl1, dl2 = torch.empty(B, C, device=DEV), torch.empty(B, C, device=DEV)
    ops.masked_ce(lg1, y, w, None, 1.0, loss1, dl1, B, C)
    ops.masked_ce(lg2, y, w, None, 0.5, loss2, dl2, B, C)
    assert float(loss1 + 0.5 * loss2) == pytest.approx(float(g[f"{tag}/loss"]), rel=3e-2)
    model.zero_grad()
    model.backward(ctx1, dl1)
    model.backward(ctx2, dl2)
    torch.cuda.synchronize()
    worst = []
    for n, gr in model.named_grads():
        gs = g.samp(f"{tag}/grad/{n}")
        a = gr.reshape(-1).cpu().numpy()[::gs["stride"]]
        if np.abs(gs["sample"]).max() == 0.0:                                                # the two dead bn1's: exactly no gradient
            assert np.abs(a).max() == 0.0, n
            continue
        if n == "conv1.bias":       # a per-channel constant in front of a BatchNorm: analytically zero gradient, round-off in the reference
            assert np.abs(gs["sample"]).max() < 1e-6 and np.abs(a).max() < 2e-4, (np.abs(a).max(), np.abs(gs["sample"]).max())
            continue
        worst.append((rel(a, gs["sample"]), n))
    worst.sort(reverse=True)
    # The engine differentiates ITS forward (bf16 convolution operands).  LeakyReLU has a kink: a pre-activation that lands on the other side
    # of zero than in the fp32 forward (a ~1 % population after a few layers) has a 10x different derivative, so against the fp32 reference
    # the gradients of a deep ReLU-type net agree only to ~20 % in L2 while every building block is exact (tests below).  The sharp check is
    # against the oracle evaluated with bf16-rounded convolution operands: same arithmetic, same kinks.
    assert worst[0][0] < 0.4 and np.median([v for v, _ in worst]) < 0.25, worst[:5]
    if tag == "tiny":
        Pt = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth_wrn_params(cfg, seed).items()}
        BUF = {k[len(tag) + 6:]: torch.from_numpy(g[k]).clone() for k in g.keys() if k.startswith(f"{tag}/buf0/")}
        o1 = W.wrn_forward(Pt, BUF, x.cpu(), cfg, train=True, update_stats=True, bf16_operands=True)
        o2 = W.wrn_forward(Pt, BUF, x2.cpu(), cfg, train=True, update_stats=False, bf16_operands=True)
        ce = torch.nn.functional.cross_entropy
        (((ce(o1["logits"], y.cpu(), reduction="none") * w.cpu()).mean()) + 0.5 * (ce(o2["logits"], y.cpu(), reduction="none") * w.cpu()).mean()).backward()
        assert rel(lg1.cpu(), o1["logits"].detach().numpy()) < 2e-3 and rel(lg2.cpu(), o2["logits"].detach().numpy()) < 2e-3
        errs = []
        for n, gr in model.named_grads():
            if Pt[n].grad is None or n == "conv1.bias":
                continue
            errs.append((rel(gr.cpu(), Pt[n].grad.numpy()), n))
        errs.sort(reverse=True)
        assert errs[0][0] < 4e-2 and np.median([v for v, _ in errs]) < 1.5e-2, errs[:5]


@pytest.mark.parametrize("B,H,C,Cout,ks,stride", [(3, 8, 32, 64, 3, 1), (2, 8, 32, 64, 3, 2), (2, 8, 16, 32, 1, 2), (4, 4, 128, 128, 3, 1)])
def test_conv_building_blocks(B, H, C, Cout, ks, stride):
    """im2col + GEMM forward, GEMM + col2im input gradient, TN weight gradient against F.conv2d autograd (fp32 on the bf16-rounded
    operands, so the comparison isolates the kernels from bf16 rounding)."""
    import torch.nn.functional as F
    rng = np.random.Generator(np.random.PCG64(B * 100 + C))
    bfr = lambda a: torch.from_numpy(a).to(torch.bfloat16)   # noqa: E731
    x = bfr(rng.standard_normal((B, H, H, C)).astype(np.float32))                  # NHWC
    Wt = (rng.standard_normal((Cout, C, ks, ks)) / np.sqrt(C * ks * ks)).astype(np.float32)
    K, Kp = C * ks * ks, (C * ks * ks + 31) // 32 * 32
    pad = ks // 2
    Ho = (H + 2 * pad - ks) // stride + 1
    rows = B * Ho * Ho
    xd = x.to(DEV).contiguous()
    Wb, WbT = torch.zeros(Cout, Kp, dtype=torch.bfloat16, device=DEV), torch.zeros(Kp, Cout, dtype=torch.bfloat16, device=DEV)
    ops.conv_weight_prep(torch.from_numpy(Wt).to(DEV).reshape(-1), Wb, WbT, Cout, C, ks, Kp)
    col = torch.empty(rows, Kp, dtype=torch.bfloat16, device=DEV)
    ops.im2col(xd, col, B, H, H, C, ks, stride, Kp)
    out = torch.empty(rows, Cout, device=DEV)
    ops.gemm_nt(ops.EPI_F32, col, Wb, out, rows, Cout, Kp)
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    Wr = Wb[:, :K].float().cpu().reshape(Cout, ks, ks, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)   # K axis: (tap, channel)
    ref = F.conv2d(xr, Wr, None, stride, pad)
    assert rel(out.cpu().reshape(B, Ho, Ho, Cout), ref.detach().permute(0, 2, 3, 1).numpy()) < 1e-5
    dy = bfr(rng.standard_normal((rows, Cout)).astype(np.float32))
    ref.backward(dy.float().reshape(B, Ho, Ho, Cout).permute(0, 3, 1, 2))
    dyd = dy.to(DEV).contiguous()
    dcol = torch.empty(rows, Kp, device=DEV)
    ops.gemm_nt(ops.EPI_F32, dyd, WbT, dcol, rows, Kp, Cout)
    dx = torch.full((B * H * H, C), 7.0, device=DEV)
    ops.col2im(dcol, dx, B, H, H, C, ks, stride, Kp, accumulate=False)
    assert rel(dx.cpu().reshape(B, H, H, C), xr.grad.permute(0, 2, 3, 1).numpy()) < 1e-5
    dW = torch.zeros(Cout, Kp, device=DEV)
    desc, npb, nt, _, _ = ops.make_group_tn_desc([(dyd, col, dW, None, Cout, Kp, rows)], DEV)
    ops.gemm_tn_grouped_f32(desc, npb, nt, alpha=1.0, beta=1.0)
    assert rel(dW[:, :K].cpu().reshape(Cout, ks, ks, C).permute(0, 3, 1, 2), Wr.grad.numpy()) < 1e-5
    # the same product as K slices meeting through atomic adds, and the gradient un-permuted into the parameter's [Cout, C, k, k] layout
    dW2 = torch.zeros(Cout, Kp, device=DEV)
    desc, npb, nt, _, _ = ops.make_group_tn_desc([(dyd, col, dW2, None, Cout, Kp, rows)], DEV, split_k=32)
    assert npb == (rows + 31) // 32 if rows >= 64 else npb == 1
    ops.gemm_tn_grouped_f32(desc, npb, nt, alpha=1.0, beta=1.0)
    assert rel(dW2.cpu(), dW.cpu()) < 1e-5
    # ... and as slices that write scratch slabs of their own + one reduce launch (SRHIP_TN_OVERWRITE, srhip_slab_reduce_f32): what the backward uses
    dW3 = dW.clone()
    desc, npb, nt, _, _ = ops.make_group_tn_desc([(dyd, col, dW3, None, Cout, Kp, rows)], DEV, split_k=32, slabs=True)
    assert (hasattr(desc, "reduce") and desc.reduce[1] == 1) if rows >= 64 else not hasattr(desc, "reduce")
    ops.gemm_tn_grouped_f32(desc, npb, nt, alpha=1.0, beta=1.0)
    assert rel(dW3.cpu(), 2 * dW.cpu()) < 1e-5                      # C += product: twice the gradient
    g = torch.zeros(Cout * K, device=DEV)
    ops.add_unpad(dW2, g, Cout, C, ks, Kp)
    assert rel(g.cpu().reshape(Cout, C, ks, ks), Wr.grad.numpy()) < 1e-5
    assert float(dW[:, K:].abs().max()) == 0.0 if Kp > K else True


@pytest.mark.parametrize("rows,C", [(96, 32), (1024, 128), (5000, 16), (65536, 32), (16384, 64), (300, 256), (70001, 4)])
def test_batchnorm_leakyrelu(rows, C):
    """srhip_bn_fwd / srhip_bn_bwd against F.batch_norm + leaky_relu autograd: batch statistics, running update (unbiased variance,
    momentum 0.001), frozen update, eval mode, input gradient with a residual term, dgamma / dbeta."""
    import torch.nn.functional as F
    rng = np.random.Generator(np.random.PCG64(rows + C))
    T = lambda a: torch.from_numpy(a.astype(np.float32))   # noqa: E731
    x = T(rng.standard_normal((rows, C)) * 1.7 + 0.3)
    gam, bet = T(1.0 + 0.1 * rng.standard_normal(C)), T(0.1 * rng.standard_normal(C))
    rm0, rv0 = T(0.1 * rng.standard_normal(C)), T(1.0 + 0.2 * rng.random(C))
    dact, resid = T(rng.standard_normal((rows, C))), T(rng.standard_normal((rows, C)))
    xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    y = F.leaky_relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.001, 1e-5), 0.1)
    y.backward(dact)
    d = lambda t: t.to(DEV).contiguous()   # noqa: E731
    xd, gd, bd = d(x), d(gam), d(bet)
    rmd, rvd = d(rm0), d(rv0)
    mean, invstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    act, af = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV), torch.empty(rows, C, device=DEV)
    ws = torch.zeros(ops.bn_ws_doubles(), dtype=torch.float64, device=DEV)          # zeroed once; every launch leaves its counter at zero
    ops.bn_fwd(xd, gd, bd, 1e-5, 0.1, 0.001, True, True, rmd, rvd, mean, invstd, act, af, ws, rows, C)
    assert int(ws[512:513].view(torch.int64)) == 0
    assert rel(af.cpu(), y.detach().numpy()) < 2e-6 and rel(act.float().cpu(), y.detach().numpy()) < 4e-3
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.numpy(), rtol=1e-6, atol=1e-7)
    dx, dg, db = torch.empty(rows, C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_bwd(d(dact), xd, mean, invstd, gd, bd, 0.1, d(resid), dx, dg, db, ws, rows, C)
    assert rel(dx.cpu(), (xr.grad + resid).numpy()) < 5e-6
    assert rel(dg.cpu(), gr.grad.numpy()) < 5e-6 and rel(db.cpu(), br.grad.numpy()) < 5e-6
    # frozen statistics (Bn_Controller) and eval mode
    rm1, rv1 = rmd.clone(), rvd.clone()
    ops.bn_fwd(xd, gd, bd, 1e-5, 0.1, 0.001, True, False, rmd, rvd, mean, invstd, act, af, ws, rows, C)
    assert torch.equal(rmd, rm1) and torch.equal(rvd, rv1) and rel(af.cpu(), y.detach().numpy()) < 2e-6
    # fp64 sums rounded to float: the same statistics launch after launch; the workspace is left clean (accumulator copies and counter zero)
    m0, i0, a0 = mean.clone(), invstd.clone(), af.clone()
    for _ in range(3):
        ops.bn_fwd(xd, gd, bd, 1e-5, 0.1, 0.001, True, False, rmd, rvd, mean, invstd, act, af, ws, rows, C)
        assert torch.equal(mean, m0) and torch.equal(invstd, i0) and torch.equal(af, a0)
    assert float(ws[512:].abs().max()) == 0.0
    ops.bn_fwd(xd, gd, bd, 1e-5, 0.1, 0.001, False, False, rmd, rvd, None, None, act, af, ws, rows, C)
    ye = F.leaky_relu(F.batch_norm(x, rm, rv, gam, bet, False, 0.001, 1e-5), 0.1)
    assert rel(af.cpu(), ye.numpy()) < 2e-6


def test_srpseudolabel_wrn_trace(golden):
    """BASELINE.json configs[0] (classic_cv: WideResNet + PseudoLabel + SemiReward, SGD) end to end on the HIP engine against a trace of
    the reference: per-pass masks, losses, features, BatchNorm running statistics (moved by the labelled forward only), rewarder updates,
    parameters after the fused SGD steps."""
    import argparse
    from oracle import semireward_ref as S
    from oracle.gen_golden import TRACE_PL_WRN as tr
    from semireward_amd.algorithms import get_algorithm
    from semireward_amd.utils import synth
    g = golden("srpseudolabel_wrn_trace")
    C, Bl, Bu, seed = tr["C"], tr["Bl"], tr["Bu"], tr["seed"]
    wcfg = W.WrnCfg(num_classes=C, **W.WRN_TINY_TEST)
    Fd = W.channels(wcfg)[3]
    args = argparse.Namespace(
        algorithm="srpseudolabel", num_classes=C, num_train_iter=tr["num_train_iter"], epoch=1, ema_m=tr["ema_m"], ulb_loss_ratio=1.0, use_cat=True,
        amp=False, optim="SGD", lr=tr["lr"], momentum=tr["momentum"], weight_decay=tr["weight_decay"], layer_decay=1.0,
        num_warmup_iter=tr["num_warmup_iter"], p_cutoff=tr["p_cutoff"], unsup_warm_up=tr["unsup_warm_up"], N_k=tr["N_k"],
        start_timing=tr["start_timing"], feature_dim=Fd, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99, gpu=0, rank=0, world_size=1,
        distributed=False, T=0.5, hard_label=True)
    alg = get_algorithm(args, wrn.wrn_tiny_test)
    Tn = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}   # noqa: E731
    alg.model.load_state_dict(Tn(synth_wrn_params(wcfg, seed)))
    assert alg.ema_m == 0.999 and alg.ema_model is not alg.model       # classic_cv yamls run ema_m 0.999 (pseudolabel_cifar100_400_0.yaml:20)
    alg.ema_model.load_state_dict(Tn(synth_wrn_params(wcfg, seed)))    # EMA.register(): shadow starts at the parameters (misc.py:146-148)
    alg.rewarder.load_state_dict(Tn(synth.synth_params(S.rewarder_shapes(Fd, C), seed + 1)))
    alg.generator.load_state_dict(Tn(synth.synth_params(S.generator_shapes(Fd), seed + 2)))
    flips = total = 0
    for n, it in enumerate(tr["its"]):
        p = f"it{it}"
        alg.it = it
        alg.optimizer.sched_step = it
        K = int(g[f"{p}/K"])
        b = synth.synth_batch(seed + 10 + n, Bl, Bu, tr["img"], C, tr["ulb_dest_len"])
        alg.trace = {}
        before = alg.rewarder.flat.clone()
        out, log = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))
        alg.out_dict, alg.log_dict = out, log
        assert alg.optimizer.lr_factor() == pytest.approx(float(g[f"{p}/lr_factor"]), rel=1e-9, abs=1e-12)
        alg.call_hook("after_train_step")
        assert alg.trace["K"] == K
        masks = np.stack([m.cpu().numpy() for m in alg.trace["masks"]])
        bad = masks != g[f"{p}/masks"]
        flips += int(bad.sum()); total += bad.size
        mpv = alg.trace["max_probs"].cpu().numpy().reshape(masks.shape)
        # the fixture's cut-off (0.18) is 1.4e-2 away from the nearest max-prob the reference thresholds: no mask may differ
        assert float(np.abs(g[f"{p}/mask_probs"] - tr["p_cutoff"]).min()) > 1e-2
        assert not bad.any(), (p, mpv[bad], g[f"{p}/mask_probs"][bad])
        for k_, v in alg.model.buffers.items():                                         # statistics: labelled forward only, momentum 0.001
            if not k_.endswith("num_batches_tracked"):
                # running_mean starts at 0: after a few steps it IS momentum * (bf16-affected batch means of a drifting trajectory, same
                # growth as the feature tolerance below); running_var starts at 1
                assert rel(v.cpu(), g[f"{p}/buf/{k_}"]) < (5e-2 + 3e-2 * n if k_.endswith("running_mean") else 1e-3), (p, k_)
        assert int(alg.model.buffers["bn1.num_batches_tracked"]) == n + 1
        assert int(not torch.equal(before, alg.rewarder.flat)) == int(g[f"{p}/rewarder_updated"]), p
        if bad.any():
            continue
        for k_ in ("sup_loss", "unsup_loss", "total_loss"):
            assert float(log["train/" + k_]) == pytest.approx(float(g[f"{p}/log/{k_}"]), rel=6e-2, abs=5e-3), (p, k_)
        for k_ in ("x_lb", "x_ulb_w"):
            assert rel(out["feat"][k_].cpu(), g[f"{p}/feat/{k_}"]) < 5e-2 + 3e-2 * n, (p, k_)   # SGD lr 0.03 trajectory drift (kink noise, see above)
    assert flips == 0 and total > 0, (flips, total)
    worst = 0.0
    for nme, v in alg.model.named_parameters():                                        # 6 SGD steps at lr 0.03 (LeakyReLU-kink gradient noise, see above)
        gs = g.samp(f"it{tr['its'][-1]}/param/{nme}")
        a = v.reshape(-1).cpu().numpy()[::gs["stride"]]
        worst = max(worst, float(np.abs(a - gs["sample"]).max()))
    assert worst < 3e-2, worst
    # EMA shadow of the run (ema_m 0.999, updated inside the fused SGD launch) against the reference's EMA / EMAHook sequence: the shadow is
    # 0.999^6 of the initial parameters + 0.1 % slices of a trajectory that agrees to `worst`, and its BatchNorm buffers are the model's
    last = f"it{tr['its'][-1]}"
    for nme, v in alg.ema_model.named_parameters():
        gs = g.samp(f"{last}/ema/{nme}")
        a = v.reshape(-1).cpu().numpy()[::gs["stride"]]
        assert float(np.abs(a - gs["sample"]).max()) < 1e-5 + 6e-3 * worst, nme
        m_ = alg.model.view(nme).reshape(-1).cpu().numpy()[::gs["stride"]]
        assert float(np.abs(a - m_).max()) > 0.0 or float(np.abs(m_).max()) == 0.0, nme       # a shadow, not a copy of the model
    for k_ in g.keys(f"{last}/emabuf/"):
        name = k_.split("/", 2)[2]
        assert torch.equal(alg.ema_model.buffers[name], alg.model.buffers[name]), name


def test_full_size_step_properties_wrn():
    """BASELINE.json configs[0] at its OWN size: SRPseudoLabel on wrn_28_2, 64 labelled + 64 unlabelled 32x32 images, 100 classes, SGD-Nesterov
    (config/classic_cv/pseudolabel/pseudolabel_cifar100_400_0.yaml + the SR keys, feature_dim 128), steady SR regime K = sr_decay() = 8.  The CPU
    oracle cannot step this in seconds, so parity goes through properties:
      * integer work bit-exact: every pass's FixedThresholdingHook mask == (max_prob >= p_cutoff) on the engine's own probabilities (numpy),
        with a cut-off placed so that the masks are mixed; mask2 == (reward >= per-pass mean);
      * the K frozen-statistics passes over the same x_ulb_w give the same logits (WRN has no stochastic layer): every pass's mask / pseudo
        label equal pass 0's -- the reference computes them K + 1 times and so does the engine (K + 2 model calls);
      * BatchNorm running statistics are moved by the LABELLED forward only (Bn_Controller, srpseudolabel.py:96-110): after the step they equal
        what the fp32 oracle's update_stats forward of x_lb alone produces, num_batches_tracked == 1;
      * logits of the labelled forward and of the unlabelled forward against the fp32 oracle (train-mode batch statistics);
      * finite losses, util_ratio = mean(mask0), reproducible step."""
    import argparse
    from semireward_amd.algorithms import get_algorithm
    from semireward_amd.utils import synth
    C, Bl, Bu = 100, 64, 64
    wcfg = W.WrnCfg(num_classes=C, **W.WRN_28_2)
    Fd = W.channels(wcfg)[3]
    params = synth_wrn_params(wcfg, 5)

    def make(p_cutoff):
        # the authored yaml of BASELINE configs[0] (configs/README.md) through the yaml loader; only the cut-off is the test's
        import os
        from semireward_amd import config as srconfig
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        args = srconfig.get_config(os.path.join(root, "configs", "classic_cv_srpseudolabel_cifar100_400_wrn_28_2.yaml"),
                                   overrides=dict(gpu=0, rank=0, world_size=1, distributed=False, p_cutoff=p_cutoff))
        assert (args.algorithm, args.net, args.num_classes, args.batch_size, args.feature_dim, args.optim) == ("srpseudolabel", "wrn_28_2", C, Bl, Fd, "SGD")
        alg = get_algorithm(args, wrn.wrn_28_2)
        sd = {k: torch.from_numpy(v) for k, v in params.items()}
        alg.model.load_state_dict(sd)
        alg.ema_model.load_state_dict(sd)
        alg.it = 200001                         # sr_decay() = max(8, 1 + 1048576 / it) = 8; not a rewarder-update step
        alg.optimizer.sched_step = alg.it
        return alg
    b = synth.synth_batch(300, Bl, Bu, 32, C, 50000)
    batch = {k: torch.from_numpy(v) for k, v in b.items()}

    def run(p_cutoff):
        alg = make(p_cutoff)
        calls = []
        ff = alg.model.forward_features
        alg.model.forward_features = lambda *a, **k: (calls.append((k.get("update_stats", True), k.get("passes", 1))), ff(*a, **k))[1]
        alg.trace = {}
        out, log = alg.train_step(**alg.process_batch(**batch))
        torch.cuda.synchronize()
        return alg, out, log, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in alg.trace.items()}, calls
    alg, out, log, tr, calls = run(0.95)
    K = tr["K"]
    # model(x_lb) (moves the running statistics) and the K + 1 forwards of model(x_ulb_w) under frozen running statistics -- K data_generator
    # passes and the pass the backward belongs to -- are K + 2 statistics groups sharing their launches (WideResNet.forward_passes)
    assert K == 8 and calls == [(True, K + 2)]
    mp = tr["max_probs"].cpu().numpy().reshape(K + 1, Bu)
    mi = tr["pseudo"].cpu().numpy().reshape(K + 1, Bu)
    assert all(float(m.sum()) == 0.0 for m in tr["masks"]) and mp.max() < 0.95      # random-init model: nothing reaches 0.95
    cut = float(np.median(mp[0]))
    alg2, out2, log2, tr2, _ = run(cut)
    mp2 = tr2["max_probs"].cpu().numpy().reshape(K + 1, Bu)
    assert np.array_equal(mp2, mp)                                                   # same state, same inputs -> same probabilities (reproducible)
    masks = np.stack([m.cpu().numpy() for m in tr2["masks"]])
    assert np.array_equal(masks, (mp2 >= np.float32(cut)).astype(np.float32)) and 0.2 < masks.mean() < 0.8
    for k in range(1, K + 1):                                                         # deterministic repeats of pass 0
        assert np.array_equal(mp2[k], mp2[0]) and np.array_equal(mi[k], mi[0]), k
    r = tr2["reward"].cpu().numpy().reshape(K, Bu)
    assert np.array_equal(tr2["mask2"].cpu().numpy().reshape(K, Bu), (r >= r.mean(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32))
    assert 0.0 <= r.min() and r.max() <= 1.0
    assert float(log2["train/util_ratio"]) == pytest.approx(float(masks[0].mean()), abs=1e-7)
    for k_ in ("sup_loss", "unsup_loss", "total_loss"):
        assert np.isfinite(float(log2["train/" + k_])) and np.isfinite(float(log["train/" + k_]))
    assert float(log2["train/unsup_loss"]) > 0.0 and float(log["train/unsup_loss"]) == 0.0
    # ---- fp32 oracle: the labelled forward (statistics move) and the frozen unlabelled forward
    Pt = {k: torch.from_numpy(v) for k, v in params.items()}
    BUF = W.init_buffers(wcfg)
    with torch.no_grad():
        o_lb = W.wrn_forward(Pt, BUF, batch["x_lb"], wcfg, train=True, update_stats=True)
        snap = {k: v.clone() for k, v in BUF.items()}
        o_u = W.wrn_forward(Pt, BUF, batch["x_ulb_w"], wcfg, train=True, update_stats=False)
    assert all(torch.equal(BUF[k], snap[k]) for k in BUF)
    TOL = 2.5e-2
    assert rel(out["feat"]["x_lb"].cpu(), o_lb["feat"].numpy()) < TOL and rel(out["feat"]["x_ulb_w"].cpu(), o_u["feat"].numpy()) < TOL
    pr = torch.softmax(o_u["logits"], -1).max(-1)[0].numpy()
    assert float(np.abs(mp[0] - pr).max()) < 2e-2 * float(pr.max())
    fresh = W.init_buffers(wcfg)
    for k_, v in alg.model.buffers.items():
        if k_.endswith("num_batches_tracked"):
            assert int(v) == 1, k_
        else:
            # running_var starts at 1 (momentum 0.001 keeps it there to 1e-3); running_mean starts at 0, so after one update it IS
            # 0.001 x the batch mean of a bf16-operand forward: compared through the update
            if k_.endswith("running_var"):
                assert rel(v.cpu(), BUF[k_].numpy()) < 2e-4, k_
            assert rel(v.cpu().numpy() - fresh[k_].numpy(), (BUF[k_] - fresh[k_]).numpy()) < 6e-2, k_     # the update itself
    # the optimizer step that follows moves every parameter (SGD-Nesterov, one launch) and keeps the EMA shadow a shadow
    p0 = alg2.model.flat.clone()
    alg2.out_dict, alg2.log_dict = out2, log2
    alg2.call_hook("after_train_step")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(alg2.model.flat).all()) and float((alg2.model.flat - p0).abs().max()) > 0.0
    assert float((alg2.ema_model.flat - p0).abs().max()) < float((alg2.model.flat - p0).abs().max())


@pytest.mark.parametrize("B,H,Cin,Cout,ks,stride,mode,resid", [
    (3, 8, 16, 32, 3, 1, 0, False), (2, 8, 32, 32, 3, 1, 0, True), (2, 8, 32, 64, 3, 2, 2, False), (2, 8, 32, 64, 1, 2, 2, False),
    (4, 4, 128, 128, 3, 1, 0, True), (2, 8, 64, 128, 3, 2, 1, False), (5, 6, 16, 16, 3, 1, 0, True), (64, 32, 32, 32, 3, 1, 0, True),
    # the layers that take the LDS-tiled kernel (output rows >= 16 pixels): every (Cin, Cout, stride, kernel size, input mode) of WRN-28-2's
    # 32 x 32 and 16 x 16 stages, with enough images that both block sizes (128 / 256 pixels per workgroup) occur
    (3, 32, 16, 32, 3, 1, 0, False), (3, 32, 16, 32, 1, 1, 2, False), (5, 32, 32, 64, 3, 2, 0, False), (5, 32, 32, 64, 1, 2, 2, False),
    (6, 16, 64, 64, 3, 1, 0, True), (6, 16, 64, 64, 3, 1, 1, True), (300, 16, 64, 64, 3, 1, 0, True), (40, 32, 32, 32, 3, 1, 2, False),
    (7, 8, 128, 128, 3, 1, 0, True), (8, 8, 128, 128, 3, 1, 0, True), (7, 16, 64, 128, 3, 2, 0, False), (6, 16, 64, 128, 3, 2, 0, False), (7, 16, 64, 128, 1, 2, 2, False)])
def test_fused_conv_equals_the_unfused_chain(B, H, Cin, Cout, ks, stride, mode, resid):
    """srhip_wrn_conv_bn (statistics of the input BatchNorm folded from its accumulator, BatchNorm + LeakyReLU on load, implicit GEMM, residual,
    sums of the output into the next accumulator) against the chain it replaces -- srhip_bn_fwd -> srhip_im2col -> srhip_gemm_nt -> srhip_bn_fwd
    statistics -- on the same inputs: the bf16 activation is the same arithmetic, so the outputs differ only by the fp32 summation order of
    the K axis; published mean / invstd / running statistics and the output's sums agree to fp64-sum round-off."""
    rng = np.random.Generator(np.random.PCG64(B * 1000 + Cin + Cout + ks))
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(DEV).contiguous()   # noqa: E731
    rows_in = B * H * H
    x = T(rng.standard_normal((rows_in, Cin)) * 1.3 + 0.2)
    gam, bet = T(1.0 + 0.1 * rng.standard_normal(Cin)), T(0.1 * rng.standard_normal(Cin))
    rm, rv = T(0.1 * rng.standard_normal(Cin)), T(1.0 + 0.2 * rng.random(Cin))
    Wt = T(rng.standard_normal((Cout, Cin, ks, ks)) / np.sqrt(Cin * ks * ks))
    K, Kp = Cin * ks * ks, (Cin * ks * ks + 31) // 32 * 32
    pad = ks // 2
    Ho = (H + 2 * pad - ks) // stride + 1
    rows = B * Ho * Ho
    Wb, WbT = torch.zeros(Cout, Kp, dtype=torch.bfloat16, device=DEV), torch.zeros(Kp, Cout, dtype=torch.bfloat16, device=DEV)
    ops.conv_weight_prep(Wt.reshape(-1), Wb, WbT, Cout, Cin, ks, Kp)
    res = T(rng.standard_normal((rows, Cout))) if resid else None
    ws = torch.zeros(ops.bn_ws_doubles(), dtype=torch.float64, device=DEV)
    # ---- the unfused chain
    mean, invstd = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
    act = torch.empty(rows_in, Cin, dtype=torch.bfloat16, device=DEV)
    if mode == 2:
        ops.cast_f32_bf16(x, act, rows_in * Cin)
        stats = None
    else:
        rm_, rv_ = rm.clone(), rv.clone()
        ops.bn_fwd(x, gam, bet, 1e-5, 0.1, 0.001, mode == 0, False, rm_, rv_, mean, invstd, act, None, ws, rows_in, Cin)
        stats = (mean, invstd) if mode == 0 else (rm, rv)
    col = torch.empty(rows, Kp, dtype=torch.bfloat16, device=DEV)
    ops.im2col(act, col, B, H, H, Cin, ks, stride, Kp)
    want = torch.empty(rows, Cout, device=DEV)
    if res is None:
        ops.gemm_nt(ops.EPI_F32, col, Wb, want, rows, Cout, Kp)
    else:
        ops.gemm_nt(ops.EPI_RESID_F32, col, Wb, want, rows, Cout, Kp, aux_in=res, ldaux=Cout)
    g2, b2 = T(np.ones(Cout)), T(np.zeros(Cout))
    wm, wi = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    rm2w, rv2w = T(0.05 * rng.standard_normal(Cout)), T(1.0 + 0.1 * rng.random(Cout))
    rm2, rv2 = rm2w.clone(), rv2w.clone()
    ops.bn_fwd(want, g2, b2, 1e-5, 0.1, 0.001, True, True, rm2w, rv2w, wm, wi, None, torch.empty(rows, Cout, device=DEV), ws, rows, Cout)
    # ---- one launch: the input BatchNorm's statistics arrive as accumulator copies (mode 3; filled here by a 1x1 identity-free trick: the
    # statistics pass of x itself), the output's sums leave as accumulator copies
    assert ops.wrn_conv_supported(Cin, Cout, ks)
    acc_in = torch.zeros(ops.bn_acc_doubles(Cin), dtype=torch.float64, device=DEV)
    xs = x.double()
    acc_in.view(16, 2 * Cin)[3, :Cin] = xs.sum(0)                      # (any split over the 16 copies folds to the same sums)
    acc_in.view(16, 2 * Cin)[11, Cin:] = (xs * xs).sum(0)
    acc_out = torch.zeros(ops.bn_acc_doubles(Cout), dtype=torch.float64, device=DEV)
    got = torch.full((rows, Cout), 7.0, device=DEV)
    pm, pi = torch.full((Cin,), 9.0, device=DEV), torch.full((Cin,), 9.0, device=DEV)
    rmp, rvp = rm.clone(), rv.clone()
    use_mode = 3 if mode == 0 else mode
    gg, bb = (gam, bet) if mode != 2 else (None, None)
    ops.wrn_conv_bn(x, use_mode, stats if mode == 1 else None, acc_in, gg, bb, 1e-5, 0.1, Wb, res, got, B, H, H, Cin, Cout, ks, stride, Kp,
                    publish=(pm, pi), running=(rmp, rvp), momentum=0.001, update_running=True, acc_out=acc_out)
    torch.cuda.synchronize()
    assert rel(got.cpu(), want.cpu().numpy()) < 3e-6
    # the published statistics of the INPUT BatchNorm == bn_fwd's (mode 0 computed them above; otherwise from torch), running update included
    xm, xv = xs.mean(0), xs.var(0, unbiased=False)
    np.testing.assert_allclose(pm.cpu().numpy(), xm.float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pi.cpu().numpy(), (1.0 / torch.sqrt(xv + 1e-5)).float().cpu().numpy(), rtol=1e-5)
    if mode == 0:
        assert torch.equal(pm, mean) and torch.equal(pi, invstd)
    np.testing.assert_allclose(rmp.cpu().numpy(), (0.999 * rm.double() + 0.001 * xm).float().cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rvp.cpu().numpy(), (0.999 * rv.double() + 0.001 * xs.var(0, unbiased=True)).float().cpu().numpy(), rtol=1e-6, atol=1e-7)
    # the sums of the output, folded: the statistics bn_fwd computes on the same tensor
    tot = acc_out.view(16, 2 * Cout).sum(0)
    m_ = tot[:Cout] / rows
    v_ = tot[Cout:] / rows - m_ * m_
    np.testing.assert_allclose(m_.float().cpu().numpy(), wm.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose((1.0 / torch.sqrt(v_.float() + 1e-5)).cpu().numpy(), wi.cpu().numpy(), rtol=2e-5)
    # without statistics or publishing (the shortcut convolution; eval mode): same output, nothing else touched
    got2 = torch.empty(rows, Cout, device=DEV)
    ops.wrn_conv_bn(x, use_mode, stats if mode == 1 else None, acc_in if mode == 0 else None, gg, bb, 1e-5, 0.1, Wb, res, got2, B, H, H, Cin, Cout,
                    ks, stride, Kp)
    assert torch.equal(got2, got)
    if mode == 0:                                                      # mode 0 (statistics handed over as mean / invstd) == mode 3
        ops.wrn_conv_bn(x, 0, (mean, invstd), None, gam, bet, 1e-5, 0.1, Wb, res, got2, B, H, H, Cin, Cout, ks, stride, Kp)
        assert torch.equal(got2, got)
    # bn_stats alone == the statistics half of bn_fwd; bn_act == its bf16 activation (the backward's recomputed operand), bit for bit
    sm, si = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    ops.bn_stats(want, 1e-5, 0.001, False, None, None, sm, si, ws, rows, Cout)
    assert torch.equal(sm, wm) and torch.equal(si, wi)
    act2 = torch.empty_like(act)
    ops.bn_act(x, stats, gam if mode != 2 else None, bet if mode != 2 else None, 1e-5, 0.1, mode, act2, rows_in, Cin)
    assert torch.equal(act2.view(torch.int16), act.view(torch.int16))


def test_passes_sharing_their_launches_equal_separate_forwards():
    """WideResNet.forward_passes (srhip_wrn_conv_bn_passes / srhip_wrn_head_passes: G forwards of one batch, each its own BatchNorm statistics
    group, one launch per convolution) against G separate forward_features calls under frozen running statistics: logits and features of every
    pass bit for bit (the statistics are fp64 sums rounded to float: order-independent), the kept activations / statistics of the last pass
    equal to a saved single forward's, and the backward from them gives the same gradients.  Running statistics do not move."""
    torch.manual_seed(0)
    m = wrn.WideResNet(num_classes=10, depth=10, widen_factor=2, first_stride=1, device=DEV)
    m.init_weights(seed=3)
    m.refresh_operands()
    m.train()
    rng = np.random.Generator(np.random.PCG64(21))
    x = torch.from_numpy(rng.standard_normal((6, 3, 16, 16)).astype(np.float32)).to(DEV)
    dl = torch.from_numpy((rng.standard_normal((6, 10)) * 0.1).astype(np.float32)).to(DEV)
    run0 = {k: v.clone() for k, v in m.buffers.items()}
    G = 3
    lg, ft, ctx = m.forward_passes(x, G, tag="ulb")
    m.zero_grad(); m.backward(ctx, dl); g_group = m.grad.clone()
    singles = [m.forward_features(x, save=False, update_stats=False, tag="one%d" % i) for i in range(G)]
    lg1, ft1, ctx1 = m.forward_features(x, save=True, update_stats=False, tag="saved")
    m.zero_grad(); m.backward(ctx1, dl); g_single = m.grad.clone()
    torch.cuda.synchronize()
    assert lg.shape == (G * 6, 10) and ft.shape[0] == G * 6
    for i, (l1, f1, _) in enumerate(singles):
        assert torch.equal(lg[i * 6:(i + 1) * 6], l1) and torch.equal(ft[i * 6:(i + 1) * 6], f1), i
    assert torch.equal(lg[(G - 1) * 6:], lg1) and torch.equal(ctx.feat, ctx1.feat)
    for a, b in zip(ctx.blocks, ctx1.blocks):
        assert torch.equal(a["x"], b["x"]) and torch.equal(a["c1"], b["c1"])
        assert all(torch.equal(u, v) for u, v in zip(a["st1"], b["st1"])) and all(torch.equal(u, v) for u, v in zip(a["st2"], b["st2"]))
    assert torch.equal(ctx.final["x"], ctx1.final["x"]) and all(torch.equal(u, v) for u, v in zip(ctx.final["st"], ctx1.final["st"]))
    assert float((g_group - g_single).abs().max()) <= 1e-6 * float(g_single.abs().max())          # (fp32 atomics of the filter gradients: unordered sums)
    assert all(torch.equal(run0[k], m.buffers[k]) for k in run0)                                    # frozen: no running statistic moved
    # ... with the labelled batch as pass 0 (the one call that moves the running statistics): == forward_saved(x_lb, update_stats=True) followed
    # by the frozen passes, running statistics and num_batches_tracked included
    xl = torch.from_numpy(rng.standard_normal((6, 3, 16, 16)).astype(np.float32)).to(DEV)
    import copy
    m2 = wrn.WideResNet(num_classes=10, depth=10, widen_factor=2, first_stride=1, device=DEV)
    m2.load_state_dict(m.state_dict()); m2.train()
    lga, fta, (cf, cl) = m.forward_passes(x, G, tag="ulb", first_img=xl)
    m.zero_grad(); m.backward(cf, dl); m.backward(cl, dl); ga = m.grad.clone()
    lgl, ftl, cl2 = m2.forward_features(xl, save=True, update_stats=True, tag="lb")
    lgu, ftu, cu2 = m2.forward_passes(x, G, tag="ulb")
    m2.zero_grad(); m2.backward(cl2, dl); m2.backward(cu2, dl); gb = m2.grad.clone()
    torch.cuda.synchronize()
    assert torch.equal(lga, torch.cat((lgl, lgu))) and torch.equal(fta, torch.cat((ftl, ftu)))
    assert all(torch.equal(m.buffers[k], m2.buffers[k]) for k in m.buffers) and not all(torch.equal(run0[k], m.buffers[k]) for k in run0)
    assert float((ga - gb).abs().max()) <= 1e-6 * float(gb.abs().max())


def test_backward_is_run_to_run_deterministic_with_sliced_filter_gradients():
    """The filter gradients of the 32 x 32 stage sum over 8 192+ pixels in slices of 4 096 (nets/wrn.py _DW_SPLIT): the slices write scratch
    slabs that one reduce launch adds in a fixed order (SRHIP_TN_OVERWRITE), so two backwards from the same activations give the same bits
    (through fp32 atomics the sums depended on the order the slices retired in)."""
    torch.manual_seed(0)
    m = wrn.WideResNet(num_classes=10, depth=10, widen_factor=2, first_stride=1, device=DEV)
    m.init_weights(seed=5)
    m.refresh_operands()
    m.train()
    rng = np.random.Generator(np.random.PCG64(33))
    x = torch.from_numpy(rng.standard_normal((12, 3, 32, 32)).astype(np.float32)).to(DEV)          # 12 288 pixels in the first stage: 3 slices
    dl = torch.from_numpy((rng.standard_normal((12, 10)) * 0.1).astype(np.float32)).to(DEV)
    lg, ft, ctx = m.forward_features(x, save=True, update_stats=False, tag="det")
    grads = []
    for _ in range(3):
        m.zero_grad(); m.backward(ctx, dl)
        torch.cuda.synchronize()
        grads.append(m.grad.clone())
    ent = m._buf_cache[("tn_desc", "det")][1][0]
    assert hasattr(ent, "reduce") and ent.reduce[1] >= 1, "the table of this batch has no sliced problem: the test does not reach the slabs"
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    assert float(grads[0].abs().max()) > 0


@pytest.mark.parametrize("B,HW2,C,K", [(64, 64, 128, 100), (5, 16, 64, 10), (3, 4, 256, 7), (2, 9, 32, 3)])
def test_network_tail_in_one_launch(B, HW2, C, K):
    """srhip_wrn_head (final BatchNorm + LeakyReLU + average pooling + classifier, statistics folded from the accumulator srhip_bn_accumulate
    filled) against torch: batch statistics with the running update, then eval mode from the running statistics; srhip_bn_fold agrees with
    srhip_bn_stats on the same tensor."""
    import torch.nn.functional as F
    rng = np.random.Generator(np.random.PCG64(B * 7 + C))
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(DEV).contiguous()   # noqa: E731
    rows = B * HW2
    x = T(rng.standard_normal((rows, C)) * 1.5 + 0.4)
    gam, bet = T(1.0 + 0.1 * rng.standard_normal(C)), T(0.1 * rng.standard_normal(C))
    rm0, rv0 = T(0.1 * rng.standard_normal(C)), T(1.0 + 0.2 * rng.random(C))
    Wc, bc = T(rng.standard_normal((K, C)) / np.sqrt(C)), T(0.1 * rng.standard_normal(K))
    # torch: [B, C, HW2] layout for batch_norm
    xt = x.view(B, HW2, C).permute(0, 2, 1).contiguous()
    rm, rv = rm0.clone(), rv0.clone()
    yt = F.leaky_relu(F.batch_norm(xt, rm, rv, gam, bet, True, 0.001, 1e-3), 0.1)
    feat_want = yt.mean(dim=2)
    logits_want = feat_want @ Wc.t() + bc
    acc = torch.zeros(ops.bn_acc_doubles(C), dtype=torch.float64, device=DEV)
    ops.bn_accumulate(x, acc, rows, C)
    feat, logits = torch.empty(B, C, device=DEV), torch.empty(B, K, device=DEV)
    pm, pi = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rmd, rvd = rm0.clone(), rv0.clone()
    ops.wrn_head(x, 3, None, acc, gam, bet, 1e-3, 0.1, Wc, bc, feat, logits, B, HW2, C, K, publish=(pm, pi), running=(rmd, rvd), momentum=0.001,
                 update_running=True)
    torch.cuda.synchronize()
    assert rel(feat.cpu(), feat_want.cpu().numpy()) < 3e-6 and rel(logits.cpu(), logits_want.cpu().numpy()) < 5e-6
    np.testing.assert_allclose(rmd.cpu().numpy(), rm.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv.cpu().numpy(), rtol=1e-6, atol=1e-7)
    # the published statistics == srhip_bn_fold == srhip_bn_stats of the same tensor
    fm, fi = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    tot = torch.empty(2 * C, dtype=torch.float64, device=DEV)
    ops.bn_fold(acc, rows, 1e-3, 0.0, False, None, None, fm, fi, tot, C)
    ws = torch.zeros(ops.bn_ws_doubles(), dtype=torch.float64, device=DEV)
    sm, si = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_stats(x, 1e-3, 0.0, False, None, None, sm, si, ws, rows, C)
    assert torch.equal(fm, pm) and torch.equal(fi, pi)
    np.testing.assert_allclose(fm.cpu().numpy(), sm.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(fi.cpu().numpy(), si.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(tot[:C].cpu().numpy(), x.double().sum(0).cpu().numpy(), rtol=1e-12)
    # eval mode: running statistics
    ye = F.leaky_relu(F.batch_norm(xt, rm, rv, gam, bet, False, 0.001, 1e-3), 0.1).mean(dim=2)
    ops.wrn_head(x, 1, (rmd, rvd), None, gam, bet, 1e-3, 0.1, Wc, bc, feat, logits, B, HW2, C, K)
    assert rel(feat.cpu(), ye.cpu().numpy()) < 3e-6 and rel(logits.cpu(), (ye @ Wc.t() + bc).cpu().numpy()) < 5e-6


def test_grouped_filter_prep_and_unpad_equal_the_single_launches():
    """srhip_conv_weight_prep_grouped / srhip_add_unpad_grouped (every convolution of a network in one launch) against one
    srhip_conv_weight_prep / srhip_add_unpad per convolution: bit for bit."""
    rng = np.random.Generator(np.random.PCG64(3))
    shapes = [(16, 3, 3), (32, 16, 3), (32, 16, 1), (64, 32, 3), (128, 128, 3)]
    Wf, single, grouped, pads, g1, g2 = [], [], [], [], [], []
    for Cout, C, ks in shapes:
        K, Kp = C * ks * ks, (C * ks * ks + 31) // 32 * 32
        w = torch.from_numpy(rng.standard_normal(Cout * K).astype(np.float32)).to(DEV)
        Wf.append(w)
        single.append((torch.zeros(Cout, Kp, dtype=torch.bfloat16, device=DEV), torch.zeros(Kp, Cout, dtype=torch.bfloat16, device=DEV)))
        grouped.append((torch.full((Cout, Kp), 3.0, dtype=torch.bfloat16, device=DEV), torch.full((Kp, Cout), 3.0, dtype=torch.bfloat16, device=DEV)))
        pads.append(torch.from_numpy(rng.standard_normal((Cout, Kp)).astype(np.float32)).to(DEV))
        g0 = torch.from_numpy(rng.standard_normal(Cout * K).astype(np.float32)).to(DEV)
        g1.append(g0.clone()); g2.append(g0.clone())
    for (Cout, C, ks), w, (a, b), p_, g in zip(shapes, Wf, single, pads, g1):
        Kp = a.shape[1]
        ops.conv_weight_prep(w, a, b, Cout, C, ks, Kp)
        ops.add_unpad(p_, g, Cout, C, ks, Kp)
    d = ops.make_conv_desc([(w, a, b, Cout, C, ks, a.shape[1]) for (Cout, C, ks), w, (a, b) in zip(shapes, Wf, grouped)], DEV,
                           lambda Cout, C, kk, Kpad: Cout * Kpad)
    ops.conv_weight_prep_grouped(*d)
    d2 = ops.make_conv_desc([(p_, g, None, Cout, C, ks, p_.shape[1]) for (Cout, C, ks), p_, g in zip(shapes, pads, g2)], DEV,
                            lambda Cout, C, kk, Kpad: Cout * C * kk)
    ops.add_unpad_grouped(*d2)
    torch.cuda.synchronize()
    for (a, b), (c, e), x, y in zip(single, grouped, g1, g2):
        assert torch.equal(a.view(torch.int16), c.view(torch.int16)) and torch.equal(b.view(torch.int16), e.view(torch.int16)) and torch.equal(x, y)


@pytest.mark.parametrize("B,H,C,ks,stride,mode", [(3, 8, 32, 3, 1, 0), (2, 8, 16, 3, 2, 0), (2, 8, 64, 1, 2, 2), (4, 4, 128, 3, 1, 2), (64, 32, 32, 3, 1, 0)])
def test_im2col_with_batchnorm_on_load_equals_act_then_im2col(B, H, C, ks, stride, mode):
    """srhip_im2col_bn (the backward's filter-gradient operand straight from the fp32 tensor in front of the BatchNorm) == srhip_bn_act followed
    by srhip_im2col, bit for bit."""
    rng = np.random.Generator(np.random.PCG64(B + C + ks))
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(DEV).contiguous()   # noqa: E731
    rows = B * H * H
    x = T(rng.standard_normal((rows, C)) * 1.4 + 0.1)
    gam, bet = T(1.0 + 0.1 * rng.standard_normal(C)), T(0.1 * rng.standard_normal(C))
    mean, isd = T(0.1 * rng.standard_normal(C)), T(1.0 + 0.2 * rng.random(C))
    Kp = (C * ks * ks + 31) // 32 * 32
    pad = ks // 2
    Ho = (H + 2 * pad - ks) // stride + 1
    act = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV)
    st = (mean, isd) if mode == 0 else None
    ops.bn_act(x, st, gam if mode == 0 else None, bet if mode == 0 else None, 1e-5, 0.1, mode, act, rows, C)
    want = torch.empty(B * Ho * Ho, Kp, dtype=torch.bfloat16, device=DEV)
    ops.im2col(act, want, B, H, H, C, ks, stride, Kp)
    got = torch.full((B * Ho * Ho, Kp), 5.0, dtype=torch.bfloat16, device=DEV)
    ops.im2col_bn(x, st, gam if mode == 0 else None, bet if mode == 0 else None, 0.1, mode, got, B, H, H, C, ks, stride, Kp)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
