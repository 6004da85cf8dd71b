"""Row n1 of SURVEY 8(f) and ParamUpdateHook's clip_grad (param_update.py:34-35) on the HIP engine, against vectors produced by the reference's
own EMA class / EMAHook call sequence / clip_grad_norm_ / get_optimizer (tests/golden/ema.npz, oracle/gen_golden.py:gen_ema):
  * the EMA shadow updated INSIDE the fused AdamW / SGD launch equals misc.py:152-155 after every step (ema_m 0.999), the BatchNorm buffers
    of the ema model are the model's (core/hooks/ema.py:23), and evaluate() reads the EMA weights;
  * clip_grad > 0: the global norm and the parameters after clipped steps;
  * a torch.optim-layout optimizer state (a reference checkpoint) loads into the flat engine state in the reference's group order."""
import argparse

import numpy as np
import pytest
import torch

from conftest import check_samp
from oracle import vit_ref as V
from oracle import wrn_ref as W
from semireward_amd.algorithms import get_algorithm
from semireward_amd.nets import vit, wrn
from semireward_amd.utils import synth

pytestmark = pytest.mark.gpu


def _args(**kw):
    base = dict(algorithm="srflexmatch", num_classes=10, num_train_iter=10, epoch=1, ema_m=0.999, ulb_loss_ratio=1.0, use_cat=True, amp=False,
                lr=5e-4, weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=2, optim="AdamW", T=0.5, p_cutoff=0.95, hard_label=True,
                thresh_warmup=True, ulb_dest_len=256, N_k=10, start_timing=5, feature_dim=128, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99, gpu=0,
                rank=0, world_size=1, distributed=False)
    base.update(kw)
    return argparse.Namespace(**base)


def _set_grads(model, grads, scale):
    for n, _ in model.names_shapes:
        model.view(n, model.grad).copy_(torch.from_numpy(grads[n]).to(model.device) * scale)


def _step_and_check(alg, g, tag, step, rtol_p, shapes_names):
    alg.out_dict, alg.log_dict = {}, {}
    alg.call_hook("after_train_step")            # ParamUpdateHook: fused optimizer + scheduler + zero_grad + EMA
    assert float(alg.model.grad.abs().max()) == 0.0
    for n in shapes_names:
        check_samp(alg.model.view(n).cpu().numpy(), g.samp(f"{tag}/model{step}/{n}"), rtol_p, 1e-7, f"{tag} model {step} {n}")
        gs = g.samp(f"{tag}/ema{step}/{n}")
        got = alg.ema_model.view(n).reshape(-1).cpu().numpy()[::gs["stride"]]
        # same three fp32 roundings as misc.py:154 on (almost) the same operands: a few ulp of the value, far below the 0.001-weighted
        # parameter movement the shadow has to pick up (checked right below)
        np.testing.assert_allclose(got, gs["sample"], rtol=3e-7, atol=1e-9, err_msg=f"{tag} ema {step} {n}")


@pytest.mark.parametrize("tag,clip", [("vit", 0.0), ("vit_clip", 0.05)])
def test_ema_and_clip_in_the_fused_adamw_step(golden, tag, clip):
    g = golden("ema")
    cfg = V.VitCfg(num_classes=10, **V.VIT_TINY_TEST)
    shapes = V.param_shapes(cfg)
    alg = get_algorithm(_args(clip_grad=clip), vit.vit_tiny_test)
    assert alg.ema_model is not alg.model and float(g["meta/ema_m"]) == alg.ema_m
    init = {k: torch.from_numpy(v) for k, v in synth.synth_params(shapes, 61).items()}
    alg.model.load_state_dict(init)
    alg.ema_model.load_state_dict(init)         # EMA.register() (misc.py:146-148): shadow = the initial parameters
    alg.optimizer.sched_step = 1                 # the generator stepped the scheduler once before the first update
    names = [n for n, _ in shapes]
    ema0 = alg.ema_model.flat.clone()
    for step in range(3):
        _set_grads(alg.model, synth.synth_params(shapes, 170 + step), 0.1)
        _step_and_check(alg, g, tag, step, 3e-6 if clip == 0.0 else 2e-5, names)
        if clip > 0:
            assert float(alg.optimizer.last_grad_norm[1]) == pytest.approx(float(g[f"{tag}/total_norm{step}"]), rel=2e-6)
            assert float(alg.optimizer.last_grad_norm[0]) == pytest.approx(clip / (float(g[f"{tag}/total_norm{step}"]) + 1e-6), rel=2e-6)
    # the shadow really moved, by ~ (1 - ema_m) of the parameter movement, and is neither the model nor the initial parameters
    moved_e = float((alg.ema_model.flat - ema0).abs().max())
    moved_p = float((alg.model.flat - ema0).abs().max())
    assert 0 < moved_e < 0.01 * moved_p
    # evaluate() runs on the EMA weights (ema.apply_shadow, algorithmbase.py:382): its logits are a forward of ema_model, not of model
    rng = np.random.Generator(np.random.PCG64(5))
    x = torch.from_numpy(rng.standard_normal((4, 3, cfg.img_size, cfg.img_size)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 10, size=(4,), dtype=np.int64))
    alg.model.view("head.weight").mul_(1.5)      # make the two weight sets visibly different
    alg.model.refresh_operands()
    out = alg.evaluate(loader=[{"x_lb": x, "y_lb": y}], return_logits=True)
    alg.ema_model.eval()
    lg_e, _, _ = alg.ema_model.forward_features(x.cuda(), None, None, save=False)
    alg.model.eval()
    lg_m, _, _ = alg.model.forward_features(x.cuda(), None, None, save=False)
    np.testing.assert_allclose(out["eval/logits"], lg_e.cpu().numpy(), rtol=1e-6, atol=1e-6)
    assert float((lg_e - lg_m).abs().max()) > 1e-3


def test_ema_in_the_fused_sgd_step_with_batchnorm_buffers(golden):
    g = golden("ema")
    wcfg = W.WrnCfg(num_classes=10, **W.WRN_TINY_TEST)
    from oracle.gen_golden import synth_wrn_params
    alg = get_algorithm(_args(algorithm="srpseudolabel", optim="SGD", lr=0.03, momentum=0.9, weight_decay=5e-4, layer_decay=1.0,
                              num_warmup_iter=0, p_cutoff=0.95, unsup_warm_up=0.4, feature_dim=W.channels(wcfg)[3]), wrn.wrn_tiny_test)
    init = {k: torch.from_numpy(v) for k, v in synth_wrn_params(wcfg, 63).items()}
    alg.model.load_state_dict(init, strict=False)
    alg.ema_model.load_state_dict(init, strict=False)
    names = [n for n, _ in W.param_shapes(wcfg)]
    for step in range(3):
        _set_grads(alg.model, synth.synth_params(W.param_shapes(wcfg), 270 + step), 0.05)
        for n, c, _ in W.bn_names(wcfg):          # what the labelled forward of a real step does to the model's statistics
            alg.model.buffers[n + ".running_mean"].copy_(torch.from_numpy(g[f"wrn/buf{step}/{n}.running_mean"]))
            alg.model.buffers[n + ".running_var"].copy_(torch.from_numpy(g[f"wrn/buf{step}/{n}.running_var"]))
        _step_and_check(alg, g, "wrn", step, 3e-6, names)
        for k in g.keys(f"wrn/emabuf{step}/"):    # ema_model.load_state_dict(model.state_dict()) (ema.py:23): the buffers are the model's
            name = k.split("/", 2)[2]
            if not name.endswith("num_batches_tracked"):
                assert np.array_equal(alg.ema_model.buffers[name].cpu().numpy(), g[k]), k


def test_torch_layout_optimizer_state_loads_in_reference_group_order(golden):
    """A reference checkpoint stores torch.optim.AdamW.state_dict() (algorithmbase.py:466): running indices over the param groups that
    param_groups_layer_decay creates.  The group order / index -> name map comes from the reference run (ema.npz vit/opt/*)."""
    g = golden("ema")
    cfg = V.VitCfg(num_classes=10, **V.VIT_TINY_TEST)
    shapes = dict(V.param_shapes(cfg))
    names = [str(n) for n in g["vit/opt/names_by_index"]]
    sizes = [int(v) for v in g["vit/opt/group_sizes"]]
    rng = np.random.Generator(np.random.PCG64(9))
    state, groups, i = {}, [], 0
    want_m, want_v = {}, {}
    for sz in sizes:
        groups.append({"params": list(range(i, i + sz)), "lr": 1.0, "weight_decay": 0.0})
        for j in range(i, i + sz):
            n = names[j]
            want_m[n] = rng.standard_normal(shapes[n]).astype(np.float32)
            want_v[n] = rng.random(shapes[n]).astype(np.float32)
            state[j] = {"step": torch.tensor(7.0), "exp_avg": torch.from_numpy(want_m[n]), "exp_avg_sq": torch.from_numpy(want_v[n])}
        i += sz
    alg = get_algorithm(_args(ema_m=0.0), vit.vit_tiny_test)
    alg.optimizer.load_state_dict({"state": state, "param_groups": groups})
    assert alg.optimizer.step_count == 7
    for n in names:
        o = alg.model.offsets[n][0]
        ln = want_m[n].size
        assert np.array_equal(alg.optimizer.m[o:o + ln].cpu().numpy(), want_m[n].ravel()), n
        assert np.array_equal(alg.optimizer.v[o:o + ln].cpu().numpy(), want_v[n].ravel()), n


def test_amp_is_refused_not_ignored():
    with pytest.raises(NotImplementedError):
        get_algorithm(_args(amp=True), vit.vit_tiny_test)


def test_torch_layout_sgd_state_loads_in_reference_group_order(golden):
    """classic_cv checkpoints of the reference hold torch.optim.SGD.state_dict() (algorithmbase.py:466): param_groups = [no_decay, decay]
    (nets/utils.py:77-97), state[idx] = {'momentum_buffer'}.  Index -> name map and group sizes come from the reference run (ema.npz wrn/opt/*)."""
    g = golden("ema")
    wcfg = W.WrnCfg(num_classes=10, **W.WRN_TINY_TEST)
    shapes = dict(W.param_shapes(wcfg))
    names = [str(n) for n in g["wrn/opt/names_by_index"]]
    sizes = [int(v) for v in g["wrn/opt/group_sizes"]]
    assert [str(k) for k in g["wrn/opt/state_keys"]] == ["momentum_buffer"]
    rng = np.random.Generator(np.random.PCG64(19))
    state, groups, i, want = {}, [], 0, {}
    for sz, wd in zip(sizes, g["wrn/opt/group_wd"]):
        groups.append({"params": list(range(i, i + sz)), "lr": 0.03, "weight_decay": float(wd), "momentum": 0.9, "nesterov": True})
        for j in range(i, i + sz):
            want[names[j]] = rng.standard_normal(shapes[names[j]]).astype(np.float32)
            state[j] = {"momentum_buffer": torch.from_numpy(want[names[j]])}
        i += sz
    alg = get_algorithm(_args(algorithm="srpseudolabel", optim="SGD", lr=0.03, momentum=0.9, weight_decay=5e-4, layer_decay=1.0,
                              num_warmup_iter=0, p_cutoff=0.95, unsup_warm_up=0.4, feature_dim=W.channels(wcfg)[3]), wrn.wrn_tiny_test)
    alg.optimizer.load_state_dict({"state": state, "param_groups": groups})
    assert alg.optimizer.step_count == 1              # buffers exist: the next step must not re-initialise them with the gradient
    for n in names:
        o = alg.model.offsets[n][0]
        assert np.array_equal(alg.optimizer.buf[o:o + want[n].size].cpu().numpy(), want[n].ravel()), n
    # the engine's own weight-decay table agrees with the reference's two groups
    wd_of = {n: float(wd) for sz, wd, lo in zip(sizes, g["wrn/opt/group_wd"], np.cumsum([0] + sizes[:-1])) for n in names[lo:lo + sz]}
    tab = alg.optimizer.table.cpu().numpy().view([("end", "<i8"), ("wd", "<f4"), ("pad", "<f4")])
    for (n, _), row in zip(alg.model.names_shapes, tab):
        assert float(row["wd"]) == pytest.approx(wd_of[n]), n
    # an empty torch state (checkpoint written before the first step) leaves the first-step initialisation in place
    alg.optimizer.load_state_dict({"state": {}, "param_groups": groups})
    assert alg.optimizer.step_count == 0 and float(alg.optimizer.buf.abs().max()) == 0.0
