"""/root/reference/train.py:394-431 replayed on the GPU with the real SRFlexMatch step (tests/_driver_replay.py holds the sequence and the
stand-ins for the reference's CPU input pipeline): get_algorithm -> count_parameters(model.model) -> send_model_cuda (the engine's, see
INTEGRATION.md) -> train() over loader_dict['train_lb'/'train_ulb'] -> save_model -> a second process-like instance resumes through
``args.resume`` + load_model (train.py:404-409) and continues."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import _driver_replay as R                                             # noqa: E402
from semireward_amd.algorithms import get_algorithm                    # noqa: E402
from semireward_amd.core.utils import count_parameters, send_model_cuda  # noqa: E402
from semireward_amd.nets import get_net_builder                        # noqa: E402


def _args(**kw):
    d = dict(algorithm="srflexmatch", net="vit_tiny_test", net_from_name=False, num_classes=10, num_train_iter=6, epoch=2, ema_m=0.0,
             ulb_loss_ratio=1.0, use_cat=True, amp=False, lr=5e-4, weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=0, optim="AdamW", T=0.5,
             p_cutoff=0.95, hard_label=True, thresh_warmup=True, N_k=2, start_timing=2, feature_dim=128, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99,
             num_eval_iter=0, num_log_iter=3, gpu=0, rank=0, world_size=1, distributed=False, dataset="stand_in", num_labels=12, data_dir="./data",
             include_lb_to_ulb=True, batch_size=4, uratio=1, eval_batch_size=4, num_workers=1, train_sampler="RandomSampler", img_size=8,
             stand_in_ulb=40, resume=False, save_name="run", save_dir="./saved_models", data_functions=(R.get_dataset, R.get_data_loader))
    d.update(kw)
    return argparse.Namespace(**d)


def test_main_worker_sequence_with_the_real_step(tmp_path):
    from semireward_amd.nets import vit
    builder = vit.vit_tiny_test                              # (train.py:389 get_net_builder(args.net, ...); the test net is not a registry name)
    args = _args()
    R.CALLS.clear()
    model, n_params, _ = R.main_worker_tail(args, get_algorithm, builder, count_parameters, send_model_cuda)
    assert n_params == model.model.numel == sum(p.numel() for p in model.model.parameters() if p.requires_grad)
    assert args.ulb_dest_len == 40                           # set_dataset's side effect sized the FlexMatch table (hook built after it)
    assert model.hooks_dict["MaskingHook"].selected_label.numel() == 40
    assert model.it == 6 and model.optimizer.step_count == 6 and np.isfinite(float(model.log_dict["train/total_loss"]))
    assert [c[0] for c in R.CALLS] == ["get_dataset"] + ["get_data_loader"] * 3
    # evaluate() on the loader set_data_loader built (algorithmbase.py:377-457 keys)
    ev = model.evaluate("eval")
    assert {"eval/loss", "eval/top-1-acc", "eval/F1"} <= set(ev)
    # checkpoint -> resume (train.py:404-409): a fresh instance continues from the saved iteration
    model.save_model("latest_model.pth", str(tmp_path))
    args2 = _args(resume=True, num_train_iter=8, epoch=4)      # (the saved epoch counter is 2: two more epochs to go)
    m2, _, _ = R.main_worker_tail(args2, get_algorithm, builder, count_parameters, send_model_cuda, load_path=os.path.join(str(tmp_path), "latest_model.pth"))
    # (get_save_dict stores it + 1 as the reference does -- it saves from a hook BEFORE the increment; saved after train() that skips one number)
    assert m2.it == 8 and m2.optimizer.step_count == 7
    assert not torch.equal(m2.model.flat, model.model.flat)


def test_registry_names_resolve_like_the_reference():
    """train.py:389 ``get_net_builder(args.net, args.net_from_name)`` and :394 ``get_algorithm``: the yaml's ``net`` / ``algorithm`` keys."""
    b = get_net_builder("vit_small_patch2_32", False)
    m = b(num_classes=100)
    assert count_parameters(m) == 21436900 and send_model_cuda(argparse.Namespace(gpu=0, distributed=False), m) is m
