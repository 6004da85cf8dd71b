"""The reference DRIVER's call contract around the plugin (/root/reference/train.py:394-431, semilearn/core/utils/misc.py:39-75,
semilearn/core/algorithmbase.py:140-175, :185-228, :346-375), host side: what ``main_worker`` touches on the algorithm object and on
``model.model`` works on the engine classes -- ``count_parameters`` through ``parameters()`` / ``requires_grad``, ``send_model_cuda``'s argument
handling and errors, ``set_dataset`` / ``set_data_loader`` calling the reference's functions with the reference's arguments, ``train()`` zipping
``loader_dict['train_lb'/'train_ulb']`` into ``train_step(**process_batch(...))`` with the hook calls in the reference's order.  No GPU: the
backbone is built on the CPU device (plumbing only, no launch) and ``train_step`` is a recording stub; tests/test_gpu_driver_contract.py replays
the same sequence with the real SRFlexMatch step."""
import argparse

import pytest
import torch

import _driver_replay as R
from semireward_amd.core.algorithmbase import AlgorithmBase
from semireward_amd.core.hooks import Hook
from semireward_amd.core.utils import count_parameters, send_model_cuda
from semireward_amd.nets import vit


def _args(**kw):
    d = dict(algorithm="hostonly", num_classes=10, num_train_iter=5, epoch=2, ema_m=0.0, ulb_loss_ratio=1.0, use_cat=True, amp=False, lr=5e-4,
             num_eval_iter=0, num_log_iter=0, gpu=None, rank=0, world_size=1, distributed=False, dataset="stand_in", num_labels=12,
             data_dir="./data", include_lb_to_ulb=True, batch_size=4, uratio=2, eval_batch_size=4, num_workers=3, train_sampler="RandomSampler",
             img_size=8, stand_in_ulb=40, ulb_dest_len=-1, data_functions=(R.get_dataset, R.get_data_loader))
    d.update(kw)
    return argparse.Namespace(**d)


class _Rec(Hook):
    def __init__(self, log):
        self.log = log

    def before_run(self, alg): self.log.append("before_run")
    def before_train_epoch(self, alg): self.log.append("before_train_epoch")
    def before_train_step(self, alg): self.log.append("before_train_step")
    def after_train_step(self, alg): self.log.append("after_train_step")
    def after_train_epoch(self, alg): self.log.append("after_train_epoch")
    def after_run(self, alg): self.log.append("after_run")


class _HostOnly(AlgorithmBase):
    """The base class's own wiring with a backbone that is never launched (CPU device) and a recording step."""

    def set_model(self):
        return vit.VisionTransformer(vit.VitConfig(img_size=8, patch_size=2, embed_dim=128, depth=2, num_heads=2, num_classes=self.num_classes),
                                     device="cpu")

    def set_ema_model(self):
        return self.model

    def set_optimizer(self):
        return None, None

    def set_hooks(self):
        self.events = []
        self.register_hook(_Rec(self.events), "Rec")

    def _check_device_flags(self):
        pass

    def train_step(self, x_lb, y_lb, idx_ulb, x_ulb_w, x_ulb_s):
        self.events.append(("train_step", tuple(x_lb.shape), tuple(y_lb.shape), tuple(idx_ulb.shape), tuple(x_ulb_w.shape), tuple(x_ulb_s.shape)))
        return {"loss": torch.zeros(())}, {"train/total_loss": 0.0}


def test_count_parameters_and_module_surface():
    m = vit.VisionTransformer(vit.VitConfig(img_size=32, patch_size=2, embed_dim=384, depth=12, num_heads=6, num_classes=100), device="cpu")
    n = sum(int(torch.Size(s).numel()) for _, s in m.names_shapes)
    assert count_parameters(m) == n == 21436900          # ViT-S/2, 100 classes: the number train.py:396 logs for the headline config
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == n            # the reference's expression verbatim (misc.py:75)
    ps = list(m.parameters())
    assert [tuple(p.shape) for p in ps] == [tuple(s) for _, s in m.names_shapes]      # named_parameters() order and shapes
    m.grad[m.offsets["head.bias"][0]] = 3.0
    hb = ps[[n_ for n_, _ in m.names_shapes].index("head.bias")]
    assert hb.requires_grad and float(hb.grad[0]) == 3.0 and hb.data_ptr() == m.view("head.bias").data_ptr()      # views, not copies
    assert m.cuda is not None and m.to("cpu") is m and m.to(device="cpu") is m
    with pytest.raises(RuntimeError, match="cannot be moved"):
        m.to("cuda:0")
    with pytest.raises(RuntimeError, match="one numeric mode"):
        m.to(torch.float16)
    assert torch.nn.SyncBatchNorm.convert_sync_batchnorm(m) is m          # misc.py:55 walks named_children(): nothing to convert, same object


def test_send_model_cuda_argument_contract(monkeypatch):
    m = vit.VisionTransformer(vit.VitConfig(img_size=8, patch_size=2, embed_dim=128, depth=2, num_heads=2, num_classes=10), device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(Exception, match="ONLY GPU TRAINING IS SUPPORTED"):          # misc.py:40-41, same message
            send_model_cuda(_args(gpu=0), m)
    # the branch logic with the device calls stubbed out
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    seen = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: seen.append(d))
    monkeypatch.setattr(type(m), "cuda", lambda self, device=None: (seen.append(("cuda", device)), self)[1])
    a = _args(gpu=3, distributed=True, batch_size=64)
    assert send_model_cuda(a, m) is m and a.batch_size == 8 and seen == [3, ("cuda", 3)]            # per node -> per GPU (misc.py:48-53)
    a2 = _args(gpu=3, distributed=True, batch_size=64)
    assert send_model_cuda(a2, m, clip_batch=False) is m and a2.batch_size == 64                    # the ema model's call (train.py:400)
    assert send_model_cuda(_args(gpu=1), m) is m
    with pytest.raises(NotImplementedError, match="DataParallel"):
        send_model_cuda(_args(gpu=None), m)


def test_main_worker_sequence_on_the_base_class():
    R.CALLS.clear()
    args = _args()
    alg, n_params, _ = R.main_worker_tail(args, lambda a, nb, tb, lg: _HostOnly(a, nb, tb, lg), None, count_parameters,
                                          lambda a, m, clip_batch=True: m)
    # set_dataset (algorithmbase.py:140-166): the reference's call, its side effects on args
    assert R.CALLS[0] == ("get_dataset", "hostonly", "stand_in", 12, 10, "./data", True)
    assert args.ulb_dest_len == 40 and args.lb_dest_len == 12
    # set_data_loader (:185-228): batch sizes, sampler, iteration counts, worker counts, eval without sampler / drop_last
    assert R.CALLS[1] == ("get_data_loader", 12, 4, "RandomSampler", 5, 2, 3, True, False)
    assert R.CALLS[2] == ("get_data_loader", 40, 8, "RandomSampler", 5, 2, 6, True, False)
    assert R.CALLS[3] == ("get_data_loader", 6, 4, None, None, None, 3, False, False) and len(R.CALLS) == 4
    assert set(alg.loader_dict) == {"train_lb", "train_ulb", "eval"}
    assert n_params == sum(int(torch.Size(s).numel()) for _, s in alg.model.names_shapes)
    # train() (:346-375): num_train_iter steps, batches routed by train_step's signature (idx_lb is dropped), hooks in the reference's order
    steps = [e for e in alg.events if isinstance(e, tuple)]
    assert alg.it == 5 and len(steps) == 5 and steps[0] == ("train_step", (4, 3, 8, 8), (4,), (8,), (8, 3, 8, 8), (8, 3, 8, 8))
    names = [e for e in alg.events if isinstance(e, str)]
    assert names[:3] == ["before_run", "before_train_epoch", "before_train_step"] and names[-1] == "after_run"
    assert names.count("before_train_step") == names.count("after_train_step") == 5
    assert names.count("before_train_epoch") == names.count("after_train_epoch")


def test_ready_dicts_and_missing_pipeline():
    # no semilearn, no data functions, no dataset name: no loaders are invented -- the caller feeds train(batches=...) or assigns loader_dict
    a = _args(data_functions=None, dataset=None)
    alg = _HostOnly(a, None)
    assert alg.dataset_dict is None and alg.loader_dict is None
    # ready loaders handed in
    ld = {"train_lb": [], "train_ulb": []}
    assert _HostOnly(_args(data_functions=None, dataset=None, loader_dict=ld), None).loader_dict is ld
    # a ready dataset_dict without any get_data_loader is an error, not a silent None
    ds = R.get_dataset(_args(), "x", "stand_in", 4, 10)
    with pytest.raises(RuntimeError, match="get_data_loader"):
        _HostOnly(_args(data_functions=None, dataset_dict=ds), None)
