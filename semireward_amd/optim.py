"""Backbone optimizer state for the flat parameter block: AdamW with the reference's layer-decay
param groups and the cosine-with-warmup schedule, executed by ONE fused HIP launch per step.

Mirrors reference semantics (not code):
  get_optimizer / param_groups_layer_decay   semilearn/core/utils/build.py:193-224, semilearn/nets/utils.py:143-204
  get_cosine_schedule_with_warmup            semilearn/core/utils/build.py:227-251
  ParamUpdateHook.after_train_step           semilearn/core/hooks/param_update.py:33-40
"""
import math
import re

import torch

from . import ops

CHUNK = 4096


def build_chunk_table(sizes, chunk=CHUNK, offsets=None):
    """int32 [n_chunks, 4] = (offset, length, tensor id, 0); a chunk never straddles two tensors.  ``offsets``: start of every tensor in
    the flat block (default: densely packed)."""
    rows, off = [], 0
    for tid, n in enumerate(sizes):
        if offsets is not None:
            off = offsets[tid]
        o = 0
        while o < n:
            ln = min(chunk, n - o)
            rows.append((off + o, ln, tid, 0))
            o += ln
        off += n
    return torch.tensor(rows, dtype=torch.int32)


def vit_layer_id(name, depth):
    """VisionTransformer.group_matcher (vit.py:311-320) via group_with_matcher(reverse=True)."""
    if name in ("cls_token", "pos_embed") or name.startswith("patch_embed"):
        return 0
    m = re.match(r"^blocks\.(\d+)", name)
    if m:
        return int(m.group(1)) + 1
    if name.startswith("norm"):
        return depth            # MATCH_PREV_GROUP: shares the last block's id
    return depth + 1            # head -> layer_max


def layer_decay_hparams(names_shapes, depth, lr, weight_decay, layer_decay, no_weight_decay=("pos_embed", "cls_token"), layer_ids=None,
                        frozen=()):
    """(lr, weight_decay) per tensor.  ``layer_ids``: (name -> layer id, layer_max) of the model's group_matcher (default: the ViT's);
    ``frozen``: parameters whose gradient is None in the reference (torch optimizers skip them entirely: no update, no decay)."""
    out = []
    for name, shape in names_shapes:
        if layer_ids is not None:
            lid, lmax = layer_ids[0][name], layer_ids[1]
        else:
            lid, lmax = vit_layer_id(name, depth), depth + 1
        scale = layer_decay ** (lmax - lid) if layer_decay != 1.0 else 1.0
        wd = 0.0 if (len(shape) == 1 or name in no_weight_decay or name.endswith(".bias")) else weight_decay
        out.append((0.0, 0.0) if name in frozen else (scale * lr, wd))
    return out


def torch_group_order(model, layer_decay):
    """Parameter names per torch param group, in the order the reference's get_optimizer creates the groups (build.py:193-224): with layer
    decay one group per (layer id, no_decay | decay) in order of first appearance while walking named_parameters() (nets/utils.py:172-201,
    no_decay = 1-D or in no_weight_decay()); without it [no_decay, decay] (nets/utils.py:77-97, no_decay also for ``.bias``)."""
    nwd = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
    if layer_decay == 1.0:
        nd = [n for n, s in model.names_shapes if len(s) <= 1 or n.endswith(".bias") or n in nwd]
        return [nd, [n for n, _ in model.names_shapes if n not in set(nd)]]
    ids = model.layer_ids()[0] if hasattr(model, "layer_ids") else {n: vit_layer_id(n, model.cfg.depth) for n, _ in model.names_shapes}
    groups, key_of = [], {}
    for n, shp in model.names_shapes:
        k = (ids[n], len(shp) == 1 or n in nwd)
        if k not in key_of:
            key_of[k] = len(groups)
            groups.append([])
        groups[key_of[k]].append(n)
    return groups


def cosine_with_warmup(step, num_training_steps, num_warmup_steps=0, num_cycles=7.0 / 16.0):
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    t = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, math.cos(math.pi * num_cycles * t))


def _clip_coef(opt, grad_scale, clip_grad):
    """clip_grad_norm_(model.parameters(), clip_grad) of the reference's ParamUpdateHook (param_update.py:34-35) without rewriting the
    gradients: the global norm of the (already all-reduced, 1/world-scaled) flat gradient block is reduced on the device and the resulting
    coefficient min(1, clip / (norm + 1e-6)) is folded into the optimizer launch.  ``opt.last_grad_norm`` keeps [coef, norm] on the device."""
    if not clip_grad or clip_grad <= 0:
        return None
    if getattr(opt, "_clip_ws", None) is None:
        dev = opt.model.grad.device
        opt._clip_ws = torch.empty(ops.clip_grad_ws_floats(), dtype=torch.float32, device=dev)
        opt.last_grad_norm = torch.zeros(2, dtype=torch.float32, device=dev)
    ops.clip_grad_coef(opt.model.grad, opt.model.grad.numel(), float(grad_scale), float(clip_grad), opt._clip_ws, opt.last_grad_norm)
    return opt.last_grad_norm


class FusedAdamW:
    """Optimizer + LambdaLR scheduler of the reference collapsed into one object (step() == optimizer.step();
    scheduler.step(); model.zero_grad()).  ``state_dict`` is ENGINE-SPECIFIC (flat ``m`` / ``v`` blocks in named_parameters() order +
    step counters), not torch.optim's per-parameter ``state`` / ``param_groups`` layout; ``load_state_dict`` also accepts that torch layout
    (a checkpoint written by the reference, algorithmbase.py:459-475) and converts it."""

    def __init__(self, model, lr, weight_decay, layer_decay, num_train_iter, num_warmup_iter, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        dev = model.flat.device
        ns = model.names_shapes
        hp = layer_decay_hparams(ns, model.cfg.depth, lr, weight_decay, layer_decay, no_weight_decay=tuple(model.no_weight_decay()),
                                 layer_ids=model.layer_ids() if hasattr(model, "layer_ids") else None,
                                 frozen=tuple(getattr(model, "frozen_params", ())))
        self.lr_t = torch.tensor([h[0] for h in hp], dtype=torch.float32, device=dev)
        self.wd_t = torch.tensor([h[1] for h in hp], dtype=torch.float32, device=dev)
        self.table = build_chunk_table([int(torch.Size(s).numel()) for _, s in ns], offsets=[model.offsets[n][0] for n, _ in ns]).to(dev)
        self.m = torch.zeros_like(model.flat)
        self.v = torch.zeros_like(model.flat)
        self.betas, self.eps = betas, eps
        self.num_train_iter, self.num_warmup_iter = num_train_iter, num_warmup_iter
        self.step_count = 0          # optimizer steps taken (bias correction)
        self.sched_step = 0          # LambdaLR last_epoch
        self.base_lr = lr
        self.layer_decay = layer_decay

    def lr_factor(self):
        return cosine_with_warmup(self.sched_step, self.num_train_iter, self.num_warmup_iter)

    def get_last_lr(self):
        return [self.base_lr * self.lr_factor()]

    def step(self, ema=None, ema_m=0.0, grad_scale=1.0, clip_grad=0.0):
        self.step_count += 1
        coef = _clip_coef(self, grad_scale, clip_grad)
        sc = getattr(self, "step_scalars", None)            # core/stepgraph.py: this step's lr factor / bias corrections in device memory
        ops.adamw_flat(self.model.flat, self.model.grad, self.m, self.v, self.model.flat_bf16, ema, self.table,
                       self.table.shape[0], self.lr_t, self.wd_t, self.lr_factor(), self.step_count, self.betas[0],
                       self.betas[1], self.eps, ema_m=ema_m, grad_scale=grad_scale, zero_grad=True, clip_coef=coef,
                       dyn=sc.adamw_ptr if sc is not None else None)
        if getattr(self.model, "lazy_transposed", False):
            self.model._wT_stale = True          # refreshed off the critical path (model.ensure_transposed)
        else:
            self.model.refresh_transposed()
        self.sched_step += 1

    def state_dict(self):
        return dict(m=self.m.cpu(), v=self.v.cpu(), step=self.step_count, sched_step=self.sched_step)

    def load_state_dict(self, sd):
        if "state" in sd and "param_groups" in sd:                 # torch.optim.AdamW.state_dict() of a reference checkpoint
            return self._load_torch_state(sd)
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.step_count, self.sched_step = int(sd["step"]), int(sd["sched_step"])

    def _load_torch_state(self, sd):
        """torch layout: param_groups[i]['params'] = running indices, state[idx] = {step, exp_avg, exp_avg_sq}.  The reference builds its
        groups with param_groups_layer_decay (nets/utils.py:143-204): group order = first appearance of (layer id, decay / no_decay) while
        walking named_parameters(), so the running index -> parameter name map is rebuilt the same way."""
        order = torch_group_order(self.model, self.layer_decay)
        idx_of = {}
        for g, names in zip(sd["param_groups"], order):
            assert len(g["params"]) == len(names), "optimizer state does not match this model's parameter groups"
            for i, n in zip(g["params"], names):
                idx_of[n] = i
        step = 0
        for n, _ in self.model.names_shapes:
            st = sd["state"].get(idx_of[n])
            if st is None:
                continue                                           # frozen parameter: no state in torch either
            o, ln = self.model.offsets[n][0], int(st["exp_avg"].numel())
            self.m[o:o + ln].copy_(st["exp_avg"].reshape(-1)); self.v[o:o + ln].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, int(st["step"]))
        self.step_count = step
        return self


class FusedSGD:
    """torch.optim.SGD(momentum, nesterov=True) with the reference's two weight-decay groups (core/utils/build.py:193-224 with
    layer_decay == 1: nets/utils.py:77-97 -- no decay for 1-D parameters, ``.bias`` and the model's ``no_weight_decay()`` names) and the
    cosine LambdaLR, as ONE launch per step over the flat block (the classic_cv configs, e.g. WRN-28-2: lr 0.03, momentum 0.9)."""

    def __init__(self, model, lr, momentum, weight_decay, num_train_iter, num_warmup_iter, nesterov=True):
        assert nesterov, "the reference builds SGD with nesterov=True (build.py:193)"
        import numpy as np
        self.model = model
        dev = model.flat.device
        ns = model.names_shapes
        nwd = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
        ends = [model.offsets[n][0] for n, _ in ns][1:] + [model.numel]          # a parameter's chunk runs to the next one's start
        tab = np.zeros(len(ns), dtype=[("end", "<i8"), ("wd", "<f4"), ("pad", "<f4")])
        for i, (n, shp) in enumerate(ns):
            tab[i] = (ends[i], 0.0 if (len(shp) <= 1 or n.endswith(".bias") or n in nwd) else weight_decay, 0.0)
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        self.nchunks = len(ns)
        self.buf = torch.zeros_like(model.flat)
        self.momentum, self.base_lr = momentum, lr
        self.num_train_iter, self.num_warmup_iter = num_train_iter, num_warmup_iter
        self.step_count = 0
        self.sched_step = 0

    def lr_factor(self):
        return cosine_with_warmup(self.sched_step, self.num_train_iter, self.num_warmup_iter)

    def get_last_lr(self):
        return [self.base_lr * self.lr_factor()]

    def step(self, ema=None, ema_m=0.0, grad_scale=1.0, clip_grad=0.0):
        coef = _clip_coef(self, grad_scale, clip_grad)
        ops.sgd_flat(self.model.flat, self.model.grad, self.buf, ema, self.table, self.nchunks, self.model.numel,
                     self.base_lr * self.lr_factor(), self.momentum, grad_scale=grad_scale, ema_m=ema_m, first_step=self.step_count == 0,
                     zero_grad=True, clip_coef=coef)
        self.step_count += 1
        self.model.refresh_operands()
        self.sched_step += 1

    def state_dict(self):
        return dict(momentum_buffer=self.buf.cpu(), step=self.step_count, sched_step=self.sched_step)

    def load_state_dict(self, sd):
        if "state" in sd and "param_groups" in sd:                 # torch.optim.SGD.state_dict() of a reference checkpoint
            return self._load_torch_state(sd)
        self.buf.copy_(sd["momentum_buffer"])
        self.step_count, self.sched_step = int(sd["step"]), int(sd["sched_step"])

    def _load_torch_state(self, sd):
        """torch layout of the reference's SGD (build.py:193-224 without layer decay): param_groups = [no_decay, decay] (nets/utils.py:77-97),
        state[idx] = {'momentum_buffer'}.  torch keeps no step counter for SGD; what matters here is only whether a momentum buffer exists
        (the first step initialises it with the gradient): step_count = 1 if any buffer was loaded; the LambdaLR position comes from the
        checkpoint's scheduler entry (AlgorithmBase.load_model)."""
        order = torch_group_order(self.model, 1.0)
        assert len(sd["param_groups"]) == len(order), "optimizer state does not match this model's parameter groups"
        idx_of = {}
        for g, names in zip(sd["param_groups"], order):
            assert len(g["params"]) == len(names), "optimizer state does not match this model's parameter groups"
            for i, n in zip(g["params"], names):
                idx_of[n] = i
        self.buf.zero_()
        loaded = 0
        for n, _ in self.model.names_shapes:
            st = sd["state"].get(idx_of[n])
            mb = None if st is None else st.get("momentum_buffer")
            if mb is None:
                continue
            o = self.model.offsets[n][0]
            self.buf[o:o + int(mb.numel())].copy_(mb.reshape(-1))
            loaded += 1
        assert loaded in (0, len(self.model.names_shapes)), "momentum buffers for only part of the parameters"
        self.step_count = 1 if loaded else 0
        return self
