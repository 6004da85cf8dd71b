"""One training step as a HIP graph: captured once per step variant, replayed every step.

Why: a step of the SemiReward hot path is ~270 small-to-medium launches on two or three HIP streams; enqueueing them from Python costs 2.5 ms of
host time per step (tools/host_overhead.py) against ~5 ms of GPU time at the reference batch and MORE than the GPU time in the pre-start_timing
regime -- the host is on the critical path there, and it is the first thing to de-phase data-parallel ranks.  A captured step replays with one call.

What has to hold for a capture to be valid for every later step:
  * every per-step host scalar reaches the kernels through device memory: the scheduler's lr factor and Adam's bias corrections (AdamW of the
    backbone, Adam of the rewarder) and the DropPath seed are written into ``StepScalars`` (one 64-byte H2D copy per step, outside the graph) and
    read by the "_dyn" entry points of the C ABI (include/srhip.h);
  * control flow that depends on ``it`` selects a VARIANT, not a branch inside a graph: ``algorithm.step_variant()`` (K = sr_decay(), which
    SemiReward update the step performs) keys the graph cache, together with the batch shapes;
  * state the step mutates lives in persistent device buffers updated in place (FlexMatch table / histogram, rewarder parameters and moments,
    ``max_reward``); Python-side counters the captured code advances (optimizer step, scheduler step, DropPath draw counter, rewarder-Adam step)
    are advanced by the same amounts at every replay;
  * inputs are copied into the static batch tensors of the capture (the graph owns clones of the first batch: one device-to-device copy per
    batch tensor and replay);
  * the out / log dictionaries a replay returns read the graph's own buffers: valid until the next step() (all variants share one memory
    pool, so a later replay of ANOTHER variant may reuse them) -- read log scalars before stepping again, as LoggingHook does.
The reference sequence this replaces per step: SRFlexMatch.train_step (srflexmatch.py:107-217) + ParamUpdateHook.after_train_step
(param_update.py:21-45).  Results are bit-identical to the eager step (tests/test_gpu_stepgraph.py).
"""
import os

import torch

from .. import ops


def _log_sources(log):
    """What the DeferredScalars of a captured step's log_dict read (device tensors of the graph's memory pool / callables on them): a DeferredScalar
    caches its value at the first read, so every replay gets fresh ones on the same sources."""
    from .algorithmbase import DeferredScalar
    return {k: (("d", v._t) if isinstance(v, DeferredScalar) else ("v", v)) for k, v in log.items()}


def _fresh_log(src):
    from .algorithmbase import DeferredScalar
    return {k: (DeferredScalar(v) if kind == "d" else v) for k, (kind, v) in src.items()}


class StepScalars:
    """The per-step scalars of the "_dyn" launches: a ring of pinned host slots + one device slot (64 bytes):
         bytes  0 ..  7   uint64  DropPath seed base of the step            (srhip_droppath_fill_cols_dyn)
         bytes 16 .. 27   fp32    lr_factor, 1 - b1^t, sqrt(1 - b2^t)       (srhip_adamw_flat_dyn, backbone)
         bytes 32 .. 39   fp32    1 - b1^t, sqrt(1 - b2^t)                  (srhip_adam_flat_dyn, rewarder)"""
    SLOTS = 1024

    def __init__(self, device):
        self.dev = torch.zeros(16, dtype=torch.float32, device=device)
        self.host = torch.zeros(self.SLOTS, 16, dtype=torch.float32).pin_memory()
        self.n = 0

    @property
    def seed_ptr(self):
        return self.dev.data_ptr()

    @property
    def adamw_ptr(self):
        return self.dev.data_ptr() + 16

    @property
    def adam_ptr(self):
        return self.dev.data_ptr() + 32

    def push(self, algorithm):
        """Values of the step that is about to run, from the host-side counters; one async copy on the current stream."""
        slot = self.host[self.n % self.SLOTS]
        self.n += 1
        m, opt = algorithm.model, algorithm.optimizer
        slot.view(torch.int64)[0] = (int(getattr(m, "seed", 0)) << 32) + int(getattr(m, "_rng_calls", 0))
        m._step_draws = 0                                     # make_droppath numbers its draws inside the step (nets/vit.py)
        if hasattr(opt, "betas"):
            bc = ops.adam_bias_corrections(opt.betas[0], opt.betas[1], opt.step_count + 1)
            slot[4], slot[5], slot[6] = opt.lr_factor(), bc[0], bc[1]
        ro = getattr(algorithm, "rewarder_optimizer", None)
        if ro is not None:
            slot[8], slot[9] = ops.adam_bias_corrections(0.9, 0.999, ro.steps + 1)
        self.dev.copy_(slot, non_blocking=True)


class StepGraph:
    """``step(**batch)`` == ``train_step(**batch)`` + ``ParamUpdateHook.after_train_step`` of the algorithm, eagerly for the first ``warm`` steps of
    a variant (and while the step schedule is still being tuned), from then on as a captured HIP graph."""

    MAX_GRAPHS = 8           # captured variants kept (least recently used one dropped beyond that)
    COUNTERS = (("optimizer", "step_count"), ("optimizer", "sched_step"), ("model", "_rng_calls"), ("rewarder_optimizer", "steps"))

    def __init__(self, algorithm, warm=2):
        if not getattr(algorithm, "graph_safe", False):
            raise ValueError("%s keeps per-step state in Python objects: not capturable" % type(algorithm).__name__)
        self.alg, self.warm = algorithm, warm
        self.scal = StepScalars(algorithm.device)
        algorithm.step_scalars = algorithm.model.step_scalars = self.scal
        if getattr(algorithm, "rewarder_optimizer", None) is not None:
            algorithm.rewarder_optimizer.step_scalars = self.scal
        algorithm.optimizer.step_scalars = self.scal
        self.graphs, self.seen = {}, {}                      # (dict order = least .. most recently used)
        self.pool, self.max_graphs = None, max(1, self.MAX_GRAPHS)
        self.replays = self.eager_steps = 0
        self.hook = algorithm.hooks_dict["ParamUpdateHook"]

    def _counters(self):
        out = []
        for obj, name in self.COUNTERS:
            o = getattr(self.alg, obj, None)
            out.append(getattr(o, name, 0) if o is not None else 0)
        return out

    def _advance(self, deltas):
        for (obj, name), d in zip(self.COUNTERS, deltas):
            o = getattr(self.alg, obj, None)
            if o is not None and d:
                setattr(o, name, getattr(o, name) + d)

    def _eager(self, batch, update=True):
        alg = self.alg
        alg.out_dict, alg.log_dict = alg.train_step(**batch)
        if update:
            self.hook.after_train_step(alg)
        return alg.out_dict, alg.log_dict

    def _split(self, variant):
        """Data parallel: may this variant be captured WITHOUT its parameter update?  The gradient exchange then runs between the replayed
        train_step and the (one-launch) optimizer step, outside the graph -- the collectives never enter a capture, the host still enqueues
        ~3 calls per step instead of ~270.  Not while the exchange is being selected, not with the exchange under the backward (its
        collectives are issued from inside the backward), not for the variants whose train_step contains a collective of its own (the
        rewarder update all-reduces its gradient; the global reward threshold)."""
        dp = self.alg.dp
        if not dp.active:
            return False
        ok = dp.settled and getattr(self.alg.model, "grad_ready_cb", None) is None and not dp.global_reward_threshold and variant[1] != 3
        return True if ok else None           # None: data parallel, but this step stays eager

    def step(self, **batch):
        alg = self.alg
        sig = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()) if torch.is_tensor(v))
        key = (alg.step_variant(), sig)
        self.scal.push(alg)
        split = self._split(key[0])
        ent = self.graphs.get(key) if split is not None else None
        if ent is not None and ent[5] == bool(split):
            self.graphs[key] = self.graphs.pop(key)          # most recently used last
            g, static, deltas, out, log, _ = ent
            for k, v in batch.items():
                if static[k] is not v:
                    static[k].copy_(v, non_blocking=True)
            g.replay()
            self._advance(deltas)
            alg.out_dict, alg.log_dict = out, _fresh_log(log)
            if split:                                        # data parallel: [gradient exchange] + optimizer launch, eagerly behind the replay
                self.hook.after_train_step(alg)
            self.replays += 1
            return out, alg.log_dict
        n = self.seen.get(key, 0)
        self.seen[key] = n + 1
        if n < self.warm or getattr(alg, "_tuners", None) or getattr(alg, "_untuned", None) or alg.trace is not None or split is None:
            self.eager_steps += 1
            return self._eager(batch)
        # ---- capture this step (nothing executes during capture), then replay it once: that IS this step
        # The graph owns its inputs (clones: a replay must not overwrite the tensors the caller handed in for an earlier step); every variant
        # captures into ONE shared memory pool (K = sr_decay() takes dozens of values over a run, times the rewarder-update variants: private
        # pools would pin a full step's activations each), and the cache is bounded -- the least recently used variant is dropped.
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        while len(self.graphs) >= self.max_graphs:
            self.graphs.pop(next(iter(self.graphs)))
        before = self._counters()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(g, pool=self.pool):
            out, log = self._eager(static, update=not split)
        deltas = [a - b for a, b in zip(self._counters(), before)]
        src = _log_sources(log)
        self.graphs[key] = (g, static, deltas, out, src, bool(split))
        g.replay()
        self.replays += 1
        alg.out_dict, alg.log_dict = out, _fresh_log(src)
        if split:
            self.hook.after_train_step(alg)
        return out, alg.log_dict
