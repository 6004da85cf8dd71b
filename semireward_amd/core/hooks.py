"""Hook machinery (mmcv-style priority list) with the reference's call contract.

  Hook / priorities        semilearn/core/hooks/hook.py, priority.py
  ParamUpdateHook          semilearn/core/hooks/param_update.py:21-45
  EMAHook                  semilearn/core/hooks/ema.py:9-24
The backbone backward is fused into train_step (SURVEY.md 8(b) "Return"), so ParamUpdateHook here only
all-reduces the flat gradient block (data parallel) and runs the fused AdamW / scheduler / zero_grad launch.
"""
import torch

PRIORITIES = {"HIGHEST": 0, "VERY_HIGH": 10, "HIGH": 30, "ABOVE_NORMAL": 40, "NORMAL": 50, "BELOW_NORMAL": 60,
              "LOW": 70, "VERY_LOW": 90, "LOWEST": 100}


def get_priority(p):
    if isinstance(p, int):
        return p
    return PRIORITIES[p.upper()]


class Hook:
    stages = ("before_run", "before_train_epoch", "before_train_step", "after_train_step", "after_train_epoch", "after_run")

    def before_run(self, algorithm): pass
    def before_train_epoch(self, algorithm): pass
    def before_train_step(self, algorithm): pass
    def after_train_step(self, algorithm): pass
    def after_train_epoch(self, algorithm): pass
    def after_run(self, algorithm): pass

    def every_n_iters(self, algorithm, n):
        return (algorithm.it + 1) % n == 0 if n > 0 else False


class DeferredElapsed:
    """Seconds between two recorded HIP events, evaluated (one event synchronisation) only when somebody reads it.  The reference calls
    torch.cuda.synchronize() twice per step to fill train/run_time (param_update.py:15-18, :42-45); the log keys stay, the syncs go."""
    __slots__ = ("_a", "_b", "_v")

    def __init__(self, a, b):
        self._a, self._b, self._v = a, b, None

    def __float__(self):
        if self._v is None:
            self._b.synchronize()
            self._v = self._a.elapsed_time(self._b) / 1000.0
            self._a = self._b = None
        return self._v

    def item(self):
        return float(self)

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))


class ParamUpdateHook(Hook):
    """after_train_step: [DP all-reduce of grads] -> optimizer.step() -> scheduler.step() -> zero_grad  (one launch).
    ``train/run_time`` (param_update.py:15-18, :42-45): GPU time from before_train_step to the end of the parameter update, when a
    TimerHook has armed it (``algorithm.start_run``)."""

    def before_train_step(self, algorithm):
        if getattr(algorithm, "start_run", None) is not None:
            algorithm.start_run = torch.cuda.Event(enable_timing=True)
            algorithm.start_run.record()

    def after_train_step(self, algorithm):
        scale = 1.0
        if (algorithm.distributed and algorithm.world_size > 1) or algorithm.dp.force:
            # sum over the ranks and 1/world belong together: a configuration that says "distributed" without an initialised process group
            # would otherwise divide local gradients by world_size (a silent learning-rate change)
            if not algorithm.dp.active:
                raise RuntimeError("args.distributed with world_size %d%s but torch.distributed is not initialised: call init_process_group "
                                   "before the first step (semilearn/train.py:374-379)" % (algorithm.world_size, " (force_dp)" if algorithm.dp.force else ""))
            algorithm.dp.all_reduce_grads(algorithm.model)
            scale = 1.0 / algorithm.world_size
        ema, ema_m = None, 0.0
        if algorithm.ema_model is not None and algorithm.ema_model is not algorithm.model:
            ema, ema_m = algorithm.ema_model.flat, algorithm.ema_m
        from .. import ops
        with ops.stream_scope():
            algorithm.optimizer.step(ema=ema, ema_m=ema_m, grad_scale=scale, clip_grad=algorithm.clip_grad)
        if getattr(algorithm, "start_run", None) is not None and isinstance(getattr(algorithm, "log_dict", None), dict):
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            algorithm.log_dict["train/run_time"] = DeferredElapsed(algorithm.start_run, end)


class TimerHook(Hook):
    """semilearn/core/hooks/timer.py:9-27: ``lr`` and ``train/prefetch_time`` (GPU time between the end of one step and the start of the
    next = what the step waits for the input pipeline) in log_dict, and it arms ParamUpdateHook's ``train/run_time``.  Events are recorded
    without host synchronisation; the values are DeferredElapsed."""

    def before_run(self, algorithm):
        algorithm.start_run = torch.cuda.Event(enable_timing=True)      # armed: ParamUpdateHook records into a fresh event every step
        algorithm.start_batch = torch.cuda.Event(enable_timing=True)
        algorithm.start_batch.record()
        algorithm.end_batch = None

    def before_train_step(self, algorithm):
        if getattr(algorithm, "start_batch", None) is None:
            self.before_run(algorithm)
        algorithm.end_batch = torch.cuda.Event(enable_timing=True)
        algorithm.end_batch.record()

    def after_train_step(self, algorithm):
        if getattr(algorithm, "end_batch", None) is None or not isinstance(getattr(algorithm, "log_dict", None), dict):
            return
        algorithm.log_dict["lr"] = algorithm.optimizer.get_last_lr()[-1]
        algorithm.log_dict["train/prefetch_time"] = DeferredElapsed(algorithm.start_batch, algorithm.end_batch)
        algorithm.start_batch = torch.cuda.Event(enable_timing=True)
        algorithm.start_batch.record()


class EMAHook(Hook):
    """The reference copies every parameter three times per step even for ema_m == 0 (ema.py:20-24).  Here the EMA
    shadow is either an alias of the model (ema_m == 0: shadow == params exactly) or updated inside the AdamW launch."""

    def before_run(self, algorithm):
        pass
