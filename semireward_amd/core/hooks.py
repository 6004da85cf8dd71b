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


class ParamUpdateHook(Hook):
    """after_train_step: [DP all-reduce of grads] -> optimizer.step() -> scheduler.step() -> zero_grad  (one launch)."""

    def after_train_step(self, algorithm):
        scale = 1.0
        if algorithm.distributed and algorithm.world_size > 1:
            algorithm.dp.all_reduce_grads(algorithm.model)
            scale = 1.0 / algorithm.world_size
        ema, ema_m = None, 0.0
        if algorithm.ema_model is not None and algorithm.ema_model is not algorithm.model:
            ema, ema_m = algorithm.ema_model.flat, algorithm.ema_m
        from .. import ops
        with ops.stream_scope():
            algorithm.optimizer.step(ema=ema, ema_m=ema_m, grad_scale=scale)


class EMAHook(Hook):
    """The reference copies every parameter three times per step even for ema_m == 0 (ema.py:20-24).  Here the EMA
    shadow is either an alias of the model (ema_m == 0: shadow == params exactly) or updated inside the AdamW launch."""

    def before_run(self, algorithm):
        pass
