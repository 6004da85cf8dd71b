"""AlgorithmBase: the reference's plugin surface (semilearn/core/algorithmbase.py:49-606) on the HIP engine.

Kept identical: constructor signature ``(args, net_builder, tb_log=None, logger=None)``, attribute names
(model, ema_model, optimizer, scheduler, loader_dict, it, epoch, hooks_dict, ce_loss, consistency_loss ...),
``process_batch`` (signature-driven kwargs filter, :282-306), ``train`` loop and hook dispatch (:346-375,
:579-593), ``sr_decay`` (:177-183), ``get_save_dict`` / ``load_model`` keys (:459-547).
Different by design: the backbone backward is fused into ``train_step`` (``out_dict['loss']`` is a
detached scalar and ``model.grad`` is already populated), the optimizer + scheduler are one fused object, the EMA
model is an alias when ``ema_m == 0``, and log scalars are fetched lazily (no per-step host sync).
Datasets / loaders are the reference's own (CPU input pipeline): ``set_dataset`` / ``set_data_loader`` call ITS ``get_dataset`` /
``get_data_loader`` when semilearn is importable (or take ready dicts from ``args``); ``train()`` then zips ``loader_dict['train_lb'/'train_ulb']``
as the reference does, or consumes any iterable of (data_lb, data_ulb) handed in.
"""
import os
from collections import OrderedDict
from inspect import signature

import torch

from .. import ops
from ..distributed import DataParallel
from ..optim import FusedAdamW, FusedSGD
from .criterions import CELoss, ConsistencyLoss
from .hooks import EMAHook, Hook, ParamUpdateHook, TimerHook, get_priority
from .utils import reference_data_functions


class DeferredScalar:
    """A device scalar that becomes a python float on first use (float(), format, comparison).  The reference calls
    .item() four times per step (srflexmatch.py:213-216): four host syncs.  LoggingHook only reads them every
    num_log_iter steps, so the sync is deferred to that moment.  ``t`` may also be a callable returning the device scalar: then even
    the reduction that produces it (util_ratio = mask.mean(), :216) is only launched when somebody reads the value."""
    __slots__ = ("_t", "_v")

    def __init__(self, t):
        self._t, self._v = t, None

    def __float__(self):
        if self._v is None:
            t = self._t() if callable(self._t) else self._t
            self._v = float(t.item())
            self._t = None
        return self._v

    def item(self):
        return float(self)

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))

    def __add__(self, o): return float(self) + float(o)
    __radd__ = __add__
    def __mul__(self, o): return float(self) * float(o)
    __rmul__ = __mul__
    def __lt__(self, o): return float(self) < float(o)
    def __gt__(self, o): return float(self) > float(o)
    def __eq__(self, o): return float(self) == float(o)
    def __hash__(self): return id(self)


class AlgorithmBase:
    def __init__(self, args, net_builder, tb_log=None, logger=None, **kwargs):
        self.args = args
        g = lambda k, d=None: getattr(args, k, d)   # noqa: E731  (yaml overlay: any key may or may not exist)
        self.num_classes = args.num_classes
        self.ema_m = g("ema_m", 0.0)
        self.epochs = g("epoch", 1)
        self.num_train_iter = args.num_train_iter
        self.num_eval_iter = g("num_eval_iter", 0)
        self.num_log_iter = g("num_log_iter", 0)
        self.num_iter_per_epoch = int(self.num_train_iter // max(1, self.epochs))
        self.lambda_u = g("ulb_loss_ratio", 1.0)
        self.use_cat = g("use_cat", True)
        self.use_amp = g("amp", False)
        if self.use_amp:
            # the reference's amp is fp16 autocast + GradScaler (algorithmbase.py:103-104, param_update.py:27-32); this engine has one
            # numeric mode (bf16 operands, fp32 accumulate / residual / optimizer) and no loss scaler: refuse rather than ignore the key
            raise NotImplementedError("amp: True is not supported by the HIP engine (bf16 operands / fp32 accumulation is always on; "
                                      "set amp: False as every config/SemiReward yaml does)")
        self.clip_grad = float(g("clip_grad", 0) or 0)      # > 0: clip_grad_norm_ folded into the optimizer launch (optim._clip_coef)
        self.save_name, self.save_dir = g("save_name", "run"), g("save_dir", "./saved_models")
        self.resume = g("resume", False)
        self.algorithm = g("algorithm", None)
        self.tb_log = tb_log
        self.print_fn = print if logger is None else logger.info
        self.gpu = g("gpu", None)
        self.rank = g("rank", 0)
        self.distributed = g("distributed", False)
        self.world_size = g("world_size", 1)
        # (without a GPU only the host-side wiring of this class can be exercised -- tests/test_cpu_driver_contract.py; every op raises)
        self.device = torch.device("cuda", self.gpu if isinstance(self.gpu, int) else torch.cuda.current_device()) \
            if torch.cuda.is_available() else torch.device("cpu")
        self.dp = DataParallel(self.world_size, self.rank, global_reward_threshold=g("global_reward_threshold", False), force=g("force_dp", None))
        self.it = 0
        self.epoch = 0
        self.start_epoch = 0
        self.best_eval_metric, self.best_it = 0.0, 0
        self.net_builder = net_builder
        self.ema = None
        self.dataset_dict = self.set_dataset()
        self.loader_dict = self.set_data_loader()
        self.model = self.set_model()
        if getattr(self.model, "couples_batch_rows", False):
            self.model.dp = self.dp          # BatchNorm backbone under data parallel = SyncBatchNorm, as the reference's send_model_cuda (misc.py:55)
        self.ema_model = self.set_ema_model()
        self.optimizer, self.scheduler = self.set_optimizer()
        self.ce_loss = CELoss()
        self.consistency_loss = ConsistencyLoss()
        self.task_type = "cls"
        self._hooks = []
        self.hooks_dict = OrderedDict()
        self.set_hooks()

    def init(self, **kwargs):
        raise NotImplementedError

    # ---- construction ---------------------------------------------------------------------------
    def set_dataset(self):
        """algorithmbase.py:140-166.  The datasets are the reference's own (CPU input pipeline): built by ITS ``get_dataset``
        when ``semilearn`` is importable (or by the functions handed in as ``args.data_functions = (get_dataset, get_data_loader)``), with the
        reference's rank-0-first barriers and its ``ulb_dest_len`` / ``lb_dest_len`` side effects on ``args``.  ``args.dataset_dict``: a ready
        dict is taken as is.  Otherwise None -- ``train(batches=...)`` or a ``loader_dict`` assigned by the caller feeds the loop."""
        a = self.args
        ready = getattr(a, "dataset_dict", None)
        get_dataset = (getattr(a, "data_functions", None) or reference_data_functions())[0]
        if ready is None and (get_dataset is None or getattr(a, "dataset", None) is None):
            return None
        if ready is None:
            if self.rank != 0 and self.distributed:
                torch.distributed.barrier()
            ready = get_dataset(a, self.algorithm, a.dataset, a.num_labels, a.num_classes, getattr(a, "data_dir", "./data"),
                                getattr(a, "include_lb_to_ulb", True))
        if ready is None:
            return None
        a.ulb_dest_len = len(ready["train_ulb"]) if ready.get("train_ulb") is not None else 0
        a.lb_dest_len = len(ready["train_lb"])
        self.print_fn("unlabeled data number: {}, labeled data number {}".format(a.ulb_dest_len, a.lb_dest_len))
        if self.rank == 0 and self.distributed and getattr(a, "dataset_dict", None) is None:
            torch.distributed.barrier()
        return ready

    def set_data_loader(self):
        """algorithmbase.py:185-228: train_lb / train_ulb / eval (/ test) loaders from ``dataset_dict`` through the reference's
        ``get_data_loader`` with the reference's arguments; ``args.loader_dict`` (ready loaders) when there is no dataset_dict."""
        a = self.args
        if self.dataset_dict is None:
            return getattr(a, "loader_dict", None)
        get_data_loader = (getattr(a, "data_functions", None) or reference_data_functions())[1]
        if get_data_loader is None:
            raise RuntimeError("a dataset_dict was given but no get_data_loader: install semilearn or pass args.data_functions")
        self.print_fn("Create train and test data loaders")
        nw = getattr(a, "num_workers", 1)
        train = dict(data_sampler=getattr(a, "train_sampler", "RandomSampler"), num_iters=self.num_train_iter, num_epochs=self.epochs,
                     distributed=self.distributed)
        ld = {"train_lb": get_data_loader(a, self.dataset_dict["train_lb"], a.batch_size, num_workers=nw, **train),
              "train_ulb": get_data_loader(a, self.dataset_dict["train_ulb"], a.batch_size * a.uratio, num_workers=2 * nw, **train)}
        for k in ("eval", "test"):                       # evaluation: no sampler, keep the last partial batch
            if self.dataset_dict.get(k) is not None:
                ld[k] = get_data_loader(a, self.dataset_dict[k], a.eval_batch_size, data_sampler=None, num_workers=nw, drop_last=False)
        self.print_fn(f"[!] data loader keys: {ld.keys()}")
        return ld

    def set_model(self):
        model = self.net_builder(num_classes=self.num_classes, device=self.device)
        model.refresh_operands()
        return model

    def set_ema_model(self):
        if not self.ema_m:
            return self.model        # ema_m == 0 -> shadow == params after every step (misc.py:152-155): alias, 0 bytes moved
        ema = self.net_builder(num_classes=self.num_classes, device=self.device)
        ema.load_state_dict(self.model.state_dict())
        if hasattr(ema, "buffers"):          # BatchNorm statistics are not averaged: EMAHook copies the model's (core/hooks/ema.py:20-24)
            ema.buffers = self.model.buffers
        return ema

    def set_optimizer(self):
        a = self.args
        if getattr(a, "optim", "AdamW") == "SGD":         # classic_cv configs (WRN): SGD + Nesterov momentum, no layer decay
            assert getattr(a, "layer_decay", 1.0) == 1.0
            opt = FusedSGD(self.model, a.lr, getattr(a, "momentum", 0.9), getattr(a, "weight_decay", 0.0), self.num_train_iter,
                           getattr(a, "num_warmup_iter", 0))
            return opt, opt
        opt = FusedAdamW(self.model, a.lr, getattr(a, "weight_decay", 0.0), getattr(a, "layer_decay", 1.0),
                         self.num_train_iter, getattr(a, "num_warmup_iter", 0))
        return opt, opt      # optimizer and scheduler are one fused object

    def set_hooks(self):
        self.register_hook(ParamUpdateHook(), None, "HIGHEST")
        self.register_hook(EMAHook(), None, "HIGH")
        self.register_hook(TimerHook(), None, "LOWEST")           # algorithmbase.py:564: train/prefetch_time, train/run_time, lr

    # ---- batch / dict helpers (algorithmbase.py:282-333) -------------------------------------------
    def process_batch(self, input_args=None, **kwargs):
        if input_args is None:
            input_args = list(signature(self.train_step).parameters.keys())
        out = {}
        for arg, var in kwargs.items():
            if arg not in input_args or var is None:
                continue
            if isinstance(var, dict):
                var = {k: v.to(self.device, non_blocking=True) for k, v in var.items()}
            else:
                var = var.to(self.device, non_blocking=True)
            out[arg] = var
        return out

    def process_out_dict(self, out_dict=None, **kwargs):
        out_dict = {} if out_dict is None else out_dict
        out_dict.update(kwargs)
        return out_dict

    def process_log_dict(self, log_dict=None, prefix="train", **kwargs):
        log_dict = {} if log_dict is None else log_dict
        for k, v in kwargs.items():
            log_dict[prefix + "/" + k] = v
        return log_dict

    def compute_prob(self, logits):
        return torch.softmax(logits, dim=-1)     # API parity only; the hot path fuses softmax into srhip_row_max

    def sr_decay(self, max_sampling_time=8):
        """algorithmbase.py:177-183."""
        return int(max(max_sampling_time, 1 + (self.num_train_iter / self.it)))

    def train_step(self, *a, **k):
        raise NotImplementedError

    def _check_device_flags(self):
        """The error conditions the kernels record on the device instead of raising mid-stream, read where the host may wait: an out-of-range
        label / idx_ulb (IndexError at the offending call in the reference), unequal per-rank batches under SyncBatchNorm (nets/wrn.py)."""
        ops.check_label_errors()
        chk = getattr(self.model, "check_equal_rows", None)
        if chk is not None:
            chk()

    # ---- outer loop (algorithmbase.py:346-375) -------------------------------------------------------
    def train(self, batches=None):
        self.model.train()
        self.call_hook("before_run")
        for epoch in range(self.start_epoch, self.epochs):
            self.epoch = epoch
            if self.it >= self.num_train_iter:
                break
            self.call_hook("before_train_epoch")
            src = batches if batches is not None else zip(self.loader_dict["train_lb"], self.loader_dict["train_ulb"])
            for data_lb, data_ulb in src:
                if self.it >= self.num_train_iter:
                    break
                self.call_hook("before_train_step")
                self.out_dict, self.log_dict = self.train_step(**self.process_batch(**data_lb, **data_ulb))
                self.call_hook("after_train_step")
                self.it += 1
                # out-of-range label / idx_ulb (IndexError at the offending call in the reference): the device flag is read at the logging
                # cadence, where the log scalars synchronise anyway -- bad data stops training within num_log_iter steps, not at the epoch end
                if self.num_log_iter and self.it % self.num_log_iter == 0:
                    self._check_device_flags()
            self.call_hook("after_train_epoch")
            self._check_device_flags()
        self.call_hook("after_run")

    # ---- evaluation (algorithmbase.py:377-457) ----------------------------------------------------------------
    @staticmethod
    def classification_metrics(y_true, y_pred):
        """accuracy / balanced accuracy / macro precision, recall, F1 with the conventions of the sklearn calls the reference makes
        (algorithmbase.py:419-423): macro averages over the labels present in y_true or y_pred, 0 for an undefined ratio; balanced
        accuracy over the classes present in y_true."""
        import numpy as np
        y_true, y_pred = np.asarray(y_true, np.int64), np.asarray(y_pred, np.int64)
        labels = np.union1d(y_true, y_pred)
        idx = {int(c): i for i, c in enumerate(labels)}
        cm = np.zeros((labels.size, labels.size), dtype=np.int64)
        np.add.at(cm, (np.vectorize(idx.get)(y_true), np.vectorize(idx.get)(y_pred)), 1)
        tp, sup, prd = np.diag(cm).astype(np.float64), cm.sum(1).astype(np.float64), cm.sum(0).astype(np.float64)
        rec = np.divide(tp, sup, out=np.zeros_like(tp), where=sup > 0)
        prec = np.divide(tp, prd, out=np.zeros_like(tp), where=prd > 0)
        f1 = np.divide(2 * prec * rec, prec + rec, out=np.zeros_like(tp), where=(prec + rec) > 0)
        return {"top-1-acc": float(tp.sum() / max(y_true.size, 1)), "balanced_acc": float(rec[sup > 0].mean()),
                "precision": float(prec.mean()), "recall": float(rec.mean()), "F1": float(f1.mean())}

    def evaluate(self, eval_dest="eval", out_key="logits", return_logits=False, loader=None):
        """Inference over ``loader_dict[eval_dest]`` (or ``loader``: an iterable of {'x_lb', 'y_lb'}) with the EMA weights
        (``ema.apply_shadow`` of the reference == evaluating ``ema_model`` here), same result keys.  The forward runs on the
        inference kernels of the training path; predictions stay on the device until ONE copy at the end (the reference
        synchronises three times per batch)."""
        import numpy as np
        assert out_key == "logits"
        self._invalidate_step_timing()
        net = self.ema_model
        if net is not self.model:
            net.refresh_operands()                    # bf16 operand copy of the EMA block
        was_training = getattr(net, "training", True)
        net.training = False                          # BatchNorm backbones: running statistics (model.eval(), :381)
        loader = loader if loader is not None else self.loader_dict[eval_dest]
        C = self.num_classes
        logits_all, y_all = [], []
        loss_sum = torch.zeros(1, dtype=torch.float32, device=self.device)
        total = 0
        with ops.stream_scope():
            for data in loader:
                x = data["x_lb"]
                if isinstance(x, dict):                                      # usb_nlp: {'input_ids', 'attention_mask'} (nlp_collactor.py:73)
                    from ..nets.bert import TokenBatch
                    x = TokenBatch.from_dict(x, self.device)
                else:
                    x = x.to(self.device, non_blocking=True).contiguous()
                y = data["y_lb"].to(self.device, non_blocking=True).contiguous()
                B = int(y.shape[0])
                lg, _, _ = net.forward_features(x, None, None, save=False)
                w = (y >= 0).to(torch.float32)                               # F.cross_entropy(ignore_index=-1): mean over the kept rows
                loss = torch.empty(1, dtype=torch.float32, device=self.device)
                ops.masked_ce(lg, y.clamp_min(0), w, None, 1.0, loss, torch.empty(B, C, dtype=torch.float32, device=self.device), B, C)
                loss_sum += loss * (B * B / w.sum().clamp_min(1.0))          # masked_ce averages over B rows -> mean over kept, times B
                logits_all.append(lg)
                y_all.append(y)
                total += B
        net.training = was_training
        logits = torch.cat(logits_all)
        y_true = torch.cat(y_all).cpu().numpy()
        y_pred = logits.argmax(dim=-1).cpu().numpy()
        m = self.classification_metrics(y_true, y_pred)
        out = {eval_dest + "/loss": float(loss_sum) / max(total, 1)}
        out.update({eval_dest + "/" + k: v for k, v in m.items()})
        if return_logits:
            out[eval_dest + "/logits"] = logits.cpu().numpy()
        return out

    # ---- checkpoints (algorithmbase.py:459-547) ----------------------------------------------------------
    def get_save_dict(self):
        d = {"model": self.model.state_dict(), "ema_model": self.ema_model.state_dict(),
             "optimizer": self.optimizer.state_dict(), "scheduler": {"last_epoch": self.optimizer.sched_step},
             "loss_scaler": {}, "it": self.it + 1, "epoch": self.epoch + 1, "best_it": self.best_it,
             "best_eval_acc": self.best_eval_metric}
        # extension (the reference's checkpoint does not hold torch's global RNG state: its resumed run draws other DropPath / dropout masks than
        # the run that never stopped): the engine's draws are counter based (seed, draw counter), two integers make the continuation identical
        if hasattr(self.model, "_rng_calls"):
            d["engine_rng"] = {"seed": int(getattr(self.model, "seed", 0)), "draws": int(self.model._rng_calls)}
        return d

    def _invalidate_step_timing(self):
        """Evaluation / checkpointing between two training steps: a schedule tuner that times steps by their start events (srflexmatch._DeferTuner)
        must not count this gap as a step."""
        for tuner, _ in getattr(self, "_tuners", {}).values():
            tuner.invalidate()

    def save_model(self, save_name, save_path):
        self._invalidate_step_timing()
        self._check_device_flags()
        os.makedirs(save_path, exist_ok=True)
        torch.save(self.get_save_dict(), os.path.join(save_path, save_name))

    def load_model(self, load_path):
        ck = torch.load(load_path, map_location="cpu")
        self.model.load_state_dict(ck["model"])
        if self.ema_model is not self.model:
            self.ema_model.load_state_dict(ck["ema_model"])
        self.it, self.start_epoch = ck["it"], ck["epoch"]
        self.epoch, self.best_it = self.start_epoch, ck["best_it"]
        self.best_eval_metric = ck.get("best_eval_acc", 0.0)
        self.optimizer.load_state_dict(ck["optimizer"])             # engine layout, or torch.optim layout of a reference checkpoint
        sch = ck.get("scheduler") or {}
        if "last_epoch" in sch:                                     # LambdaLR.state_dict() (algorithmbase.py:468)
            self.optimizer.sched_step = int(sch["last_epoch"])
        if "engine_rng" in ck and hasattr(self.model, "_rng_calls"):
            self.model.seed, self.model._rng_calls = int(ck["engine_rng"]["seed"]), int(ck["engine_rng"]["draws"])
        return ck

    # ---- hooks (algorithmbase.py:548-599) ---------------------------------------------------------------------
    def register_hook(self, hook, name=None, priority="NORMAL"):
        assert isinstance(hook, Hook)
        hook.priority = get_priority(priority)
        hook.name = name if name is not None else type(hook).__name__
        inserted = False
        for i in range(len(self._hooks) - 1, -1, -1):
            if hook.priority >= self._hooks[i].priority:
                self._hooks.insert(i + 1, hook)
                inserted = True
                break
        if not inserted:
            self._hooks.insert(0, hook)
        self.hooks_dict = OrderedDict((h.name, h) for h in self._hooks)

    def call_hook(self, fn_name, hook_name=None, *args, **kwargs):
        if hook_name is not None:
            return getattr(self.hooks_dict[hook_name], fn_name)(self, *args, **kwargs)
        for hook in self.hooks_dict.values():
            if hasattr(hook, fn_name):
                getattr(hook, fn_name)(self, *args, **kwargs)

    def registered_hook(self, hook_name):
        return hook_name in self.hooks_dict

    @staticmethod
    def get_argument():
        return {}
