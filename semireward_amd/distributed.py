"""Data-parallel plumbing: one process per GPU, torch.distributed 'nccl' (= RCCL over xGMI) or 'gloo' (CPU tests).

Reference: plain DDP (semilearn/core/utils/misc.py:55-58) -> bucketed gradient all-reduce, per-rank hook state.
Here parameters / gradients are ONE flat fp32 block (85.7 MB for ViT-S), so the gradient exchange is few, large collectives -- the shape that
suits xGMI's point-to-point links -- and the 1 / world scaling is folded into the AdamW launch (grad_scale).

Which exchange (``SR_GRAD_EXCHANGE``, default ``auto``):
  allreduce      one all-reduce of the flat block after the backward;
  rs_ag          the same as reduce-scatter + all-gather: every rank reduces one contiguous 1 / world shard in place (256-byte aligned) and the
                 shards are gathered back -- the two halves of an all-reduce as separate collectives, which on xGMI's all-to-all mesh (7 links per
                 GPU) can go direct between every pair of GPUs instead of around a ring bound by ONE link;
  overlap        the backward finishes its weight gradients in layer groups, last layers first, and each finished contiguous range of the block
                 is reduced on a communication stream while the backward of the earlier layers continues (install_overlap);
  rs_ag_overlap  both;
  allreduce_bf16 the block exchanged as bf16 (half the bytes; a gradient sum rounded to 8 bits of mantissa per hop, which DDP's fp32 buckets do
                 not do): explicit opt-in only, never chosen by ``auto``;
  auto           MEASURED at start-up on the live backend, agreed between the ranks (ExchangeTuner): no environment variable is needed for
                 the first run on a node whose interconnect nobody has timed yet.
Optional extension named in BASELINE.json (off by default = reference parity): a global reward threshold (``global_reward_threshold``: one packed
all-reduce of (sum reward per pass, n) per step, ``reward_means``).  The FlexMatch class histogram stays per rank, as under the reference's DDP
(srflexmatch/utils.py:24-35 recounts the rank's own ``selected_label``).
"""
import os

import torch
import torch.distributed as dist

EXCHANGES = ("allreduce", "rs_ag", "overlap", "rs_ag_overlap")           # what ``auto`` chooses between
_EXPLICIT = EXCHANGES + ("allreduce_bf16", "auto")


class ExchangeTuner:
    """Start-up selection of the gradient exchange, driven by DataParallel.all_reduce_grads (called once per step):
      step 0      both collective forms -- one all-reduce, reduce-scatter + all-gather -- are timed on a scratch block of the gradient's size
                  (1 warm + 3 timed launches each, HIP events; nothing touches the real gradient), the ranks agree on the maximum over the
                  ranks of each median (ONE small blocking all-reduce) and keep the faster form;
      then        WARM + TIMED steps with the exchange after the backward, WARM + TIMED steps with it under the backward (when the backbone
                  reports finished layer groups), step time = HIP events at consecutive exchange calls, again agreed as the maximum over
                  the ranks (one blocking read per phase) -- the faster schedule is kept.
    Every rank takes every decision from the SAME agreed numbers at the SAME step, so the ranks never issue different collectives.  While it
    runs ``settled`` is False and the algorithm's own step-schedule tuner waits (two tuners varying the step at once would time each other)."""
    WARM, TIMED, COLL_TIMED = 1, 3, 3

    def __init__(self, dp):
        self.dp = dp
        self.phase = 0                 # 0: collective forms; 1: exchange after the backward; 2: under the backward; 3: done
        self.marks = []
        self.report = {}
        self.syncs = 0

    def _agree(self, values, device):
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.syncs += 1
        return [float(x) for x in t.cpu()]

    def _probe_rs_ag(self, grad):
        """Does the live backend take the reduce-scatter + all-gather form (in place: this rank's shard is a slice of the input)?  Asked ONCE on
        a tiny block before anything large is timed, and the answer is AGREED (all-reduce MAX of a refusal flag): an error that only some ranks
        see must not leave the ranks with different collective sequences in their queues.  Returns None or the refusal text."""
        dp = self.dp
        tiny = torch.zeros(max(1, dp.world_size) * dp.SHARD_ALIGN * 2, dtype=grad.dtype, device=grad.device)
        err, keep = None, dp.collective
        try:
            dp.collective = "rs_ag"
            dp._sum_over_ranks(tiny)
            if grad.is_cuda:
                torch.cuda.current_stream().synchronize()
        except RuntimeError as exc:
            err = str(exc)[:200]
        finally:
            dp.collective = keep
        refused = self._agree([1.0 if err is not None else 0.0], grad.device)[0] > 0.0
        if refused:
            return err or "refused on another rank"
        return None

    def _time_collectives(self, grad):
        dp = self.dp
        refusal = self._probe_rs_ag(grad)
        if refusal is not None:
            self.report["rs_ag_refused"] = refusal
        scratch = torch.zeros_like(grad)
        cuda = grad.is_cuda
        out = []
        for form in ("allreduce", "rs_ag"):
            if form == "rs_ag" and refusal is not None:
                out.append(float("inf"))
                continue
            dp.collective = form
            ts = []
            for i in range(1 + self.COLL_TIMED):
                if cuda:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); dp._sum_over_ranks(scratch); e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1))
                else:
                    import time
                    t0 = time.perf_counter(); dp._sum_over_ranks(scratch); ts.append(1e3 * (time.perf_counter() - t0))
            out.append(sorted(ts[1:])[len(ts[1:]) // 2])
        del scratch                           # (86 MB for ViT-S: not kept for the tuner's lifetime)
        return out

    def _mark(self, device):
        if device.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record()
        else:
            import time
            e = time.perf_counter()
        self.marks.append(e)

    def _median_step_ms(self):
        m = self.marks
        if not isinstance(m[0], float):
            m[-1].synchronize()
            gaps = [m[j].elapsed_time(m[j + 1]) for j in range(self.WARM, len(m) - 1)]
        else:
            gaps = [1e3 * (m[j + 1] - m[j]) for j in range(self.WARM, len(m) - 1)]
        return sorted(gaps)[len(gaps) // 2]

    def step(self, model):
        """Called at the top of every all_reduce_grads while tuning.  Sets dp.collective / overlap for the exchange that follows."""
        dp, grad = self.dp, model.grad
        if self.phase == 0:
            ms = self._agree(self._time_collectives(grad), grad.device)
            dp.collective = "allreduce" if ms[0] <= ms[1] else "rs_ag"
            self.report["collective_ms"] = {"allreduce": round(ms[0], 4), "rs_ag": round(ms[1], 4) if ms[1] != float("inf") else None}
            self.phase, self.marks = 1, []
            if not dp.can_overlap(model):                   # nothing to compare a schedule with: done after one step
                self._finish(False, model)
            return
        self._mark(grad.device)
        if len(self.marks) < self.WARM + self.TIMED + 1:
            return
        ms = self._agree([self._median_step_ms()], grad.device)[0]
        if self.phase == 1:
            self.report["step_ms_exchange_after_backward"] = round(ms, 4)
            self._after = ms
            dp.install_overlap(model)                       # from the next backward on the layer groups are reduced under it
            self.phase, self.marks = 2, []
        else:
            self.report["step_ms_exchange_under_backward"] = round(ms, 4)
            self._finish(ms < self._after, model)

    def _finish(self, overlap, model):
        dp = self.dp
        if not overlap:
            # This step's backward has ALREADY reduced its layer-group ranges on the communication stream (the decision falls at the top of
            # all_reduce_grads): only the callback goes now; the exchange that follows consumes the reported ranges (reducing them again would
            # sum them twice) and clears them.
            dp.uninstall_overlap(model, keep_reported=True)
        dp.exchange = ("rs_ag" if dp.collective == "rs_ag" else "allreduce") if not overlap else ("rs_ag_overlap" if dp.collective == "rs_ag" else "overlap")
        self.report["chosen"] = dp.exchange
        self.phase = 3
        dp.tuner = None
        dp.exchange_report = dict(self.report, agreement_syncs=self.syncs)


class DataParallel:
    def __init__(self, world_size=1, rank=0, global_reward_threshold=False, exchange=None, force=None):
        self.world_size, self.rank = world_size, rank
        # force: the data-parallel path with ONE rank (the reference wraps the model in DDP whenever args.distributed is set, whatever the world
        # size: misc.py:55-58) -- every collective is issued on the live backend.  args.force_dp / SR_FORCE_DP=1; how a box with one GPU puts
        # the engine's exchanges through RCCL (tests/test_gpu_rccl_one_rank.py, bench.py --force-dp).
        self.force = bool(os.environ.get("SR_FORCE_DP", "0") != "0") if force is None else bool(force)
        self.global_reward_threshold = global_reward_threshold
        self.comm_events = None      # bench.py: list that receives a HIP-event pair around the gradient all-reduce of every step
        req = exchange if exchange is not None else os.environ.get("SR_GRAD_EXCHANGE", "auto")
        if req not in _EXPLICIT:
            raise ValueError("SR_GRAD_EXCHANGE must be one of %s, not %r" % (", ".join(_EXPLICIT), req))
        self.requested = req
        self.bf16_grads = req == "allreduce_bf16"
        self.exchange = "allreduce" if req in ("auto", "allreduce_bf16") else req           # what runs now (auto: until the tuner has decided)
        self.collective = "rs_ag" if self.exchange.startswith("rs_ag") else "allreduce"
        self.tuner = ExchangeTuner(self) if req == "auto" else None
        self.exchange_report = None if req == "auto" else {"chosen": req, "requested": True}
        self.agreement_syncs = 0     # blocking rank agreements so far (bench.py shows that none happens inside a timed region)
        self._g16 = None

    @property
    def settled(self):
        """False while the start-up selection of the gradient exchange is still measuring steps (the algorithm's schedule tuner waits for it)."""
        return not (self.active and self.tuner is not None)

    def attach(self, model):
        """Called once by the algorithm with its backbone: an explicitly requested overlapped exchange is installed here."""
        if self.active and self.exchange.endswith("overlap"):
            self.install_overlap(model)

    def can_overlap(self, model):
        return torch.cuda.is_available() and hasattr(model, "grad_ready_cb") and getattr(model.grad, "is_cuda", False)

    SHARD_ALIGN = 64                 # elements: shards of the reduce-scatter start on 256-byte boundaries

    def _sum_over_ranks(self, t):
        """Sum of a contiguous 1-D slice of a flat block over the ranks, in place: ONE all-reduce, or (rs_ag) reduce-scatter into this rank's
        shard + all-gather of the shards, with the < world * SHARD_ALIGN elements that do not divide evenly all-reduced behind them."""
        if self.collective != "rs_ag" or t.dim() != 1 or not t.is_contiguous():
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return
        # shard boundaries on ABSOLUTE 256-byte boundaries of the flat block: a slice that starts off such a boundary (a layer group of the
        # overlapped exchange) sends its first < SHARD_ALIGN elements with the tail
        w = self.world_size
        head = (-t.storage_offset()) % self.SHARD_ALIGN
        if head >= t.numel():
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return
        body = t[head:]
        n = body.numel()
        chunk = (n // (w * self.SHARD_ALIGN)) * self.SHARD_ALIGN
        main = chunk * w
        if chunk:
            shard = body[self.rank * chunk:(self.rank + 1) * chunk]
            dist.reduce_scatter_tensor(shard, body[:main], op=dist.ReduceOp.SUM)
            dist.all_gather_into_tensor(body[:main], shard)
        if main < n:
            dist.all_reduce(body[main:], op=dist.ReduceOp.SUM)
        if head:
            dist.all_reduce(t[:head], op=dist.ReduceOp.SUM)

    @property
    def engaged(self):
        """The configuration asks for the data-parallel path (more than one rank, or one rank forced through it)."""
        return self.world_size > 1 or self.force

    @property
    def active(self):
        return self.engaged and dist.is_available() and dist.is_initialized()

    # ---- all-reduce under the backward ------------------------------------------------------------------------------------------------
    # The engines finish their weight gradients in layer groups, last layers first, and report every finished contiguous range of the flat
    # gradient block; each range is all-reduced on a communication stream while the backward of the earlier layers continues (xGMI ring time
    # of the 86 MB block is otherwise serial with the step).  all_reduce_grads then reduces what was not reported and joins the stream.
    def install_overlap(self, model):
        if not (self.active and self.can_overlap(model)):
            return False
        if getattr(self, "_comm", None) is None:
            from . import ops
            # (beside the step's stream AND its second stream: the exchange of a finished layer group must not queue behind the deferred rows)
            self._comm = ops.concurrent_stream(model.grad.device, beside=[getattr(self, "side_stream", None)])
        self._done, self._model = [], model
        model.grad_ready_cb = self._reduce_range
        return True

    def uninstall_overlap(self, model, keep_reported=False):
        """keep_reported: called between a backward that reported ranges and the exchange of the same step -- the ranges stay (and with them
        the join of the communication stream) for _all_reduce_grads, which clears them."""
        if getattr(model, "grad_ready_cb", None) is not None:
            model.grad_ready_cb = None
        if not keep_reported:
            self._done, self._model = [], None

    def _reduce_range(self, lo, hi):
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)                                   # every gradient of [lo, hi) is final on the compute stream here
        self._comm.wait_event(ready)
        with torch.cuda.stream(self._comm):
            self._sum_over_ranks(self._model.grad[lo:hi])
        self._done.append((lo, hi))

    def all_reduce_grads(self, model):
        if not self.active:
            return
        if self.tuner is not None:
            self.tuner.step(model)
        if self.comm_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._all_reduce_grads(model)
            e1.record()
            self.comm_events.append((e0, e1))
            return
        self._all_reduce_grads(model)

    def _all_reduce_grads(self, model):
        done = sorted(getattr(self, "_done", ())) if getattr(self, "_model", None) is model else []
        if not done:
            if self.bf16_grads:
                g = model.grad
                if self._g16 is None or self._g16.numel() != g.numel() or self._g16.device != g.device:
                    self._g16 = torch.empty(g.numel(), dtype=torch.bfloat16, device=g.device)
                if g.is_cuda:
                    from . import ops
                    ops.cast_f32_bf16(g, self._g16, g.numel())
                else:
                    self._g16.copy_(g)
                self._sum_over_ranks(self._g16)
                g.copy_(self._g16)
                return
            self._sum_over_ranks(model.grad)
            return
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        self._comm.wait_event(ready)
        with torch.cuda.stream(self._comm):
            pos = 0
            for lo, hi in done + [(model.grad.numel(), model.grad.numel())]:
                if lo > pos:
                    self._sum_over_ranks(model.grad[pos:lo])
                pos = max(pos, hi)
            fin = torch.cuda.Event()
            fin.record(self._comm)
        main.wait_event(fin)
        self._done = []

    def all_reduce_flat(self, t):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def max_over_ranks(self, value, device):
        """A host scalar every rank must agree on (a measured time a schedule decision hangs on): its maximum over the ranks."""
        if not self.active:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.agreement_syncs += 1
        return float(t)

    def broadcast_params(self, *modules):
        """DDP broadcasts rank 0's parameters at construction; do the same for the flat blocks."""
        if self.active:
            for m in modules:
                dist.broadcast(m.flat, src=0)
                for fn in ("refresh_operands", "prepare"):            # derived copies (bf16 / transposed operands) follow the parameters
                    if hasattr(m, fn):
                        getattr(m, fn)()

    def gather_stats(self, max_probs, colsum, hist):
        """FreeMatch / SoftMatch statistics of the GLOBAL batch: the reference all-gathers [Bu, C] probabilities
        (algorithms/utils/ops.py:35-45); the sufficient statistics are enough -- column sums and histogram are all-reduced
        in place, only the Bu max-probs are gathered (needed for the quantile).  Returns (maxp_all, n_all)."""
        if colsum is not None:
            dist.all_reduce(colsum)
        if hist is not None:
            dist.all_reduce(hist)
        out = [torch.empty_like(max_probs) for _ in range(self.world_size)]
        dist.all_gather(out, max_probs.contiguous())
        allp = torch.cat(out)
        return allp, allp.numel()

    def reward_means(self, reward, groups):
        """Per-group reward mean.  Local (reference, srflexmatch.py:100) unless the global-threshold extension is on,
        in which case (sum, n) is all-reduced: one packed message of groups+1 floats."""
        B = reward.numel() // groups
        s = reward.view(groups, B).sum(dim=1)
        if self.global_reward_threshold and self.active:
            packed = torch.cat([s, torch.tensor([float(B)], device=s.device)])
            dist.all_reduce(packed)
            return packed[:-1] / packed[-1]
        return s / B


def shard_indices(n, rank, world):
    """DistributedSampler rank-stride sharding of the reference (semilearn/datasets/samplers/sampler.py:70): perm[rank::world]."""
    return list(range(rank, n, world))
