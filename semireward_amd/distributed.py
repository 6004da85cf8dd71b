"""Data-parallel plumbing: one process per GPU, torch.distributed 'nccl' (= RCCL over xGMI) or 'gloo' (CPU tests).

Reference: plain DDP (semilearn/core/utils/misc.py:55-58) -> bucketed gradient all-reduce, per-rank hook state.
Here parameters/gradients are ONE flat fp32 block, so the gradient exchange is a single large all-reduce
(85.7 MB for ViT-S) issued right after the hand-written backward -- the "few, large collectives" shape that suits
xGMI's point-to-point links -- and the 1/world scaling is folded into the AdamW launch (grad_scale).
Optional extension named in BASELINE.json (off by default = reference parity): a global reward threshold (``global_reward_threshold``:
one packed all-reduce of (sum reward per pass, n) per step, ``reward_means``).  The FlexMatch class histogram stays per rank, as under the
reference's DDP (srflexmatch/utils.py:24-35 recounts the rank's own ``selected_label``).  And (SR_ALLREDUCE_BF16=1) the gradient block exchanged as bf16 --
half the xGMI ring time of the one large all-reduce (42.9 instead of 85.7 MB for ViT-S), at the price of a gradient sum rounded to 8 bits
of mantissa per hop, which DDP's fp32 buckets do not do: opt-in, never the default.
``SR_GRAD_EXCHANGE=rs_ag`` (opt-in until it has been timed on RCCL) runs the same exchange as reduce-scatter + all-gather on the flat block: every
rank reduces one contiguous 1/world shard (in place, 256-byte aligned) and the shards are gathered back -- the two halves of an all-reduce as
separate collectives, which on xGMI's all-to-all mesh (7 links per GPU) can go direct between every pair of GPUs instead of
around a ring bound by ONE link; `bench.py --gpus N` times it beside the single all-reduce (`overlap_allreduce.rs_ag`).
"""
import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, world_size=1, rank=0, global_reward_threshold=False):
        self.world_size, self.rank = world_size, rank
        self.global_reward_threshold = global_reward_threshold
        self.comm_events = None      # bench.py: list that receives a HIP-event pair around the gradient all-reduce of every step
        self.bf16_grads = os.environ.get("SR_ALLREDUCE_BF16", "0") != "0"
        self.exchange = os.environ.get("SR_GRAD_EXCHANGE", "allreduce")       # "allreduce" | "rs_ag"
        if self.exchange not in ("allreduce", "rs_ag"):
            raise ValueError("SR_GRAD_EXCHANGE must be 'allreduce' or 'rs_ag', not %r" % (self.exchange,))
        self._g16 = None

    SHARD_ALIGN = 64                 # elements: shards of the reduce-scatter start on 256-byte boundaries

    def _sum_over_ranks(self, t):
        """Sum of a contiguous 1-D slice of a flat block over the ranks, in place: ONE all-reduce, or (rs_ag) reduce-scatter into this rank's
        shard + all-gather of the shards, with the < world * SHARD_ALIGN elements that do not divide evenly all-reduced behind them."""
        if self.exchange != "rs_ag" or t.dim() != 1 or not t.is_contiguous():
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
    def active(self):
        return self.world_size > 1 and dist.is_available() and dist.is_initialized()

    # ---- all-reduce under the backward ------------------------------------------------------------------------------------------------
    # The engines finish their weight gradients in layer groups, last layers first, and report every finished contiguous range of the flat
    # gradient block; each range is all-reduced on a communication stream while the backward of the earlier layers continues (xGMI ring time
    # of the 86 MB block is otherwise serial with the step).  all_reduce_grads then reduces what was not reported and joins the stream.
    def install_overlap(self, model):
        if not (self.active and torch.cuda.is_available() and hasattr(model, "grad_ready_cb")):
            return False
        self._comm = torch.cuda.Stream(device=model.grad.device)
        self._done, self._model = [], model
        model.grad_ready_cb = self._reduce_range
        return True

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
