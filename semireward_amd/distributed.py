"""Data-parallel plumbing: one process per GPU, torch.distributed 'nccl' (= RCCL over xGMI) or 'gloo' (CPU tests).

Reference: plain DDP (semilearn/core/utils/misc.py:55-58) -> bucketed gradient all-reduce, per-rank hook state.
Here parameters/gradients are ONE flat fp32 block, so the gradient exchange is a single large all-reduce
(85.7 MB for ViT-S) issued right after the hand-written backward -- the "few, large collectives" shape that suits
xGMI's point-to-point links -- and the 1/world scaling is folded into the AdamW launch (grad_scale).
Optional extensions named in BASELINE.json (off by default = reference parity): a global reward threshold
(all-reduce of (sum reward, n)) and a global FlexMatch class histogram.
"""
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, world_size=1, rank=0, global_reward_threshold=False):
        self.world_size, self.rank = world_size, rank
        self.global_reward_threshold = global_reward_threshold

    @property
    def active(self):
        return self.world_size > 1 and dist.is_available() and dist.is_initialized()

    def all_reduce_grads(self, model):
        if self.active:
            dist.all_reduce(model.grad, op=dist.ReduceOp.SUM)

    def all_reduce_flat(self, t):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

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
