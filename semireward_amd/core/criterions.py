"""Losses with fused analytic backward (semilearn/core/criterions/cross_entropy.py:11-31, consistency.py:13-45).

Each call returns (loss 0-d device tensor, dlogits [B,C]) from ONE launch; dlogits already carries the
loss weight (``grad_scale``) so the backbone backward can start from it directly.  ``dl_out`` (a contiguous [B,C] row block of a larger
buffer): the gradient is written there -- the step's losses fill ONE upstream-gradient buffer instead of being torch.cat'ed."""
import torch

from .. import ops


class CELoss:
    def __call__(self, logits, targets, reduction="mean", grad_scale=1.0, want_grad=True, dl_out=None):
        assert reduction == "mean", "the SemiReward hot path only uses reduction='mean' for the supervised loss"
        B, C = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = (dl_out if dl_out is not None else torch.empty_like(logits)) if want_grad else None
        ops.masked_ce(logits.contiguous(), targets.contiguous(), None, None, grad_scale, loss, dl, B, C)
        return loss[0], dl


class ConsistencyLoss:
    def __call__(self, logits, targets, name="ce", mask=None, mask2=None, grad_scale=1.0, want_grad=True, dl_out=None):
        assert name == "ce", "hard-label 'ce' is the only consistency loss on the SemiReward classification path"
        B, C = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = (dl_out if dl_out is not None else torch.empty_like(logits)) if want_grad else None
        ops.masked_ce(logits.contiguous(), targets.contiguous(), mask, mask2, grad_scale, loss, dl, B, C)
        return loss[0], dl
