"""Pseudo-label / masking hooks on device (state never leaves the GPU).

  PseudoLabelingHook          semilearn/algorithms/hooks/pseudo_label.py:17-52
  MaskingHook / Fixed...      semilearn/algorithms/hooks/masking.py:9-57
  FlexMatchThresholdingHook   semilearn/algorithms/srflexmatch/utils.py:11-63
Same hook names, same call signatures (``call_hook("masking", "MaskingHook", logits_x_ulb=..., idx_ulb=...)``),
same state attributes (``selected_label`` int64, ``classwise_acc`` fp32) so get_save_dict / load_model keep working.
"""
import torch

from .. import ops
from ..core.hooks import Hook


def _row_max(x, is_probs):
    B, C = x.shape
    mp = torch.empty(B, dtype=torch.float32, device=x.device)
    mi = torch.empty(B, dtype=torch.int64, device=x.device)
    ops.row_max(x.contiguous(), is_probs, None, mp, mi, B, C)
    return mp, mi


class PseudoLabelingHook(Hook):
    @torch.no_grad()
    def gen_ulb_targets(self, algorithm, logits, use_hard_label=True, T=1.0, softmax=True, label_smoothing=0.0):
        assert use_hard_label and not label_smoothing, "SemiReward needs hard integer labels (SURVEY.md A.4)"
        # argmax(softmax(z)) == argmax(z); with softmax=False the input already holds probabilities
        return _row_max(logits, is_probs=not softmax)[1]


class MaskingHook(Hook):
    def update(self, *a, **k):
        pass

    def masking(self, algorithm, logits_x_lb=None, logits_x_ulb=None, idx_lb=None, idx_ulb=None, softmax_x_lb=True,
                softmax_x_ulb=True, *a, **k):
        raise NotImplementedError


class FixedThresholdingHook(MaskingHook):
    @torch.no_grad()
    def masking(self, algorithm, logits_x_ulb, softmax_x_ulb=True, *a, **k):
        mp, _ = _row_max(logits_x_ulb, is_probs=not softmax_x_ulb)
        mask = torch.empty_like(mp)
        ops.fixed_mask(mp, float(algorithm.p_cutoff), mask, mp.numel())
        return mask


class FlexMatchThresholdingHook(MaskingHook):
    def __init__(self, ulb_dest_len, num_classes, thresh_warmup=True, device="cuda", *a, **k):
        super().__init__()
        self.ulb_dest_len, self.num_classes, self.thresh_warmup = ulb_dest_len, num_classes, thresh_warmup
        self._sel = torch.full((ulb_dest_len,), -1, dtype=torch.int64, device=device)
        self.classwise_acc = torch.zeros(num_classes, dtype=torch.float32, device=device)
        self.hist = torch.zeros(num_classes + 1, dtype=torch.int32, device=device)
        ops.flexmatch_rebuild_hist(self._sel, self.hist, ulb_dest_len, num_classes)

    # selected_label is assignable (load_model does it, srflexmatch.py:226-231): keep the histogram in sync
    @property
    def selected_label(self):
        return self._sel

    @selected_label.setter
    def selected_label(self, v):
        self._sel = v.to(self.hist.device, torch.int64).contiguous()
        ops.flexmatch_rebuild_hist(self._sel, self.hist, self.ulb_dest_len, self.num_classes)

    @torch.no_grad()
    def masking_from_max(self, algorithm, max_probs, max_idx, idx_ulb):
        mask = torch.empty_like(max_probs)
        ops.flexmatch_mask(max_probs, max_idx, idx_ulb.contiguous(), float(algorithm.p_cutoff), self._sel, self.hist,
                           self.classwise_acc, mask, max_probs.numel(), self.num_classes, self.ulb_dest_len, self.thresh_warmup)
        return mask

    @torch.no_grad()
    def masking(self, algorithm, logits_x_ulb, idx_ulb, softmax_x_ulb=True, *a, **k):
        mp, mi = _row_max(logits_x_ulb, is_probs=not softmax_x_ulb)
        return self.masking_from_max(algorithm, mp, mi, idx_ulb)
