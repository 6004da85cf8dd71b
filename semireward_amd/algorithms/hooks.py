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
    def masking_passes(self, algorithm, max_probs, max_idx, idx_ulb, n_pass):
        """The masking calls of the n_pass data_generator passes of one step (same idx_ulb, pass order) in one launch: [n_pass * nu] in,
        list of n_pass masks out; state ends where n_pass ``masking_from_max`` calls would leave it."""
        mask = torch.empty_like(max_probs)
        nu = max_probs.numel() // n_pass
        ops.flexmatch_mask_passes(max_probs, max_idx, idx_ulb.contiguous(), float(algorithm.p_cutoff), self._sel, self.hist,
                                  self.classwise_acc, mask, n_pass, nu, self.num_classes, self.ulb_dest_len, self.thresh_warmup)
        return [mask[k * nu:(k + 1) * nu] for k in range(n_pass)]

    @torch.no_grad()
    def masking(self, algorithm, logits_x_ulb, idx_ulb, softmax_x_ulb=True, *a, **k):
        mp, mi = _row_max(logits_x_ulb, is_probs=not softmax_x_ulb)
        return self.masking_from_max(algorithm, mp, mi, idx_ulb)


class FreeMatchThresholdingHook(MaskingHook):
    """Self-adaptive thresholding of FreeMatch (semilearn/algorithms/freematch/utils.py:10-66): EMA of the global confidence
    (``time_p``, mean or 0.8-quantile of the max-probs), of the class marginals (``p_model``) and of the predicted-label
    histogram (``label_hist``); state advances at EVERY masking call.  Same attribute names as the reference so the
    algorithm's get_save_dict / load_model keep working."""

    def __init__(self, num_classes, momentum=0.999, device="cuda", *a, **k):
        super().__init__()
        self.num_classes, self.m = num_classes, momentum
        self.p_model = torch.ones(num_classes, dtype=torch.float32, device=device) / num_classes
        self.label_hist = torch.ones(num_classes, dtype=torch.float32, device=device) / num_classes
        self.time_p = self.p_model.mean().reshape(1)
        self._colsum = torch.empty(num_classes, dtype=torch.float32, device=device)
        self._hist = torch.empty(num_classes, dtype=torch.float32, device=device)

    @torch.no_grad()
    def masking_from_probs(self, algorithm, probs, max_probs, max_idx):
        B, C = probs.shape
        ops.freematch_stats(probs, max_idx, self._colsum, self._hist, B, C)
        maxp_all, n_all = max_probs, B
        dp = getattr(algorithm, "dp", None)
        if dp is not None and dp.active:                      # reference: concat_all_gather(probs) (utils.py:25-26)
            maxp_all, n_all = dp.gather_stats(max_probs, self._colsum, self._hist)
        mask = torch.empty_like(max_probs)
        ops.freematch_update(maxp_all, n_all, self._colsum, self._hist, max_probs, max_idx, self.time_p, self.p_model, self.label_hist,
                             mask, B, C, self.m, bool(algorithm.use_quantile), bool(algorithm.clip_thresh))
        algorithm.p_model, algorithm.label_hist, algorithm.time_p = self.p_model, self.label_hist, self.time_p   # utils.py:41-43
        return mask

    @torch.no_grad()
    def masking(self, algorithm, logits_x_ulb, softmax_x_ulb=True, *a, **k):
        B, C = logits_x_ulb.shape
        probs = torch.empty(B, C, dtype=torch.float32, device=logits_x_ulb.device)
        mp = torch.empty(B, dtype=torch.float32, device=probs.device)
        mi = torch.empty(B, dtype=torch.int64, device=probs.device)
        ops.row_max(logits_x_ulb.contiguous(), not softmax_x_ulb, probs, mp, mi, B, C)
        return self.masking_from_probs(algorithm, probs, mp, mi)


class DistAlignEMAHook(Hook):
    """EMA distribution alignment (semilearn/algorithms/hooks/dist_align.py:10-71), p_target_type 'uniform' or 'model'.  Same
    attribute names as the reference (``p_model``, ``p_target``) so get_save_dict / load_model keep working; ``p_model`` holds
    zeros until the first call (the reference keeps None)."""

    def __init__(self, num_classes, momentum=0.999, p_target_type="uniform", p_target=None, device="cuda"):
        super().__init__()
        assert p_target_type in ("uniform", "model"), "p_target_type 'gt' (a given prior) is not used by the SemiReward configs"
        self.num_classes, self.m = num_classes, momentum
        self.update_p_target = p_target_type == "model"
        self.p_target = torch.ones(num_classes, dtype=torch.float32, device=device) / num_classes
        self.p_model = torch.zeros(num_classes, dtype=torch.float32, device=device)
        self.inited = torch.zeros(1, dtype=torch.int32, device=device)
        self._cs_u = torch.empty(num_classes, dtype=torch.float32, device=device)
        self._cs_l = torch.empty(num_classes, dtype=torch.float32, device=device)
        self._hist = torch.empty(num_classes, dtype=torch.float32, device=device)

    @torch.no_grad()
    def align(self, algorithm, probs_x_ulb, probs_x_lb=None):
        """Returns (aligned probabilities [B,C], their row max [B], argmax [B])."""
        B, C = probs_x_ulb.shape
        dev = probs_x_ulb.device
        zi = torch.zeros(max(B, probs_x_lb.shape[0] if probs_x_lb is not None else 0), dtype=torch.int64, device=dev)
        ops.freematch_stats(probs_x_ulb.contiguous(), zi, self._cs_u, self._hist, B, C)                # column sums of the local rows
        n_u, n_l, cs_l = B, 0, None
        if self.update_p_target:
            assert probs_x_lb is not None
            n_l = probs_x_lb.shape[0]
            ops.freematch_stats(probs_x_lb.contiguous(), zi, self._cs_l, self._hist, n_l, C)
            cs_l = self._cs_l
        dp = getattr(algorithm, "dp", None)
        if dp is not None and dp.active:                      # reference: concat_all_gather of both batches (dist_align.py:42-45)
            dp.all_reduce_flat(self._cs_u)
            n_u *= dp.world_size
            if cs_l is not None:
                dp.all_reduce_flat(cs_l)
                n_l *= dp.world_size
        aligned = torch.empty(B, C, dtype=torch.float32, device=dev)
        mp = torch.empty(B, dtype=torch.float32, device=dev)
        mi = torch.empty(B, dtype=torch.int64, device=dev)
        ops.distalign(probs_x_ulb.contiguous(), self._cs_u, n_u, cs_l, n_l, self.p_model, self.p_target, self.inited, self.m, aligned, mp, mi, B, C)
        return aligned, mp, mi

    @torch.no_grad()
    def dist_align(self, algorithm, probs_x_ulb, probs_x_lb=None):
        return self.align(algorithm, probs_x_ulb, probs_x_lb)[0]


class SoftMatchWeightingHook(MaskingHook):
    """SoftMatch truncated-Gaussian sample weighting (semilearn/algorithms/srsoftmatch/utils.py:12-76), per_class False (the configs'
    value).  The EMA mean / variance advance at EVERY masking call.  ``prob_max_mu_t`` / ``prob_max_var_t`` as in the reference."""

    def __init__(self, num_classes, n_sigma=2, momentum=0.999, per_class=False, device="cuda", *a, **k):
        super().__init__()
        assert not per_class, "per_class SoftMatch statistics are not on the SemiReward hot path (configs use per_class False)"
        self.num_classes, self.n_sigma, self.m = num_classes, n_sigma, momentum
        self.mu_var = torch.tensor([1.0 / num_classes, 1.0], dtype=torch.float32, device=device)

    @property
    def prob_max_mu_t(self):
        return self.mu_var[0]

    @property
    def prob_max_var_t(self):
        return self.mu_var[1]

    @torch.no_grad()
    def masking_from_max(self, algorithm, max_probs):
        B = max_probs.numel()
        maxp_all, n_all = max_probs, B
        dp = getattr(algorithm, "dp", None)
        if dp is not None and dp.active:                      # reference: concat_all_gather(probs) (utils.py:34-35)
            maxp_all, n_all = dp.gather_stats(max_probs, None, None)
        mask = torch.empty_like(max_probs)
        ops.softmatch_mask(maxp_all.contiguous(), n_all, max_probs.contiguous(), self.mu_var, self.m, self.n_sigma, mask, B)
        return mask

    @torch.no_grad()
    def masking(self, algorithm, logits_x_ulb, softmax_x_ulb=True, *a, **k):
        mp, _ = _row_max(logits_x_ulb, is_probs=not softmax_x_ulb)
        return self.masking_from_max(algorithm, mp)
