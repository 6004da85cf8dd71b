"""Oracle: pseudo-label / threshold hooks and losses (TEST INFRASTRUCTURE, see oracle/__init__.py).

Integer / compare work is numpy with explicit fp32 op-by-op arithmetic (no FMA, no fp64
shortcuts) because the masks must be bit-exact:
  FlexMatchThresholdingHook  semilearn/algorithms/srflexmatch/utils.py:11-63
  FixedThresholdingHook      semilearn/algorithms/hooks/masking.py:42-57
  PseudoLabelingHook         semilearn/algorithms/hooks/pseudo_label.py:17-52
  ce_loss                    semilearn/core/criterions/cross_entropy.py:11-31
  consistency_loss           semilearn/core/criterions/consistency.py:13-45
  sr_decay                   semilearn/core/algorithmbase.py:177-183
  SoftMatchWeightingHook     semilearn/algorithms/srsoftmatch/utils.py:12-76
  DistAlignEMAHook           semilearn/algorithms/hooks/dist_align.py:10-71
"""
import numpy as np
import torch

F32 = np.float32


def sr_decay(num_train_iter, it, max_sampling_time=8):
    """algorithmbase.py:177-183 (note: ``max`` -> K >= 8 forever)."""
    return int(max(max_sampling_time, 1 + num_train_iter / it))


def softmax_probs(logits):
    """AlgorithmBase.compute_prob, algorithmbase.py:332-333."""
    return torch.softmax(logits, dim=-1)


def pseudo_label_hard(probs):
    """pseudo_label.py:40 -- argmax over classes (first maximal index, as torch.argmax)."""
    return np.argmax(np.asarray(probs), axis=-1).astype(np.int64)


def fixed_threshold_mask(probs, p_cutoff):
    """masking.py:55-56."""
    mp = np.asarray(probs, dtype=F32).max(axis=-1)
    return (mp >= F32(p_cutoff)).astype(F32)


class FlexMatchState:
    """srflexmatch/utils.py:15-21 state: selected_label[ulb_dest_len] i64 = -1, classwise_acc[C] f32 = 0."""

    def __init__(self, ulb_dest_len, num_classes, thresh_warmup=True):
        self.ulb_dest_len = ulb_dest_len
        self.num_classes = num_classes
        self.thresh_warmup = thresh_warmup
        self.selected_label = -np.ones((ulb_dest_len,), dtype=np.int64)
        self.classwise_acc = np.zeros((num_classes,), dtype=F32)

    def update(self):
        """utils.py:24-35.  Counter over selected_label incl. the -1 bucket; python float
        division then fp32 store == correctly rounded fp32 quotient of the two integers."""
        sel = self.selected_label
        cnt = np.bincount(sel[sel >= 0], minlength=self.num_classes).astype(np.int64)
        n_unused = int((sel == -1).sum())
        mx_all = max(int(cnt.max()) if cnt.size else 0, n_unused)
        if mx_all < self.ulb_dest_len:                       # :26
            if self.thresh_warmup:
                den = mx_all                                  # :29 includes the -1 bucket
            else:
                den = int(cnt.max())                          # :31-35 without the -1 bucket
            for i in range(self.num_classes):
                self.classwise_acc[i] = F32(float(cnt[i]) / float(den))

    def masking(self, probs, idx_ulb, p_cutoff):
        """utils.py:38-63.  probs [Bu,C] f32 (already softmaxed), idx_ulb [Bu] i64.
        Returns mask [Bu] f32 in {0,1}; mutates selected_label / classwise_acc."""
        probs = np.asarray(probs, dtype=F32)
        idx_ulb = np.asarray(idx_ulb, dtype=np.int64)
        max_probs = probs.max(axis=-1)
        max_idx = probs.argmax(axis=-1)
        acc = self.classwise_acc[max_idx]                     # f32
        # :53 evaluated op by op in fp32: fl(0.95) * (acc / (fl(2) - acc))
        thr = F32(p_cutoff) * (acc / (F32(2.0) - acc))
        mask = (max_probs >= thr).astype(F32)
        select = max_probs >= F32(p_cutoff)
        if select.any():                                      # :59-60
            self.selected_label[idx_ulb[select]] = max_idx[select]
        self.update()                                         # :61
        return mask


def ce_loss_rows(logits, targets):
    """cross_entropy.py:29-30 with reduction='none' (hard targets)."""
    logp = torch.log_softmax(logits, dim=-1)
    return -logp.gather(1, targets.view(-1, 1)).squeeze(1)


def ce_loss_mean(logits, targets):
    return ce_loss_rows(logits, targets).mean()


def consistency_loss(logits, targets, mask=None, mask2=None):
    """consistency.py:38-45 ('ce'): mean over ALL rows of nll*mask*mask2."""
    loss = ce_loss_rows(logits, targets)
    if mask is not None:
        loss = loss * mask
    if mask2 is not None:
        loss = loss * mask2
    return loss.mean()


class FreeMatchState:
    """freematch/utils.py:10-66 (self-adaptive thresholds).  torch-CPU fp32, same op order as the reference so the
    EMA state is reproduced bit for bit.  ``masking`` = update(probs) then mask (utils.py:47-65)."""

    def __init__(self, num_classes, momentum=0.999, use_quantile=True, clip_thresh=False):
        self.num_classes, self.m = num_classes, momentum
        self.use_quantile, self.clip_thresh = use_quantile, clip_thresh
        self.p_model = torch.ones(num_classes) / num_classes
        self.label_hist = torch.ones(num_classes) / num_classes
        self.time_p = self.p_model.mean()

    def update(self, probs):
        max_probs, max_idx = torch.max(probs, dim=-1, keepdim=True)
        if self.use_quantile:
            self.time_p = self.time_p * self.m + (1 - self.m) * torch.quantile(max_probs, 0.8)     # :30
        else:
            self.time_p = self.time_p * self.m + (1 - self.m) * max_probs.mean()                   # :32
        if self.clip_thresh:
            self.time_p = torch.clip(self.time_p, 0.0, 0.95)
        self.p_model = self.p_model * self.m + (1 - self.m) * probs.mean(dim=0)                    # :37
        hist = torch.bincount(max_idx.reshape(-1), minlength=self.num_classes).to(self.p_model.dtype)
        self.label_hist = self.label_hist * self.m + (1 - self.m) * (hist / hist.sum())            # :38-39

    def masking(self, probs):
        probs = torch.as_tensor(probs, dtype=torch.float32)
        self.update(probs)
        max_probs, max_idx = probs.max(dim=-1)
        mod = self.p_model / torch.max(self.p_model, dim=-1)[0]                                    # :63
        return max_probs.ge(self.time_p * mod[max_idx]).to(max_probs.dtype)                        # :64


def freematch_entropy_loss(mask, logits_s, prob_model, label_hist):
    """srfreematch.py:16-44 (fairness / entropy term).  Returns the loss (0-d tensor, differentiable w.r.t. logits_s)."""
    sel = mask.bool()
    ls = logits_s[sel]
    prob_s = ls.softmax(dim=-1)
    pred = prob_s.argmax(dim=-1)
    hist_s = torch.bincount(pred, minlength=ls.shape[1]).to(ls.dtype)
    hist_s = hist_s / hist_s.sum()
    inv = lambda v: torch.where(torch.isinf(1 / v), torch.zeros_like(v), 1 / v)   # noqa: E731  replace_inf_to_zero(1 / v)
    mod_prob_model = prob_model.reshape(1, -1) * inv(label_hist.reshape(1, -1)).detach()
    mod_prob_model = mod_prob_model / mod_prob_model.sum(dim=-1, keepdim=True)
    mod_mean = prob_s.mean(dim=0, keepdim=True) * inv(hist_s).detach()
    mod_mean = mod_mean / mod_mean.sum(dim=-1, keepdim=True)
    return (mod_prob_model * torch.log(mod_mean + 1e-12)).sum(dim=1).mean()


class DistAlignState:
    """dist_align.py:10-71 (EMA distribution alignment).  torch-CPU fp32, same op order as the reference.
    p_target_type 'uniform' (SoftMatch default, fixed 1/C) or 'model' (EMA of the labelled-batch marginals)."""

    def __init__(self, num_classes, momentum=0.999, p_target_type="uniform"):
        assert p_target_type in ("uniform", "model")
        self.m = momentum
        self.update_p_target = p_target_type == "model"
        self.p_target = torch.ones((num_classes,)) / num_classes                                    # :62-66
        self.p_model = None

    def dist_align(self, probs_x_ulb, probs_x_lb=None):
        probs_x_ulb = torch.as_tensor(probs_x_ulb, dtype=torch.float32)
        if self.p_model is None:                                                                    # :49-52
            self.p_model = torch.mean(probs_x_ulb, dim=0)
        else:
            self.p_model = self.p_model * self.m + torch.mean(probs_x_ulb, dim=0) * (1 - self.m)
        if self.update_p_target:                                                                    # :54-56
            self.p_target = self.p_target * self.m + torch.mean(torch.as_tensor(probs_x_lb, dtype=torch.float32), dim=0) * (1 - self.m)
        aligned = probs_x_ulb * (self.p_target + 1e-6) / (self.p_model + 1e-6)                       # :31
        return aligned / aligned.sum(dim=-1, keepdim=True)                                           # :32


class SoftMatchState:
    """srsoftmatch/utils.py:12-76 with per_class=False (the configs' value): EMA of mean / unbiased variance of the max-probs,
    truncated-Gaussian sample weight.  ``.item()`` makes the new statistic a python double before it meets the fp32 state."""

    def __init__(self, num_classes, n_sigma=2, momentum=0.999):
        self.n_sigma, self.m = n_sigma, momentum
        self.mu = torch.tensor(1.0 / num_classes)                                                   # :24
        self.var = torch.tensor(1.0)                                                                # :25

    def masking(self, probs):
        probs = torch.as_tensor(probs, dtype=torch.float32)
        max_probs, _ = probs.max(dim=-1)
        mu_b, var_b = torch.mean(max_probs), torch.var(max_probs, unbiased=True)                    # :38-39
        self.mu = self.m * self.mu + (1 - self.m) * mu_b.item()                                     # :40
        self.var = self.m * self.var + (1 - self.m) * var_b.item()                                  # :41
        return torch.exp(-((torch.clamp(max_probs - self.mu, max=0.0) ** 2) / (2 * self.var / (self.n_sigma ** 2))))   # :75
