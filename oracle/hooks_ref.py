"""Oracle: pseudo-label / threshold hooks and losses (TEST INFRASTRUCTURE, see oracle/__init__.py).

Integer / compare work is numpy with explicit fp32 op-by-op arithmetic (no FMA, no fp64
shortcuts) because the masks must be bit-exact:
  FlexMatchThresholdingHook  semilearn/algorithms/srflexmatch/utils.py:11-63
  FixedThresholdingHook      semilearn/algorithms/hooks/masking.py:42-57
  PseudoLabelingHook         semilearn/algorithms/hooks/pseudo_label.py:17-52
  ce_loss                    semilearn/core/criterions/cross_entropy.py:11-31
  consistency_loss           semilearn/core/criterions/consistency.py:13-45
  sr_decay                   semilearn/core/algorithmbase.py:177-183
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
