"""SRPseudoLabel (PseudoLabel + SemiReward) on the HIP engine -- registry key 'srpseudolabel'.

Reference: semilearn/algorithms/srpseudolabel/srpseudolabel.py (classification path; train_step :92-201,
data_generator :59-90).  No strong view: pass 0 forwards x_lb and x_ulb_w (two separate model calls in the reference --
identical to one batched call for a BatchNorm-free backbone; Bn_Controller :65,:76 is a no-op there), passes 1..K forward
x_ulb_w only; the self-training loss sits on the weak logits of the LAST pass; the total loss carries the unsup warm-up
``clip(it / (unsup_warm_up * num_train_iter), 0, 1)`` (:194-195).  The regression branch of the reference references an
undefined ``self.range`` (:176) and is not reproduced.
"""
import numpy as np
import torch

from .. import ops
from ..core.algorithmbase import DeferredScalar
from ..core.registry import ALGORITHMS
from .hooks import FixedThresholdingHook, PseudoLabelingHook
from .srflexmatch import SRConsistencyBase, _Plan
from .utils import SSL_Argument, str2bool


_SHARE_PASS_LAUNCHES = True      # BatchNorm backbone: the K + 1 forwards of x_ulb_w share their launches (module constant: tests flip it)


@ALGORITHMS.register("srpseudolabel")
class SRPseudoLabel(SRConsistencyBase):
    def _init_thresholds(self, args):
        self.init(p_cutoff=args.p_cutoff, unsup_warm_up=getattr(args, "unsup_warm_up", 0.4))
        self.task_type = "cls"

    def init(self, p_cutoff, unsup_warm_up=0.4):
        self.p_cutoff, self.unsup_warm_up = p_cutoff, unsup_warm_up

    def set_hooks(self):
        self.register_hook(PseudoLabelingHook(), "PseudoLabelingHook")
        self.register_hook(FixedThresholdingHook(), "MaskingHook")
        super().set_hooks()

    def _plan(self, nl, nu, K):
        key = ("pl", nl, nu, K)
        if key not in self._plans:
            cols_img = list(range(nl + nu)) + [nl + j for _ in range(K) for j in range(nu)]   # imgs = cat(x_lb, x_ulb_w)
            last = (nl + nu) + (K - 1) * nu if K > 0 else nl
            pl = _Plan(cols_img, list(range(nl)) + list(range(last, last + nu)), self.device)
            pl.weak = [slice(nl, nl + nu)] + [slice(nl + nu + k * nu, nl + nu + (k + 1) * nu) for k in range(K)]
            self._plans[key] = pl
        return self._plans[key]

    def train_step(self, x_lb, y_lb, x_ulb_w):
        with ops.stream_scope():
            return self._step(x_lb, y_lb, x_ulb_w)

    def _step(self, x_lb, y_lb, x_ulb_w):
        it = self.it
        K = self.sr_decay() if it > self.start_timing else 0                                        # :126, :62
        P, C = K + 1, self.num_classes
        if getattr(self.model, "takes_tokens", False):       # usb_nlp: dict batches, padded separately (see SRConsistencyBase._train_step)
            tb = [self._tokens(x) for x in (x_lb, x_ulb_w)]
            nl, nu = tb[0].S, tb[1].S
            imgs = self._token_cat(tb)
        else:
            nl, nu = y_lb.shape[0], x_ulb_w.shape[0]
            imgs = torch.cat((x_lb, x_ulb_w)).contiguous()
        pl = self._plan(nl, nu, K)
        dpc = None
        if self.inject_droppath is not None:        # tests: [ (dp_lb, dp_ulb), dp_ulb(pass 1), ... ]
            dpc = torch.cat([self.inject_droppath[0][0], self.inject_droppath[0][1]] + list(self.inject_droppath[1:P]), dim=2)
        bn_backbone = getattr(self.model, "couples_batch_rows", False)
        if bn_backbone:
            # BatchNorm backbone (classic_cv, WRN): every model call of the reference is its own statistics group, so the calls stay
            # separate launches -- model(x_lb) moves the running statistics (:96), every model(x_ulb_w) runs under Bn_Controller.freeze_bn
            # (:100-110, :65-76).  Only the labelled forward and the LAST unlabelled forward carry a gradient.
            # The K + 1 forwards of x_ulb_w (K data_generator passes + the one whose loss is kept) are K + 1 statistics groups over the same
            # batch: they share ONE launch per convolution (WideResNet.forward_passes; 28 launches instead of 28 (K + 1), each large enough
            # to amortise its statistics prologue / epilogue); every pass is computed, the last one keeps its activations for the backward.
            xl, xu = x_lb.contiguous(), x_ulb_w.contiguous()
            if _SHARE_PASS_LAUNCHES and tuple(xl.shape) == tuple(xu.shape):
                # ... and model(x_lb) (:96, the call that moves the running statistics) rides along as statistics group 0 when the two
                # batches have one shape (uratio 1, every classic_cv SR setting)
                logits, feats, ctx = self.model.forward_passes(xu, P, tag="ulb", first_img=xl)
            else:
                lg_lb, ft_lb, ctx_lb = self.model.forward_saved(xl, update_stats=True, tag="lb")
                if _SHARE_PASS_LAUNCHES:
                    lg_u, ft_u, ctx_u = self.model.forward_passes(xu, P, tag="ulb")
                else:            # one launch train per pass (the comparison the tests / A-B runs flip to)
                    outs = [self.model.forward_frozen(xu, tag="ulb_inf") for _ in range(K)]
                    lg_k, ft_k, ctx_u = self.model.forward_saved(xu, update_stats=False, tag="ulb")
                    lg_u, ft_u = torch.cat([o[0] for o in outs] + [lg_k]), torch.cat([o[1] for o in outs] + [ft_k])
                logits, feats, ctx = torch.cat((lg_lb, lg_u)), torch.cat((ft_lb, ft_u)), (ctx_lb, ctx_u)
        else:
            logits, feats, ctx = self._forward_plan(imgs, pl, dpc)
            self._join_grad()
        # weak logits of every pass -> softmax max / argmax in one launch (FixedThresholdingHook softmaxes logits, masking.py:48-50)
        Lw = torch.cat([logits[sl] for sl in pl.weak])                                              # [P*nu, C]
        Fw = torch.cat([feats[sl] for sl in pl.weak])
        mp = torch.empty(P * nu, dtype=torch.float32, device=self.device)
        mi = torch.empty(P * nu, dtype=torch.int64, device=self.device)
        ops.row_max(Lw, False, None, mp, mi, P * nu, C)
        m_all = torch.empty_like(mp)
        ops.fixed_mask(mp, float(self.p_cutoff), m_all, P * nu)                                     # :119 / :78
        masks = [m_all[k * nu:(k + 1) * nu] for k in range(P)]
        pl0 = mi[:nu]
        warm = float(np.clip(it / (self.unsup_warm_up * self.num_train_iter), 0.0, 1.0))            # :194
        sup_loss, dl_lb = self.ce_loss(logits[:nl], y_lb, reduction="mean")                         # :116
        if K > 0:
            self.rewarder.eval()
            reward = self.rewarder.score(Fw[nu:].contiguous(), mi[nu:], groups=K)                   # :84
            mask2 = torch.empty_like(reward)
            ops.reward_mask2(reward, mask2, None, K, nu)                                            # :85-86
            unsup_loss, dl_u = self.consistency_loss(Lw[K * nu:], mi[K * nu:], "ce", mask=masks[K], mask2=mask2[(K - 1) * nu:],
                                                     grad_scale=self.lambda_u * warm)               # :87
        else:
            reward = mask2 = None
            unsup_loss, dl_u = self.consistency_loss(Lw[:nu], pl0, "ce", mask=masks[0], grad_scale=self.lambda_u * warm)   # :130
        if bn_backbone:
            self.model.backward(ctx[0], dl_lb)
            self.model.backward(ctx[1], dl_u)
        else:
            self.model.backward(ctx, torch.cat((dl_lb, dl_u)))
        fx, fu0 = feats[:nl], feats[nl:nl + nu]
        if it > 0:                                                                                  # :135-191
            if it >= self.start_timing:
                r0 = self.rewarder.score(fu0.contiguous(), pl0)
                rm = r0.mean()
                self.max_reward = torch.where(rm > self.max_reward, rm, self.max_reward)     # srpseudolabel.py:153 (a NaN mean keeps the old maximum)
                if it % self.N_k == 0 and it > self.start_timing:
                    self.max_reward = torch.full((), -float("inf"), device=self.device)
                    gen2 = self.generator.forward_with_labels(fu0.contiguous())[1]
                    self._sr_update(fu0, gen2, pl0)
            else:
                gen = self.generator.forward_with_labels(fx.contiguous())[1]
                self._sr_update(fx, gen, y_lb)
        total_loss = sup_loss + self.lambda_u * unsup_loss * warm                                   # :195
        if self.trace is not None:
            self.trace.update(K=K, masks=masks, max_probs=mp, pseudo=mi, reward=reward, mask2=mask2, unsup_warmup=warm)
        out_dict = self.process_out_dict(loss=total_loss, feat={"x_lb": fx, "x_ulb_w": fu0})
        log_dict = self.process_log_dict(sup_loss=DeferredScalar(sup_loss), unsup_loss=DeferredScalar(unsup_loss),
                                         total_loss=DeferredScalar(total_loss), util_ratio=DeferredScalar(masks[0].mean()))
        return out_dict, log_dict

    @staticmethod
    def get_argument():
        return [SSL_Argument("--p_cutoff", float, 0.95), SSL_Argument("--unsup_warm_up", float, 0.4, "warm up ratio for unsupervised loss"),
                SSL_Argument("--task_type", str, "cls"), SSL_Argument("--start_timing", int, 20000),
                SSL_Argument("--feature_dim", int, 384), SSL_Argument("--sr_lr", float, 0.0005), SSL_Argument("--N_k", int, 10),
                SSL_Argument("--sr_ema", str2bool, True), SSL_Argument("--sr_ema_m", float, 0.999), SSL_Argument("--range", int, 100)]
