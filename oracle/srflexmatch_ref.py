"""Oracle: one SRFlexMatch training step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates the control flow of reference ``semilearn/algorithms/srflexmatch/srflexmatch.py``
  data_generator :72-104    train_step :107-217
followed by ``ParamUpdateHook.after_train_step`` (core/hooks/param_update.py:21-45):
backward, AdamW (layer-decay groups), scheduler, zero_grad.  All the reference quirks
of SURVEY.md Appendix A are reproduced (inert generator, no-op max_reward filter,
K+1 forwards with hook side effects, mean over all rows, local reward mean).

DropPath randomness is injected: ``droppath[pass][depth,2,Bt]`` (pass 0 = the forward
outside the loop, passes 1..K = data_generator).  Uses torch autograd on CPU -- it is
the checker, not the product.
"""
import numpy as np
import torch

from . import hooks_ref as H
from . import optim_ref as O
from . import semireward_ref as S
from . import vit_ref as V


class SRFlexMatchOracle:
    def __init__(self, cfg, vit_params, rewarder_params, generator_params, *,
                 num_train_iter, start_timing, N_k=10, p_cutoff=0.95, lambda_u=1.0,
                 ulb_dest_len=50000, thresh_warmup=True, sr_lr=5e-4,
                 lr=5e-4, weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=0, algorithm="srflexmatch"):
        # algorithm: 'srflexmatch' (FlexMatchThresholdingHook, needs idx_ulb) or 'srfixmatch'
        # (FixedThresholdingHook, semilearn/algorithms/srfixmatch/fixmatch.py:13-225 -- same step otherwise)
        self.algorithm = algorithm
        self.cfg = cfg
        self.P = {k: v.clone() for k, v in vit_params.items()}
        self.R = {k: v.clone() for k, v in rewarder_params.items()}
        self.G = {k: v.clone() for k, v in generator_params.items()}
        self.num_train_iter, self.start_timing, self.N_k = num_train_iter, start_timing, N_k
        self.p_cutoff, self.lambda_u = p_cutoff, lambda_u
        self.sr_lr = sr_lr
        self.hook = H.FlexMatchState(ulb_dest_len, cfg.num_classes, thresh_warmup)
        self.max_reward = -float("inf")
        self.it = 0
        # optimizer state
        self.hp = self._hparams(cfg, lr, weight_decay, layer_decay)
        self.num_warmup_iter = num_warmup_iter
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.opt_step = 0
        self.rm = {k: torch.zeros_like(v) for k, v in self.R.items()}
        self.rv = {k: torch.zeros_like(v) for k, v in self.R.items()}
        self.r_step = 0

    # -- helpers -------------------------------------------------------------------
    def _hparams(self, cfg, lr, weight_decay, layer_decay):
        return O.vit_param_hparams(V.param_shapes(cfg), cfg.depth, lr, weight_decay, layer_decay)

    def _forward(self, P, x_lb, x_ulb_w, x_ulb_s, dp):
        nl = x_lb.shape[0]
        out = V.vit_forward(P, torch.cat((x_lb, x_ulb_w, x_ulb_s)), self.cfg, droppath=dp)
        lg, ft = out["logits"], out["feat"]
        lw, ls = lg[nl:].chunk(2)
        fw, fs = ft[nl:].chunk(2)
        return lg[:nl], lw, ls, ft[:nl], fw, fs

    def _mask(self, probs, idx_ulb):
        if self.algorithm == "srfixmatch":
            return H.fixed_threshold_mask(probs.numpy(), self.p_cutoff)                      # masking.py:55-56
        return self.hook.masking(probs.numpy(), idx_ulb.numpy(), self.p_cutoff)              # srflexmatch/utils.py:38-63

    def _sr_update(self, feats, gen_labels, ref_labels):
        target = S.cosine_target(gen_labels, ref_labels, self.cfg.num_classes)
        reward, grads, lg, lr_ = S.rewarder_update_grads(self.R, feats, gen_labels, target)
        self.r_step += 1
        S.adam_step(self.R, grads, self.rm, self.rv, self.r_step, self.sr_lr)
        return dict(sr_reward=reward, sr_target=target, generator_loss=lg, rewarder_loss=lr_,
                    sr_grads=grads)

    # -- the step --------------------------------------------------------------------
    def train_step(self, x_lb, y_lb, idx_ulb, x_ulb_w, x_ulb_s, droppath):
        """droppath: list of [depth,2,Bt] tensors (or None entries), len == 1 + K."""
        it = self.it
        tr = {}
        P = {k: v.detach().clone().requires_grad_(True) for k, v in self.P.items()}
        lx, lw, ls, fx, fw, fs = self._forward(P, x_lb, x_ulb_w, x_ulb_s, droppath[0])
        sup_loss = H.ce_loss_mean(lx, y_lb)                                       # :132
        probs = H.softmax_probs(lw.detach())                                      # :135
        mask0 = torch.from_numpy(self._mask(probs, idx_ulb))                             # :141
        pl0 = torch.from_numpy(H.pseudo_label_hard(probs.numpy()))                # :142
        tr["passes"] = [dict(mask=mask0.clone(), pseudo_label=pl0.clone(),
                             classwise_acc=self.hook.classwise_acc.copy())]
        K = 0
        if it > self.start_timing:                                                # :147
            K = H.sr_decay(self.num_train_iter, it)
            for k in range(1, K + 1):                                             # :75
                _, lwk, lsk, _, fwk, _ = self._forward(P, x_lb, x_ulb_w, x_ulb_s, droppath[k])
                pk = H.softmax_probs(lwk.detach())
                plk = torch.from_numpy(H.pseudo_label_hard(pk.numpy()))
                mk = torch.from_numpy(self._mask(pk, idx_ulb))
                reward = S.rewarder_forward(self.R, fwk.detach(), plk)            # :99 (no grad flows, A.10)
                mask2 = S.reward_mask2(reward)                                    # :100-101
                unsup_loss = H.consistency_loss(lsk, plk, mk, mask2)              # :102
                tr["passes"].append(dict(mask=mk.clone(), pseudo_label=plk.clone(), reward=reward.detach().clone(),
                                         mask2=mask2.clone(), classwise_acc=self.hook.classwise_acc.copy()))
        else:
            unsup_loss = H.consistency_loss(ls, pl0, mask0)                       # :152
        tr["K"] = K

        if it > 0:                                                                # :154
            gen = S.generated_labels(self.G, fx.detach())                         # :158-159
            if it >= self.start_timing:                                           # :163
                r = S.rewarder_forward(self.R, fw.detach(), pl0).mean()           # :166-167
                r = float(r)
                if r > self.max_reward:                                           # :170
                    self.max_reward = r
                tr["mean_reward_ulb"] = r
                if it % self.N_k == 0 and it > self.start_timing:                 # :173
                    self.max_reward = -float("inf")
                    gen2 = S.generated_labels(self.G, fw.detach())                # :177-178
                    tr.update(self._sr_update(fw.detach(), gen2, pl0))            # :179-193
                    tr["sr_stage"] = 2
            else:
                tr.update(self._sr_update(fx.detach(), gen, y_lb))                # :194-208
                tr["sr_stage"] = 1

        total = sup_loss + self.lambda_u * unsup_loss                             # :210
        total.backward()                                                          # param_update.py:33
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
        tr.update(sup_loss=float(sup_loss.detach()), unsup_loss=float(unsup_loss.detach()), total_loss=float(total.detach()),
                  util_ratio=float(mask0.mean()), grads=grads,
                  feat=dict(x_lb=fx.detach(), x_ulb_w=fw.detach(), x_ulb_s=fs.detach()),
                  logits_x_lb=lx.detach(), logits_x_ulb_w=lw.detach(), logits_x_ulb_s=ls.detach())
        # optimizer.step() with lr = base * factor(it); scheduler.step(); zero_grad
        fac = O.cosine_warmup_factor(it, self.num_train_iter, self.num_warmup_iter)
        self.opt_step += 1
        for k in self.P:
            lr, wd = self.hp[k]
            O.adamw_step(self.P[k], grads[k], self.m[k], self.v[k], self.opt_step, lr * fac, wd)
        tr["lr_factor"] = fac
        self.it += 1
        return tr
