"""Oracle: one SRPseudoLabel training step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates reference ``semilearn/algorithms/srpseudolabel/srpseudolabel.py`` (classification path):
  data_generator :59-90    train_step :92-201
Differences from SRFlexMatch: no strong view; labelled and unlabeled batches go through SEPARATE forwards
(pass 0 = model(x_lb) + model(x_ulb_w), passes 1..K = model(x_ulb_w) only); the masking hook is fed LOGITS
(softmax inside, masking.py:48-50); the unsup loss is on the weak logits themselves (self-training);
total = sup + lambda_u * unsup * clip(it / (unsup_warm_up * num_train_iter), 0, 1)   (:194-195).
Bn_Controller (:65,76) is a no-op for the ViT backbones of the usb_cv SemiReward configs (no BatchNorm).
DropPath injection: droppath[0] = (dp_lb [depth,2,Bl], dp_ulb [depth,2,Bu]); droppath[k>=1] = dp_ulb.
"""
import numpy as np
import torch

from . import hooks_ref as H
from . import optim_ref as O
from . import semireward_ref as S
from . import vit_ref as V
from .srflexmatch_ref import SRFlexMatchOracle


class SRPseudoLabelOracle(SRFlexMatchOracle):
    def __init__(self, *a, unsup_warm_up=0.4, **k):
        k["algorithm"] = "srfixmatch"          # FixedThresholdingHook
        super().__init__(*a, **k)
        self.unsup_warm_up = unsup_warm_up

    def train_step(self, x_lb, y_lb, x_ulb_w, droppath):
        it = self.it
        tr = {}
        P = {k: v.detach().clone().requires_grad_(True) for k, v in self.P.items()}
        dpl, dpu = droppath[0]
        o_lb = V.vit_forward(P, x_lb, self.cfg, droppath=dpl)                         # :96
        o_u = V.vit_forward(P, x_ulb_w, self.cfg, droppath=dpu)                       # :103
        lx, fx, lu, fu = o_lb["logits"], o_lb["feat"], o_u["logits"], o_u["feat"]
        sup_loss = H.ce_loss_mean(lx, y_lb)                                            # :116
        probs = H.softmax_probs(lu.detach())
        mask0 = torch.from_numpy(H.fixed_threshold_mask(probs.numpy(), self.p_cutoff))  # :119
        pl0 = torch.from_numpy(H.pseudo_label_hard(probs.numpy()))                     # :122-124
        tr["passes"] = [dict(mask=mask0.clone(), pseudo_label=pl0.clone())]
        K = 0
        if it > self.start_timing:                                                     # :126
            K = H.sr_decay(self.num_train_iter, it)
            for k in range(1, K + 1):                                                  # :62
                ok = V.vit_forward(P, x_ulb_w, self.cfg, droppath=droppath[k])
                pk = H.softmax_probs(ok["logits"].detach())
                mk = torch.from_numpy(H.fixed_threshold_mask(pk.numpy(), self.p_cutoff))
                plk = torch.from_numpy(H.pseudo_label_hard(pk.numpy()))
                reward = S.rewarder_forward(self.R, ok["feat"].detach(), plk)          # :84
                mask2 = S.reward_mask2(reward)                                         # :85-86
                unsup_loss = H.consistency_loss(ok["logits"], plk, mk, mask2)          # :87
                tr["passes"].append(dict(mask=mk.clone(), pseudo_label=plk.clone(), reward=reward.detach().clone(), mask2=mask2.clone()))
        else:
            unsup_loss = H.consistency_loss(lu, pl0, mask0)                            # :130-132
        tr["K"] = K
        if it > 0:                                                                     # :135
            gen = S.generated_labels(self.G, fx.detach())
            if it >= self.start_timing:
                r = float(S.rewarder_forward(self.R, fu.detach(), pl0).mean())         # :149-150
                if r > self.max_reward:
                    self.max_reward = r
                if it % self.N_k == 0 and it > self.start_timing:                      # :154
                    self.max_reward = -float("inf")
                    gen2 = S.generated_labels(self.G, fu.detach())
                    tr.update(self._sr_update(fu.detach(), gen2, pl0))
                    tr["sr_stage"] = 2
            else:
                tr.update(self._sr_update(fx.detach(), gen, y_lb))                     # :175-191
                tr["sr_stage"] = 1
        warm = float(np.clip(it / (self.unsup_warm_up * self.num_train_iter), 0.0, 1.0))   # :194
        total = sup_loss + self.lambda_u * unsup_loss * warm                           # :195
        total.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
        tr.update(sup_loss=float(sup_loss.detach()), unsup_loss=float(unsup_loss.detach()), total_loss=float(total.detach()),
                  util_ratio=float(mask0.mean()), grads=grads, feat=dict(x_lb=fx.detach(), x_ulb_w=fu.detach()), unsup_warmup=warm)
        fac = O.cosine_warmup_factor(it, self.num_train_iter, self.num_warmup_iter)
        self.opt_step += 1
        for k in self.P:
            lr, wd = self.hp[k]
            O.adamw_step(self.P[k], grads[k], self.m[k], self.v[k], self.opt_step, lr * fac, wd)
        tr["lr_factor"] = fac
        self.it += 1
        return tr


class SRPseudoLabelWrnOracle(SRPseudoLabelOracle):
    """The same step on the classic_cv backbone (WideResNet, oracle/wrn_ref.py): model(x_lb) moves the BatchNorm running statistics
    (srpseudolabel.py:96), every model(x_ulb_w) runs under Bn_Controller.freeze_bn (:100-110, :65-76) -- batch statistics of its own
    rows, running statistics untouched; no DropPath; SGD + Nesterov with the two weight-decay groups (core/utils/build.py:193-224)."""

    def __init__(self, wcfg, wrn_params, buffers, rewarder_params, generator_params, *, lr=0.03, momentum=0.9, weight_decay=1e-3, **k):
        from . import wrn_ref as W
        self.W = W
        cfg = type("C", (), dict(num_classes=wcfg.num_classes, depth=0))()
        k.update(lr=lr, weight_decay=weight_decay, layer_decay=1.0)
        self._wrn_init(cfg, wcfg, wrn_params, buffers, rewarder_params, generator_params, momentum, **k)

    def _wrn_init(self, cfg, wcfg, wrn_params, buffers, rp, gp, momentum, *, num_train_iter, start_timing, N_k=10, p_cutoff=0.95,
                  lambda_u=1.0, sr_lr=5e-4, lr=0.03, weight_decay=1e-3, layer_decay=1.0, num_warmup_iter=0, unsup_warm_up=0.4, **_):
        self.algorithm, self.cfg, self.wcfg = "srfixmatch", cfg, wcfg
        self.P = {k_: v.clone() for k_, v in wrn_params.items()}
        self.BUF = {k_: v.clone() for k_, v in buffers.items()}
        self.R = {k_: v.clone() for k_, v in rp.items()}
        self.G = {k_: v.clone() for k_, v in gp.items()}
        self.num_train_iter, self.start_timing, self.N_k = num_train_iter, start_timing, N_k
        self.p_cutoff, self.lambda_u, self.sr_lr, self.unsup_warm_up = p_cutoff, lambda_u, sr_lr, unsup_warm_up
        self.max_reward, self.it = -float("inf"), 0
        nwd = {n for n in self.P if "bn" in n or "bias" in n}                                   # wrn.py:143-148
        self.wd = {n: (0.0 if (v.ndim <= 1 or n.endswith(".bias") or n in nwd) else weight_decay) for n, v in self.P.items()}
        self.lr, self.momentum, self.num_warmup_iter = lr, momentum, num_warmup_iter
        self.mom = {k_: torch.zeros_like(v) for k_, v in self.P.items()}
        self.opt_step = 0
        self.rm = {k_: torch.zeros_like(v) for k_, v in self.R.items()}
        self.rv = {k_: torch.zeros_like(v) for k_, v in self.R.items()}
        self.r_step = 0

    def train_step(self, x_lb, y_lb, x_ulb_w):
        W, it = self.W, self.it
        tr = {}
        P = {k: v.detach().clone().requires_grad_(True) for k, v in self.P.items()}
        o_lb = W.wrn_forward(P, self.BUF, x_lb, self.wcfg, train=True, update_stats=True)              # :96
        o_u = W.wrn_forward(P, self.BUF, x_ulb_w, self.wcfg, train=True, update_stats=False)           # :100-110
        lx, fx, lu, fu = o_lb["logits"], o_lb["feat"], o_u["logits"], o_u["feat"]
        sup_loss = H.ce_loss_mean(lx, y_lb)
        probs = H.softmax_probs(lu.detach())
        mask0 = torch.from_numpy(H.fixed_threshold_mask(probs.numpy(), self.p_cutoff))
        pl0 = torch.from_numpy(H.pseudo_label_hard(probs.numpy()))
        tr["passes"] = [dict(mask=mask0.clone(), pseudo_label=pl0.clone())]
        K = 0
        if it > self.start_timing:
            K = H.sr_decay(self.num_train_iter, it)
            for k in range(1, K + 1):
                ok = W.wrn_forward(P, self.BUF, x_ulb_w, self.wcfg, train=True, update_stats=False)    # :65-76
                pk = H.softmax_probs(ok["logits"].detach())
                mk = torch.from_numpy(H.fixed_threshold_mask(pk.numpy(), self.p_cutoff))
                plk = torch.from_numpy(H.pseudo_label_hard(pk.numpy()))
                reward = S.rewarder_forward(self.R, ok["feat"].detach(), plk)
                mask2 = S.reward_mask2(reward)
                unsup_loss = H.consistency_loss(ok["logits"], plk, mk, mask2)
                tr["passes"].append(dict(mask=mk.clone(), pseudo_label=plk.clone(), reward=reward.detach().clone(), mask2=mask2.clone()))
        else:
            unsup_loss = H.consistency_loss(lu, pl0, mask0)
        tr["K"] = K
        if it > 0:
            gen = S.generated_labels(self.G, fx.detach())
            if it >= self.start_timing:
                r = float(S.rewarder_forward(self.R, fu.detach(), pl0).mean())
                if r > self.max_reward:
                    self.max_reward = r
                if it % self.N_k == 0 and it > self.start_timing:
                    self.max_reward = -float("inf")
                    gen2 = S.generated_labels(self.G, fu.detach())
                    tr.update(self._sr_update(fu.detach(), gen2, pl0))
                    tr["sr_stage"] = 2
            else:
                tr.update(self._sr_update(fx.detach(), gen, y_lb))
                tr["sr_stage"] = 1
        warm = float(np.clip(it / (self.unsup_warm_up * self.num_train_iter), 0.0, 1.0))
        total = sup_loss + self.lambda_u * unsup_loss * warm
        total.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
        tr.update(sup_loss=float(sup_loss.detach()), unsup_loss=float(unsup_loss.detach()), total_loss=float(total.detach()),
                  util_ratio=float(mask0.mean()), grads=grads, feat=dict(x_lb=fx.detach(), x_ulb_w=fu.detach()), unsup_warmup=warm)
        fac = O.cosine_warmup_factor(it, self.num_train_iter, self.num_warmup_iter)
        for k in self.P:
            W.sgd_nesterov_step(self.P[k], grads[k], self.mom[k], self.lr * fac, self.momentum, self.wd[k], first=self.opt_step == 0)
        self.opt_step += 1
        tr["lr_factor"] = fac
        self.it += 1
        return tr
