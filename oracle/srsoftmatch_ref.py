"""Oracle: one SRSoftMatch training step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates reference ``semilearn/algorithms/srsoftmatch/srsoftmatch.py`` (use_cat path): data_generator :62-96, train_step
:108-217.  Differences from the SRFlexMatch skeleton:
  * pass 0: softmax of the weak AND the labelled logits, DistAlignEMAHook on the weak probabilities (:137), the SoftMatch
    weight is taken on the ALIGNED probabilities (:140) while the pseudo label is the argmax of the raw logits (:143-148);
  * loop passes (:78-94): no distribution alignment -- weight and pseudo label from the plain softmax; the weighting hook's
    EMA state advances at EVERY call (1 + K times per step);
  * the mask is a weight in (0, 1], util_ratio its mean.
"""
import torch

from . import hooks_ref as H
from . import optim_ref as O
from . import semireward_ref as S
from .srflexmatch_ref import SRFlexMatchOracle


class SRSoftMatchOracle(SRFlexMatchOracle):
    def __init__(self, *a, ema_p=0.999, n_sigma=2, dist_uniform=True, **k):
        super().__init__(*a, **k)
        self.sm = H.SoftMatchState(self.cfg.num_classes, n_sigma, ema_p)
        self.da = H.DistAlignState(self.cfg.num_classes, ema_p, "uniform" if dist_uniform else "model")

    def train_step(self, x_lb, y_lb, x_ulb_w, x_ulb_s, droppath):
        it = self.it
        tr = {}
        P = {k: v.detach().clone().requires_grad_(True) for k, v in self.P.items()}
        lx, lw, ls, fx, fw, fs = self._forward(P, x_lb, x_ulb_w, x_ulb_s, droppath[0])
        sup_loss = H.ce_loss_mean(lx, y_lb)
        probs_lb = H.softmax_probs(lx.detach())                                        # :133
        probs = H.softmax_probs(lw.detach())                                           # :134
        aligned = self.da.dist_align(probs, probs_lb)                                  # :137
        mask0 = self.sm.masking(aligned)                                               # :140
        pl0 = torch.from_numpy(H.pseudo_label_hard(lw.detach().numpy()))               # :143-148 (argmax of the LOGITS)
        snap = lambda: dict(mu=float(self.sm.mu), var=float(self.sm.var), p_model=self.da.p_model.clone(), p_target=self.da.p_target.clone())   # noqa: E731
        tr["passes"] = [dict(mask=mask0.clone(), pseudo_label=pl0.clone(), **snap())]
        K = 0
        if it > self.start_timing:
            K = H.sr_decay(self.num_train_iter, it)
            for k in range(1, K + 1):
                _, lwk, lsk, _, fwk, _ = self._forward(P, x_lb, x_ulb_w, x_ulb_s, droppath[k])
                pk = H.softmax_probs(lwk.detach())                                     # :81
                plk = torch.from_numpy(H.pseudo_label_hard(pk.numpy()))                # :82-86
                mk = self.sm.masking(pk)                                               # :87
                reward = S.rewarder_forward(self.R, fwk.detach(), plk)
                mask2 = S.reward_mask2(reward)
                unsup_loss = H.consistency_loss(lsk, plk, mk, mask2)
                tr["passes"].append(dict(mask=mk.clone(), pseudo_label=plk.clone(), reward=reward.detach().clone(), mask2=mask2.clone(), **snap()))
        else:
            unsup_loss = H.consistency_loss(ls, pl0, mask0)
        tr["K"] = K
        if it > 0:
            gen = S.generated_labels(self.G, fx.detach())
            if it >= self.start_timing:
                r = float(S.rewarder_forward(self.R, fw.detach(), pl0).mean())
                if r > self.max_reward:
                    self.max_reward = r
                if it % self.N_k == 0 and it > self.start_timing:
                    self.max_reward = -float("inf")
                    gen2 = S.generated_labels(self.G, fw.detach())
                    tr.update(self._sr_update(fw.detach(), gen2, pl0))
                    tr["sr_stage"] = 2
            else:
                tr.update(self._sr_update(fx.detach(), gen, y_lb))
                tr["sr_stage"] = 1
        total = sup_loss + self.lambda_u * unsup_loss
        total.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
        tr.update(sup_loss=float(sup_loss.detach()), unsup_loss=float(unsup_loss.detach()), total_loss=float(total.detach()),
                  util_ratio=float(mask0.mean()), grads=grads, feat=dict(x_lb=fx.detach(), x_ulb_w=fw.detach(), x_ulb_s=fs.detach()))
        fac = O.cosine_warmup_factor(it, self.num_train_iter, self.num_warmup_iter)
        self.opt_step += 1
        for k in self.P:
            lr, wd = self.hp[k]
            O.adamw_step(self.P[k], grads[k], self.m[k], self.v[k], self.opt_step, lr * fac, wd)
        tr["lr_factor"] = fac
        self.it += 1
        return tr


class SRSoftMatchBertOracle(SRSoftMatchOracle):
    """usb_nlp flavour (BASELINE.json configs[3]; config/SemiReward/usb_nlp/softmatch/*.yaml): ClassificationBert backbone, dict batches
    ``(ids, mask)`` each padded to its own longest row, ``use_cat: False`` -- train_step forwards x_lb, x_ulb_s and (no_grad) x_ulb_w in
    separate model calls (srsoftmatch.py:118-128) and data_generator only x_ulb_s and x_ulb_w (:74-80); AdamW with layer_decay through
    ClassificationBert.group_matcher.  ``seeds``: dropout seed per model call or None (dropout off)."""

    def _hparams(self, cfg, lr, weight_decay, layer_decay):
        from . import bert_ref as BR
        return O.bert_param_hparams(BR.param_shapes(cfg), cfg.layers, lr, weight_decay, layer_decay)

    def _forward(self, P, x_lb, x_ulb_w, x_ulb_s, dp):
        from . import bert_ref as BR
        lx = fx = None
        if dp == "pass0":
            o = BR.bert_forward(P, x_lb[0], x_lb[1], self.cfg)
            lx, fx = o["logits"], o["feat"]
        os_ = BR.bert_forward(P, x_ulb_s[0], x_ulb_s[1], self.cfg)
        with torch.no_grad():
            ow = BR.bert_forward(P, x_ulb_w[0], x_ulb_w[1], self.cfg)
        return lx, ow["logits"], os_["logits"], fx, ow["feat"], os_["feat"]

    def train_step(self, x_lb, y_lb, x_ulb_w, x_ulb_s, droppath=None):
        K = H.sr_decay(self.num_train_iter, self.it) if self.it > self.start_timing else 0
        return super().train_step(x_lb, y_lb, x_ulb_w, x_ulb_s, ["pass0"] + ["loop"] * K)
