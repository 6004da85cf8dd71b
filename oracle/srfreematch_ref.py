"""Oracle: one SRFreeMatch training step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates reference ``semilearn/algorithms/srfreematch/srfreematch.py`` (use_cat path): data_generator :76-109,
train_step :116-228.  Same skeleton as SRFlexMatch with FreeMatchThresholdingHook (EMA state advanced at EVERY masking
call, i.e. 1+K times per step) and the extra fairness term  lambda_e * entropy_loss(mask0, logits_s(pass 0),
p_model, label_hist)  evaluated with the hook state AFTER all passes (:216-219)."""
import torch

from . import hooks_ref as H
from . import optim_ref as O
from . import semireward_ref as S
from .srflexmatch_ref import SRFlexMatchOracle


class SRFreeMatchOracle(SRFlexMatchOracle):
    def __init__(self, *a, ema_p=0.999, use_quantile=True, clip_thresh=False, lambda_e=1e-4, **k):
        super().__init__(*a, **k)
        self.fm = H.FreeMatchState(self.cfg.num_classes, ema_p, use_quantile, clip_thresh)
        self.lambda_e = lambda_e

    def train_step(self, x_lb, y_lb, x_ulb_w, x_ulb_s, droppath):
        it = self.it
        tr = {}
        P = {k: v.detach().clone().requires_grad_(True) for k, v in self.P.items()}
        lx, lw, ls, fx, fw, fs = self._forward(P, x_lb, x_ulb_w, x_ulb_s, droppath[0])
        sup_loss = H.ce_loss_mean(lx, y_lb)
        probs = H.softmax_probs(lw.detach())
        mask0 = self.fm.masking(probs)                                                 # :147 (softmax inside the hook)
        pl0 = torch.from_numpy(H.pseudo_label_hard(probs.numpy()))
        snap = lambda: dict(time_p=float(self.fm.time_p), p_model=self.fm.p_model.clone(), label_hist=self.fm.label_hist.clone())   # noqa: E731
        tr["passes"] = [dict(mask=mask0.clone(), pseudo_label=pl0.clone(), **snap())]
        K = 0
        if it > self.start_timing:
            K = H.sr_decay(self.num_train_iter, it)
            for k in range(1, K + 1):
                _, lwk, lsk, _, fwk, _ = self._forward(P, x_lb, x_ulb_w, x_ulb_s, droppath[k])
                pk = H.softmax_probs(lwk.detach())
                plk = torch.from_numpy(H.pseudo_label_hard(pk.numpy()))
                mk = self.fm.masking(pk)                                               # :102
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
        if float(mask0.sum()) > 0:                                                     # :216-219
            ent_loss = H.freematch_entropy_loss(mask0, ls, self.fm.p_model, self.fm.label_hist)
        else:
            ent_loss = torch.zeros(())
        total = sup_loss + self.lambda_u * unsup_loss + self.lambda_e * ent_loss       # :220
        total.backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in P.items()}
        tr.update(sup_loss=float(sup_loss.detach()), unsup_loss=float(unsup_loss.detach()), total_loss=float(total.detach()),
                  ent_loss=float(ent_loss.detach()), util_ratio=float(mask0.mean()), grads=grads,
                  feat=dict(x_lb=fx.detach(), x_ulb_w=fw.detach(), x_ulb_s=fs.detach()))
        fac = O.cosine_warmup_factor(it, self.num_train_iter, self.num_warmup_iter)
        self.opt_step += 1
        for k in self.P:
            lr, wd = self.hp[k]
            O.adamw_step(self.P[k], grads[k], self.m[k], self.v[k], self.opt_step, lr * fac, wd)
        tr["lr_factor"] = fac
        self.it += 1
        return tr


class SRFreeMatchW2vOracle(SRFreeMatchOracle):
    """usb_audio flavour (BASELINE.json configs[4]): Wav2Vec2 / HuBERT backbone on raw waveforms, ``use_cat: False`` -- train_step forwards x_lb,
    x_ulb_s and (no_grad) x_ulb_w in separate model calls (srfreematch.py:128-137), data_generator only x_ulb_s and x_ulb_w (:87-94); AdamW with
    layer decay through the backbone's group_matcher.  Train-mode randomness off (dropout / LayerDrop / SpecAugment probabilities 0)."""
    hubert = False

    def _hparams(self, cfg, lr, weight_decay, layer_decay):
        from . import w2v2_ref as WR
        return O.w2v_param_hparams(WR.param_shapes(cfg), cfg.layers, lr, weight_decay, layer_decay, hubert=self.hubert)

    def _forward(self, P, x_lb, x_ulb_w, x_ulb_s, dp):
        from . import w2v2_ref as WR
        lx = fx = None
        if dp == "pass0":
            o = WR.w2v_forward(P, x_lb, self.cfg)
            lx, fx = o["logits"], o["feat"]
        os_ = WR.w2v_forward(P, x_ulb_s, self.cfg)
        with torch.no_grad():
            ow = WR.w2v_forward(P, x_ulb_w, self.cfg)
        return lx, ow["logits"], os_["logits"], fx, ow["feat"], os_["feat"]

    def train_step(self, x_lb, y_lb, x_ulb_w, x_ulb_s, droppath=None):
        K = H.sr_decay(self.num_train_iter, self.it) if self.it > self.start_timing else 0
        return super().train_step(x_lb, y_lb, x_ulb_w, x_ulb_s, ["pass0"] + ["loop"] * K)
