"""SRSoftMatch (SoftMatch + SemiReward) on the HIP engine -- registry key 'srsoftmatch'.

Reference: semilearn/algorithms/srsoftmatch/srsoftmatch.py (train_step :108-217, data_generator :62-96), hooks
srsoftmatch/utils.py:12-76 (SoftMatchWeightingHook) and hooks/dist_align.py:10-71 (DistAlignEMAHook).
Same batched (1+K)-pass step as SRFlexMatch.  Pass 0 weights the rows by the truncated Gaussian of the max of the
DISTRIBUTION-ALIGNED probabilities (:137-140) while the pseudo label stays the argmax of the raw logits (:143-148); the loop
passes use the plain softmax (:81-87).  The Gaussian's EMA mean / variance advance at every masking call (order dependent ->
sequential launches); the mask is a weight in (0, 1].
"""
import torch

from .. import ops
from ..core.registry import ALGORITHMS
from .hooks import DistAlignEMAHook, PseudoLabelingHook, SoftMatchWeightingHook
from .srflexmatch import SRConsistencyBase
from .utils import SSL_Argument, str2bool


@ALGORITHMS.register("srsoftmatch")
class SRSoftMatch(SRConsistencyBase):
    def _init_thresholds(self, args):
        self.init(T=args.T, hard_label=args.hard_label, dist_align=args.dist_align, dist_uniform=args.dist_uniform, ema_p=args.ema_p,
                  n_sigma=args.n_sigma, per_class=args.per_class)

    def init(self, T, hard_label=True, dist_align=True, dist_uniform=True, ema_p=0.999, n_sigma=2, per_class=False):
        self.T, self.use_hard_label, self.dist_align, self.dist_uniform = T, hard_label, dist_align, dist_uniform
        self.ema_p, self.n_sigma, self.per_class = ema_p, n_sigma, per_class

    def set_hooks(self):
        self.register_hook(PseudoLabelingHook(), "PseudoLabelingHook")
        self.register_hook(DistAlignEMAHook(num_classes=self.num_classes, momentum=self.args.ema_p,
                                            p_target_type="uniform" if self.args.dist_uniform else "model", device=self.device), "DistAlignHook")
        self.register_hook(SoftMatchWeightingHook(num_classes=self.num_classes, n_sigma=self.args.n_sigma, momentum=self.args.ema_p,
                                                  per_class=self.args.per_class, device=self.device), "MaskingHook")
        super().set_hooks()

    @property
    def masks_read_labelled_rows(self):
        return bool(self.hooks_dict["DistAlignHook"].update_p_target)

    def _masks(self, mp, mi, idx_ulb, P, nu, weak_logits=None):
        C = self.num_classes
        da, sm = self.hooks_dict["DistAlignHook"], self.hooks_dict["MaskingHook"]
        # pass 0: softmax of the weak (and, for the 'model' target, labelled) rows -> alignment -> weight of the ALIGNED max (:133-140)
        probs0 = torch.empty(nu, C, dtype=torch.float32, device=self.device)
        t_mp = torch.empty(nu, dtype=torch.float32, device=self.device)
        t_mi = torch.empty(nu, dtype=torch.int64, device=self.device)
        ops.row_max(weak_logits[:nu], False, probs0, t_mp, t_mi, nu, C)
        probs_lb = None
        if da.update_p_target:
            lb = self._lb_logits0.contiguous()
            nl = lb.shape[0]
            probs_lb = torch.empty(nl, C, dtype=torch.float32, device=self.device)
            ops.row_max(lb, False, probs_lb, torch.empty(nl, dtype=torch.float32, device=self.device),
                        torch.empty(nl, dtype=torch.int64, device=self.device), nl, C)
        _, amp, _ = da.align(self, probs0, probs_lb)
        masks = [sm.masking_from_max(self, amp)]
        # loop passes: plain softmax max-probs, one EMA update per pass, in order (:87)
        for k in range(1, P):
            masks.append(sm.masking_from_max(self, mp[k * nu:(k + 1) * nu]))
        return masks

    def train_step(self, x_lb, y_lb, x_ulb_w, x_ulb_s):
        with self._step_scope():
            return self._train_step(x_lb, y_lb, None, x_ulb_w, x_ulb_s)

    def get_save_dict(self):
        d = super().get_save_dict()
        da, sm = self.hooks_dict["DistAlignHook"], self.hooks_dict["MaskingHook"]
        d["p_model"], d["p_target"] = da.p_model.cpu(), da.p_target.cpu()
        d["prob_max_mu_t"], d["prob_max_var_t"] = sm.prob_max_mu_t.cpu(), sm.prob_max_var_t.cpu()
        d["dist_align_inited"] = da.inited.cpu()          # the reference keeps p_model = None until the first call
        return d

    def load_model(self, load_path):
        ck = super().load_model(load_path)
        da, sm = self.hooks_dict["DistAlignHook"], self.hooks_dict["MaskingHook"]
        da.p_model.copy_(ck["p_model"]); da.p_target.copy_(ck["p_target"])
        da.inited.copy_(ck.get("dist_align_inited", torch.ones(1, dtype=torch.int32)))
        sm.mu_var[0] = float(ck["prob_max_mu_t"]); sm.mu_var[1] = float(ck["prob_max_var_t"])
        return ck

    @staticmethod
    def get_argument():
        return [SSL_Argument("--hard_label", str2bool, True), SSL_Argument("--T", float, 0.5), SSL_Argument("--dist_align", str2bool, True),
                SSL_Argument("--dist_uniform", str2bool, True), SSL_Argument("--ema_p", float, 0.999), SSL_Argument("--n_sigma", int, 2),
                SSL_Argument("--per_class", str2bool, False), SSL_Argument("--start_timing", int, 20000),
                SSL_Argument("--feature_dim", int, 384), SSL_Argument("--sr_lr", float, 0.0005), SSL_Argument("--N_k", int, 10),
                SSL_Argument("--sr_ema", str2bool, True), SSL_Argument("--sr_ema_m", float, 0.999)]
