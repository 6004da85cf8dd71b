"""SRFreeMatch (FreeMatch + SemiReward) on the HIP engine -- registry key 'srfreematch'.

Reference: semilearn/algorithms/srfreematch/srfreematch.py (train_step :116-228, data_generator :76-109, entropy_loss :16-44).
Same batched (1+K)-pass step as SRFlexMatch with the FreeMatch self-adaptive thresholds (their EMA state advances at
every masking call, order dependent -> sequential launches) and the fairness term on the pass-0 strong logits, whose rows
therefore also carry a backward graph.  The reference's host sync ``if mask.sum() > 0`` (:216) is resolved on device.
"""
import torch

from .. import ops
from ..core.registry import ALGORITHMS
from .hooks import FreeMatchThresholdingHook, PseudoLabelingHook
from .srflexmatch import SRConsistencyBase
from .utils import SSL_Argument, str2bool


@ALGORITHMS.register("srfreematch")
class SRFreeMatch(SRConsistencyBase):
    fairness_rows = True

    def _init_thresholds(self, args):
        self.init(T=args.T, hard_label=args.hard_label, ema_p=args.ema_p, use_quantile=args.use_quantile, clip_thresh=args.clip_thresh)
        self.lambda_e = args.ent_loss_ratio

    def init(self, T, hard_label=True, ema_p=0.999, use_quantile=True, clip_thresh=False):
        self.T, self.use_hard_label, self.ema_p = T, hard_label, ema_p
        self.use_quantile, self.clip_thresh = use_quantile, clip_thresh

    def set_hooks(self):
        self.register_hook(PseudoLabelingHook(), "PseudoLabelingHook")
        self.register_hook(FreeMatchThresholdingHook(num_classes=self.num_classes, momentum=self.args.ema_p, device=self.device), "MaskingHook")
        super().set_hooks()

    def _masks(self, mp, mi, idx_ulb, P, nu, weak_logits=None):
        # the hook needs the full probability rows (class marginals): softmax of every pass in one launch, then the
        # order-dependent EMA updates pass by pass
        C = self.num_classes
        probs = torch.empty(P * nu, C, dtype=torch.float32, device=self.device)
        ops.row_max(weak_logits, False, probs, mp, mi, P * nu, C)
        hook = self.hooks_dict["MaskingHook"]
        return [hook.masking_from_probs(self, probs[k * nu:(k + 1) * nu], mp[k * nu:(k + 1) * nu], mi[k * nu:(k + 1) * nu]) for k in range(P)]

    def _fairness(self, logits_s0, mask0, dl_into):
        B, C = logits_s0.shape
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        ws = torch.empty(B * C, dtype=torch.float32, device=self.device)
        dl = dl_into if dl_into is not None else torch.empty(B, C, dtype=torch.float32, device=self.device)
        hook = self.hooks_dict["MaskingHook"]
        ops.freematch_entropy(logits_s0.contiguous(), mask0, hook.p_model, hook.label_hist, float(self.lambda_e), loss, dl, ws, B, C,
                              accumulate=dl_into is not None)
        return loss[0], dl

    def train_step(self, x_lb, y_lb, x_ulb_w, x_ulb_s):
        with self._step_scope():
            return self._train_step(x_lb, y_lb, None, x_ulb_w, x_ulb_s)

    def get_save_dict(self):
        d = super().get_save_dict()
        h = self.hooks_dict["MaskingHook"]
        d["p_model"], d["time_p"], d["label_hist"] = h.p_model.cpu(), h.time_p.cpu(), h.label_hist.cpu()
        return d

    def load_model(self, load_path):
        ck = super().load_model(load_path)
        h = self.hooks_dict["MaskingHook"]
        h.p_model.copy_(ck["p_model"]); h.time_p.copy_(ck["time_p"].reshape(1)); h.label_hist.copy_(ck["label_hist"])
        return ck

    @staticmethod
    def get_argument():
        return [SSL_Argument("--hard_label", str2bool, True), SSL_Argument("--T", float, 0.5), SSL_Argument("--p_cutoff", float, 0.95),
                SSL_Argument("--thresh_warmup", str2bool, True), SSL_Argument("--use_quantile", str2bool, False),
                SSL_Argument("--clip_thresh", str2bool, False), SSL_Argument("--start_timing", int, 20000),
                SSL_Argument("--feature_dim", int, 384), SSL_Argument("--sr_lr", float, 0.0005), SSL_Argument("--N_k", int, 10),
                SSL_Argument("--sr_ema", str2bool, True), SSL_Argument("--sr_ema_m", float, 0.999)]
