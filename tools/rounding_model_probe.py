"""Engine vs fp32 oracle vs the CPU model of the engine's rounding points (oracle.vit_ref.vit_forward_engine_rounding) on one forward of the
inference path (fused kernels): relative L2 of the logits / features, pairwise.  GPU box: python tools/rounding_model_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vit_ref as V
from oracle.gen_golden import trace_vit_params
from semireward_amd.nets import vit
from semireward_amd.utils import synth

rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())   # noqa: E731
for gain in (1.0, 24.0):
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    P = {k: torch.from_numpy(v) for k, v in trace_vit_params(cfg, 0, gain).items()}
    B = 72                                        # >= _FUSED_MLP_MIN_ROWS / 257 rows: the fused kernels of the big launches
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).standard_normal((B, 3, 32, 32)).astype(np.float32))
    for dps in (None, torch.from_numpy(synth.synth_droppath(11, V.drop_path_probs(cfg), B))):
        m = vit.vit_small_patch2_32(num_classes=100)
        m.load_state_dict(P)
        lg, ft, _ = m.forward_features(x.cuda(), None, dps.cuda() if dps is not None else None, save=False)
        torch.cuda.synchronize()
        with torch.no_grad():
            o32 = V.vit_forward(P, x, cfg, dps)
            oem = V.vit_forward_engine_rounding(P, x, cfg, dps)
        pe, p32, pm = (torch.softmax(t.float().cpu(), -1).max(-1).values for t in (lg, o32["logits"], oem["logits"]))
        print("gain %g droppath %s: logits rel engine-vs-fp32 %.2e, engine-vs-model %.2e, model-vs-fp32 %.2e | feat %.2e / %.2e / %.2e | max-prob abs dev %.3e / %.3e / %.3e" % (
            gain, dps is not None, rel(lg.cpu(), o32["logits"]), rel(lg.cpu(), oem["logits"]), rel(oem["logits"], o32["logits"]),
            rel(ft.cpu(), o32["feat"]), rel(ft.cpu(), oem["feat"]), rel(oem["feat"], o32["feat"]),
            float((pe - p32).abs().max()), float((pe - pm).abs().max()), float((pm - p32).abs().max())), flush=True)
