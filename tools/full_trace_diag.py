"""tests/test_gpu_srflexmatch.py::test_full_size_reference_trace, row by row: for every pass of both steps of tests/golden/srflexmatch_full_trace.npz
the reference's max-prob, the threshold it was compared with, the engine's max-prob, deviation and margin (GPU box)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import semireward_ref as S, vit_ref as V
from oracle.gen_golden import FULL, full_hook_state, trace_vit_params
from semireward_amd.algorithms import get_algorithm
from semireward_amd.nets import vit
from semireward_amd.utils import synth

fx = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/srflexmatch_full_trace.npz"
g = np.load(fx)
tr = FULL
C, Bl, Bu = tr["C"], tr["Bl"], tr["Bu"]
cfg = V.VitCfg(num_classes=C, **V.VIT_SMALL_P2_32)
T_ = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
P0 = trace_vit_params(cfg, tr["seed"], tr["head_gain"])
bseed = int(g["meta/bseed"])
b = synth.synth_batch(bseed, Bl, Bu, cfg.img_size, C, tr["ulb_dest_len"])
sel0, acc0 = full_hook_state(b["idx_ulb"])
for it in [int(i) for i in g["meta/its"]]:
    p = f"it{it}"
    K = int(g[f"{p}/K"])
    args = argparse.Namespace(algorithm="srflexmatch", num_classes=C, num_train_iter=tr["num_train_iter"], epoch=1, ema_m=0.0, ulb_loss_ratio=1.0,
                              use_cat=True, amp=False, lr=tr["lr"], weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=tr["num_warmup_iter"], optim="AdamW",
                              T=0.5, p_cutoff=tr["p_cutoff"], hard_label=True, thresh_warmup=True, ulb_dest_len=tr["ulb_dest_len"], N_k=tr["N_k"],
                              start_timing=tr["start_timing"], feature_dim=cfg.embed_dim, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99, gpu=0, rank=0,
                              world_size=1, distributed=False)
    alg = get_algorithm(args, vit.vit_small_patch2_32)
    alg.model.load_state_dict(T_(P0))
    alg.rewarder.load_state_dict(T_(synth.synth_params(S.rewarder_shapes(cfg.embed_dim, C), tr["seed"] + 1)))
    alg.generator.load_state_dict(T_(synth.synth_params(S.generator_shapes(cfg.embed_dim), tr["seed"] + 2)))
    h = alg.hooks_dict["MaskingHook"]
    h.selected_label = torch.from_numpy(sel0.copy())
    h.classwise_acc = torch.from_numpy(acc0.copy()).to("cuda")
    alg.it = it
    alg.optimizer.sched_step = it
    alg.inject_droppath = [torch.from_numpy(synth.synth_droppath(int(g[f"{p}/dp_seed0"]) + k, V.drop_path_probs(cfg), Bl + 2 * Bu)) for k in range(K + 1)]
    alg.trace = {}
    out, log = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))
    torch.cuda.synchronize()
    want = g[f"{p}/masks"]
    mpv = alg.trace["max_probs"].cpu().numpy().reshape(want.shape)
    refp, thr = g[f"{p}/mask_probs"], g[f"{p}/mask_thr"]
    devs = np.abs(mpv - refp)
    margin = np.minimum(np.minimum(np.abs(refp - thr), np.abs(refp - tr["p_cutoff"])), g[f"{p}/label_gap"])
    masks = np.stack([m.cpu().numpy() for m in alg.trace["masks"]])
    print("== %s K=%d  flips %d  pseudo-label mismatches %d  worst dev %.4f  smallest slack %.4f  rows with dev >= margin: %d of %d" % (
        p, K, int((masks != want).sum()), int((alg.trace["pseudo"].cpu().numpy().reshape(want.shape) != g[f"{p}/pseudo_label"]).sum()),
        float(devs.max()), float((margin - devs).min()), int((devs >= margin).sum()), devs.size))
    for k in range(K + 1):
        for r in range(Bu):
            flag = " <-- dev >= margin" if devs[k, r] >= margin[k, r] else ""
            print("  pass %d row %d  ref %.4f  thr %.4f  engine %.4f  dev %.4f  margin %.4f%s" % (k, r, refp[k, r], thr[k, r], mpv[k, r], devs[k, r], margin[k, r], flag))
    for k_ in ("sup_loss", "unsup_loss", "total_loss"):
        print("  %s engine %.5f reference %.5f" % (k_, float(log["train/" + k_]), float(g[f"{p}/log/{k_}"])))
    num = den = 0.0
    for nme, gv in alg.model.named_grads():
        st = int(g[f"{p}/grad/{nme}/stride"]); sm = g[f"{p}/grad/{nme}/sample"]
        a = gv.cpu().numpy().ravel()[::st].astype(np.float64)
        num += float(((a - sm) ** 2).sum()); den += float((sm.astype(np.float64) ** 2).sum())
    print("  step gradient rel-L2 (pooled samples) %.5f" % ((num / den) ** 0.5))
    if K:
        r = alg.trace["reward"].cpu().numpy().reshape(K, Bu)
        print("  reward max abs deviation %.5f" % float(np.abs(r - g[f"{p}/reward"]).max()))
