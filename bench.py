#!/usr/bin/env python
"""bench.py -- unlabeled images/s of the SemiReward hot path (SRFlexMatch, ViT-S/2, CIFAR-100 shapes) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  When the process was NOT started by a launcher (no WORLD_SIZE in the environment) it starts the N
ranks itself, the way the reference does (train.py:344 ``mp.spawn(main_worker, nprocs=ngpus_per_node)``, :374-379 ``init_process_group``);
under ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`` it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.

A "step" = one reference-semantics training iteration in the steady SR regime (it > start_timing, K = sr_decay() = 8 extra backbone
passes, rewarder update every N_k = 10 steps): SRFlexMatch.train_step (batched (1+K)-pass forward, score filter, rewarder scoring,
masked losses, hand-written backward) + ParamUpdateHook (gradient all-reduce when N > 1, fused AdamW / scheduler / zero_grad).  Inputs
are synthetic (seeded N(0,1) images), resident in HBM before the timed region; weights are random-init of the reference
architecture.  Headline workload = BASELINE.json configs[1]: config/SemiReward/usb_cv/flexmatch/flexmatch_cifar100_200_0.yaml (per-GPU
batch 8 labelled + 8 weak + 8 strong).

Prints ONE JSON line on rank 0 (contract in the task statement).  ``value`` is the median of ``--repeats`` timed regions of exactly
``--steps`` steps, each bracketed by barrier + synchronize (all of them are listed in ``repeats``).  Besides ``roofline`` and
``cpu_baseline`` the line carries ``also``: short legs of the other workloads BASELINE.json's north_star names (ViT-S/16 on 224x224x3, a
scaled per-GPU batch, the pre-start_timing regime K = 0: each with its own ms_per_step and dominant-kernel roofline; the usb_nlp BERT-base
and usb_audio Wav2Vec2-base + FreeMatch steps of configs[3] / [4] under their own metric names).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NS = dict(algorithm="srflexmatch", num_classes=100, num_train_iter=204800, epoch=200, ema_m=0.0, ulb_loss_ratio=1.0,
          use_cat=True, amp=False, lr=5e-4, weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=5120, optim="AdamW",
          T=0.5, p_cutoff=0.95, hard_label=True, thresh_warmup=True, ulb_dest_len=50000, N_k=10, start_timing=20000,
          feature_dim=384, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99)
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (the 5 PF headline is 2:1 sparse)
HBM_PEAK_TBS = 8.0                        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 measured with a float4 copy)
START_IT = 30000                          # it >= 25601 -> sr_decay() == 8 (88 % of the reference run)
def _latest_traffic_json():
    """profiles/rNN_hbm_traffic.json of the latest round (PMC passes of this same command, tools/traffic.sh; static, labelled as such in the line)."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))
    return os.path.relpath(fs[-1], ROOT) if fs else os.path.join("profiles", "r05_hbm_traffic.json")


TRAFFIC_JSON = _latest_traffic_json()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; value = their median")
    ap.add_argument("--bu", type=int, default=8, help="per-GPU unlabeled batch (reference yaml: 8, uratio 1)")
    ap.add_argument("--bl", type=int, default=0, help="per-GPU labelled batch (default = --bu, uratio 1)")
    ap.add_argument("--regime", choices=["sr", "pre"], default="sr", help="sr: it > start_timing (K=8); pre: K=0")
    ap.add_argument("--img", type=int, choices=[32, 224], default=32,
                    help="32: ViT-S/2 on 32x32 (north-star config); 224: ViT-S/16 on 224x224 (vit_small_patch16_224, 197 tokens)")
    ap.add_argument("--net", choices=["vit", "bert", "hubert", "wave2vec", "wrn"], default="vit",
                    help="vit: the headline workload (BASELINE.json metric); bert: bert_base_uncased on [B, --seq-len] token batches (usb_nlp "
                         "shapes, configs[3]); wave2vec / hubert: wave2vecv2_base / hubert_base on [B, --samples] waveforms (usb_audio shapes, "
                         "configs[4]) -- reported under their own metric names")
    ap.add_argument("--alg", choices=["auto", "srflexmatch", "srfixmatch", "srsoftmatch", "srfreematch"], default="auto",
                    help="auto: srflexmatch (vit), srsoftmatch (bert, hubert), srfreematch (wave2vec = BASELINE.json configs[4])")
    ap.add_argument("--samples", type=int, default=64000, help="waveform length (usb_audio: 4 s at 16 kHz)")
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--elide-unread-rows", action="store_true",
                    help="NOT the reference's work (never the default, flagged in config): skip the (pass, image) rows whose outputs nothing reads")
    ap.add_argument("--force-dp", action="store_true",
                    help="N = 1: the data-parallel path with ONE rank on a 1-rank nccl (RCCL) communicator -- every collective of the step is issued "
                         "(gradient exchange selected by the start-up tuner, rccl_ranks: 1, allreduce_ms_per_step in the line); what a one-GPU box can "
                         "show of configs[2]'s backend")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-allreduce-ab", action="store_true", help="N > 1: skip the pinned legs of the gradient exchange (allreduce / overlap / rs_ag ...)")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary legs (224x224, scaled batch, K = 0 regime, BERT, Wav2Vec2)")
    ap.add_argument("--legs", default="", help="comma-separated subset of the secondary legs (default: all of them)")
    ap.add_argument("--extra-budget", type=float, default=900.0,
                    help="seconds the phases AFTER the headline (secondary legs, gradient-exchange legs, CPU baseline) may take before the line is "
                         "printed with what is complete")
    return ap.parse_args(argv)


# ---- self-launch (train.py:344 of the reference spawns its ranks the same way) -----------------------------------------------------
def launch_ranks(n):
    """Starts n copies of this command, one per GPU, with the torch.distributed environment variables; relays rank 0's stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SR_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs and rc == 0:
            for p in list(procs):
                c = p.poll()
                if c is not None:
                    procs.remove(p)
                    rc = rc or c
            time.sleep(0.05)
    finally:
        for p in procs:                                            # a rank failed: stop the ones we started (exact PIDs)
            p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def _thread_candidates():
    import torch
    top = torch.get_num_threads()
    return sorted({n for n in (8, 16, 32, 64, 128) if n <= top} | {top})


def cpu_baseline(bl, bu):
    """Reference-semantics step on the host cores, timed on a bounded sample with the CPU oracle (kind = 'port').  Protocol:
    ``torch.set_num_threads`` is swept over {8, 16, 32, 64, 128, all} with ONE (Bl+2Bu)-image ViT-S/2 forward each (a 6-GFLOP-per-layer
    problem is oversubscribed by 128 threads); at the best count three graph forwards (median), two one-graph backwards and one AdamW
    sweep; the K = 8 step time is (1+K)*t_fwd + 2*t_bwd + t_opt, exactly the reference's work per step (two graphs are back-propagated).
    The all-threads number is reported beside it (its forward scaled into the same formula)."""
    import numpy as np
    import torch
    from oracle import hooks_ref as H
    from oracle import optim_ref as O
    from oracle import vit_ref as V
    from semireward_amd.utils import synth
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.synth_params(V.param_shapes(cfg), 0).items()}
    b = synth.synth_batch(0, bl, bu, 32, 100, 50000)
    x = torch.from_numpy(np.concatenate([b["x_lb"], b["x_ulb_w"], b["x_ulb_s"]]))
    y = torch.from_numpy(np.concatenate([b["y_lb"], b["y_lb"][:1].repeat(2 * bu)]))
    dp = torch.from_numpy(synth.synth_droppath(1, V.drop_path_probs(cfg), x.shape[0]))
    all_threads = torch.get_num_threads()
    V.vit_forward(P, x[:2], cfg, dp[:, :, :2], aten_ops=True)        # warm the allocator / threads
    sweep = {}
    for n in _thread_candidates():
        torch.set_num_threads(n)
        with torch.no_grad():
            V.vit_forward(P, x[:2], cfg, dp[:, :, :2], aten_ops=True)    # the pool at its new size
        t0 = time.perf_counter()
        with torch.no_grad():                                         # the sweep only ranks thread counts: no graph, the allocator reuses its blocks
            V.vit_forward(P, x, cfg, dp, aten_ops=True)               # LayerNorm / GELU as the ATen kernels the reference's modules call
        sweep[n] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    tf, tb, outs = [], [], []
    for _ in range(3):
        t0 = time.perf_counter()
        outs.append(V.vit_forward(P, x, cfg, dp, aten_ops=True))
        tf.append(time.perf_counter() - t0)
    for o in outs[:2]:
        t0 = time.perf_counter()
        H.ce_loss_mean(o["logits"], y).backward()
        tb.append(time.perf_counter() - t0)
    del outs
    t0 = time.perf_counter()
    hp = O.vit_param_hparams(V.param_shapes(cfg), cfg.depth, 5e-4, 5e-4, 0.5)
    with torch.no_grad():
        for k, p in P.items():
            O.adamw_step(p, p.grad, torch.zeros_like(p), torch.zeros_like(p), 1, hp[k][0], hp[k][1])
    t_opt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    t_fwd, t_bwd = sorted(tf)[1], 0.5 * (tb[0] + tb[1])
    K = 8
    t_step = (1 + K) * t_fwd + 2 * t_bwd + t_opt
    t_step_all = t_step * sweep[all_threads] / sweep[best]           # same formula with the all-threads forward (backward scaled alike)
    return {"value": bu / t_step, "unit": "unlabeled images/s", "cores": best, "kind": "port",
            "sample": "oracle ViT-S/2 fp32 (ATen LayerNorm / GELU as in the reference's modules), Bt=%d, %d threads (best of the sweep): 3 graph forwards (%s s, median %.2f) + 2 one-graph backwards (%s s, mean %.2f) + AdamW (%.2fs); "
                      "K=8 step = 9*fwd + 2*bwd + opt = %.1fs" % (x.shape[0], best, "/".join("%.2f" % t for t in tf), t_fwd,
                                                                 "/".join("%.2f" % t for t in tb), t_bwd, t_opt, t_step),
            "spread_pct": 100.0 * (max(tf) - min(tf)) / t_fwd,
            "thread_sweep_fwd_s": {str(n): round(t, 3) for n, t in sweep.items()},
            "all_threads": {"cores": all_threads, "value": bu / t_step_all, "note": "one forward at all host threads, scaled into the same step formula"}}


def cpu_baseline_wrn(bl, bu):
    """BASELINE.json configs[0] ("CPU reference"): SRPseudoLabel on WRN-28-2 at 64 / 64, oracle port (oracle/wrn_ref.py, fp32 ATen convolutions),
    thread count swept like cpu_baseline.  The reference's step (srpseudolabel.py:92-201, K = 8): model(x_lb) with graph + (1 + K) x
    model(x_ulb_w) with graph + one backward through the labelled graph and the last unlabelled graph + SGD."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import wrn_ref as W
    from oracle.gen_golden import synth_wrn_params
    from semireward_amd.utils import synth
    cfg = W.WrnCfg(num_classes=100, **W.WRN_28_2)
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth_wrn_params(cfg, 0).items()}
    BUF = W.init_buffers(cfg)
    b = synth.synth_batch(0, bl, bu, 32, 100, 50000)
    xl, xu, y = torch.from_numpy(b["x_lb"]), torch.from_numpy(b["x_ulb_w"]), torch.from_numpy(b["y_lb"])
    all_threads = torch.get_num_threads()
    sweep = {}
    for n in _thread_candidates():
        torch.set_num_threads(n)
        with torch.no_grad():
            W.wrn_forward(P, BUF, xu[:4], cfg, train=True, update_stats=False)
            W.wrn_forward(P, BUF, xu, cfg, train=True, update_stats=False)       # (first call at this size: oneDNN primitives are built)
            t0 = time.perf_counter()
            W.wrn_forward(P, BUF, xu, cfg, train=True, update_stats=False)
            sweep[n] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    tf, tb = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        ol = W.wrn_forward(P, BUF, xl, cfg, train=True, update_stats=True)
        ou = W.wrn_forward(P, BUF, xu, cfg, train=True, update_stats=False)
        tf.append(0.5 * (time.perf_counter() - t0))
        t0 = time.perf_counter()
        (F.cross_entropy(ol["logits"], y) + F.cross_entropy(ou["logits"], y)).backward()
        tb.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    with torch.no_grad():
        for k, p in P.items():
            if p.grad is not None:                                   # (a bn1 whose output feeds nothing has no gradient, as in the reference)
                W.sgd_nesterov_step(p, p.grad, torch.zeros_like(p), 0.03, 0.9, 1e-3, True)
    t_opt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    K = 8
    t_fwd, t_bwd = sorted(tf)[1], sorted(tb)[1]
    t_step = (2 + K) * t_fwd + t_bwd + t_opt
    return {"value": bu / t_step, "unit": "unlabeled images/s", "cores": best, "kind": "port",
            "sample": "oracle WRN-28-2 fp32, %d / %d images, %d threads (best of the sweep): 3 graph forwards (%s s, median %.3f), 3 backwards of (labelled + last unlabelled) graph (%s s, median), SGD %.3fs; "
                      "K=8 step = 10*fwd + bwd + opt = %.2fs" % (bl, bu, best, "/".join("%.3f" % t for t in tf), t_fwd, "/".join("%.3f" % t for t in tb),
                                                                 t_opt, t_step),
            "spread_pct": 100.0 * (max(tf) - min(tf)) / t_fwd, "thread_sweep_fwd_s": {str(n): round(t, 3) for n, t in sweep.items()}}


class Leg:
    """One workload of the bench: builds the algorithm object and its resident batch, runs steps."""

    def __init__(self, a, ctx, net="vit", img=32, bu=8, bl=0, regime="sr", alg="auto", elide=False):
        import torch
        from semireward_amd.algorithms import get_algorithm
        from semireward_amd.utils import synth
        self.a, self.ctx, self.net, self.img, self.bu, self.bl, self.regime = a, ctx, net, img, bu, bl or bu, regime
        world, rank, local = ctx["world"], ctx["rank"], ctx["local"]
        self.dp_on = world > 1 or bool(ctx.get("force_dp"))
        common = dict(gpu=local, rank=rank, world_size=world, distributed=self.dp_on, force_dp=bool(ctx.get("force_dp")))
        self.elide = elide
        if net == "vit":
            from semireward_amd.nets import vit
            self.alg_name = "srflexmatch" if alg == "auto" else alg
            args = argparse.Namespace(**common, **dict(NS, algorithm=self.alg_name))
            m = get_algorithm(args, vit.vit_small_patch2_32 if img == 32 else vit.vit_small_patch16_224)
            P = synth.synth_params(m.model.names_shapes, 0)
            m.model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
            m.dp.broadcast_params(m.model, m.rewarder, m.generator)
            b = synth.synth_batch(100 + rank, self.bl, bu, img, 100, 50000)          # each rank: its own shard of the unlabeled stream
            self.batch = m.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
            start = START_IT
            self.metric, self.unit = "unlabeled images/sec/node (FlexMatch+SR, ViT-S CIFAR-100)", "unlabeled images/s"
            self.workload = ("SRFlexMatch ViT-S/2@32 CIFAR-100 shapes, flexmatch_cifar100_200_0.yaml, " if img == 32 else
                             "SRFlexMatch ViT-S/16@224 (vit_small_patch16_224) 224x224x3 batches, 100 classes, ")
        elif net == "wrn":
            # BASELINE.json configs[0]: configs/classic_cv_srpseudolabel_cifar100_400_wrn_28_2.yaml (authored: the reference's classic_cv pseudolabel
            # yaml + the SR keys) through the yaml loader: WRN-28-2, batch 64 / uratio 1, SGD-Nesterov lr 0.03 wd 1e-3, ema_m 0.999
            from semireward_amd import config as srconfig
            from semireward_amd.nets import wrn
            self.alg_name = "srpseudolabel"
            ya = srconfig.get_config(os.path.join(ROOT, "configs", "classic_cv_srpseudolabel_cifar100_400_wrn_28_2.yaml"),
                                     overrides=dict(common, ulb_dest_len=50000, num_warmup_iter=0))
            assert ya.net == "wrn_28_2" and ya.algorithm == "srpseudolabel" and ya.batch_size == 64 and ya.num_classes == 100
            m = get_algorithm(ya, wrn.wrn_28_2)          # reference init (wrn.py:108-117), seed 0
            m.dp.broadcast_params(m.model, m.rewarder, m.generator)
            b = synth.synth_batch(100 + rank, self.bl, bu, 32, 100, 50000)
            self.batch = m.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
            start = 200001                                               # > start_timing (100000); sr_decay(): max(8, 1 + 2^20 / it) = 8
            self.metric, self.unit = "unlabeled images/sec/node (PseudoLabel+SR, WRN-28-2 CIFAR-100)", "unlabeled images/s"
            self.workload = "SRPseudoLabel WRN-28-2@32 CIFAR-100 shapes, configs/classic_cv_srpseudolabel_cifar100_400_wrn_28_2.yaml, SGD, "
        else:
            # BASELINE.json configs[3] / [4] through their authored yamls (configs/usb_nlp_srsoftmatch_aclImdb_20_bert_base.yaml: IMDB, 2 classes;
            # configs/usb_audio_srfreematch_urbansound8k_100_wave2vecv2_base.yaml); the HuBERT leg (not a BASELINE config; the net every usb_audio SR
            # yaml of the reference names) keeps the keys of config/SemiReward/usb_audio/softmatch/softmatch_urbansound8k_100_0.yaml as a dict
            from semireward_amd import config as srconfig
            ypath = None
            if net == "bert":
                from semireward_amd.nets import bert
                builder, ypath = bert.bert_base_uncased, "usb_nlp_srsoftmatch_aclImdb_20_bert_base.yaml"
            elif net == "hubert":
                from semireward_amd.nets import hubert
                builder, C, ld = hubert.hubert_base, 10, 0.75
            else:
                from semireward_amd.nets import wave2vec
                builder, ypath = wave2vec.wave2vecv2_base, "usb_audio_srfreematch_urbansound8k_100_wave2vecv2_base.yaml"
            if ypath is not None and alg in ("auto", srconfig.load_yaml(os.path.join(ROOT, "configs", ypath))["algorithm"]):
                args = srconfig.get_config(os.path.join(ROOT, "configs", ypath), overrides=common)
                self.alg_name, C = args.algorithm, args.num_classes
                self.config_file = "configs/" + ypath
            else:
                if ypath is not None:
                    C, ld = (2, 0.75) if net == "bert" else (10, 0.75)
                self.alg_name = "srsoftmatch" if alg == "auto" else alg
                extra = dict(dist_align=True, dist_uniform=True, ema_p=0.999, n_sigma=2, per_class=False) if self.alg_name == "srsoftmatch" else \
                    dict(ema_p=0.999, use_quantile=False, clip_thresh=False, ent_loss_ratio=0.001) if self.alg_name == "srfreematch" else {}
                cfg = dict(NS, algorithm=self.alg_name, num_classes=C, num_train_iter=102400, start_timing=10000, lr=5e-5, layer_decay=ld,
                           use_cat=False, feature_dim=768, **extra)
                args = argparse.Namespace(**common, **cfg)
            m = get_algorithm(args, builder)                              # random init (no network for the checkpoint)
            m.dp.broadcast_params(m.model, m.rewarder, m.generator)
            g = torch.Generator().manual_seed(100 + rank)
            if net == "bert":
                mk = lambda n: {"input_ids": torch.randint(1, 30522, (n, a.seq_len), generator=g),       # noqa: E731
                                "attention_mask": torch.ones(n, a.seq_len, dtype=torch.int64)}          # full-length rows
                self.metric = "unlabeled sequences/sec/node (%s, BERT-base, L=%d)" % (self.alg_name, a.seq_len)
                self.unit = "unlabeled sequences/s"
                self.workload = "%s bert_base_uncased, [B, %d] token batches, %d classes (%s), use_cat False, " % (
                    self.alg_name, a.seq_len, C, getattr(self, "config_file", "usb_nlp SR yaml shapes"))
            else:
                mk = lambda n: torch.randn(n, a.samples, generator=g)   # noqa: E731
                nm = "HuBERT-base" if net == "hubert" else "Wav2Vec2-base"
                self.metric = "unlabeled clips/sec/node (%s, %s, %d samples)" % (self.alg_name, nm, a.samples)
                self.unit = "unlabeled clips/s"
                self.workload = "%s %s, [B, %d] waveforms, %d classes (%s), use_cat False, " % (
                    self.alg_name, "hubert_base" if net == "hubert" else "wave2vecv2_base", a.samples, C, getattr(self, "config_file", "usb_audio SR yaml shapes"))
            kw = dict(x_lb=mk(self.bl), y_lb=torch.randint(0, C, (self.bl,), generator=g), x_ulb_w=mk(bu), x_ulb_s=mk(bu))
            self.batch = m.process_batch(**kw)
            start = 90001                                                 # sr_decay(): max(8, 1 + 102400 / it) = 8
        m.model.seed = 1234 + rank
        m.elide_unread_rows = bool(elide)
        m.it = start if regime == "sr" else 1000
        m.optimizer.sched_step = m.it
        m.model.train()
        self.alg = m
        # HIP-graph replay of the step (core/stepgraph.py; algorithms whose step keeps no Python-side state, single rank).  SR_HIP_GRAPH = auto
        # (default): decided from a measurement in run() -- replay when the host needs more than HOST_BOUND of a step's time to enqueue it (a
        # slow or oversubscribed host); on one MI355X with its own host the step is GPU-bound (1.9 ms of enqueue per 4.9 ms step, replay
        # 5.32 vs 5.23 ms eager) and eager launches stay.  1 / 0 force it.
        self.graph, self.graph_mode = None, os.environ.get("SR_HIP_GRAPH", "auto")
        # (data parallel: train_step is replayed, the gradient exchange + the one optimizer launch follow eagerly -- core/stepgraph.py _split)
        self.graph_ok = getattr(m, "graph_safe", False) and net == "vit"
        if self.graph_mode not in ("auto", "0") and self.graph_ok:
            self._enable_graph()
        self.workload += "steady SR regime" if regime == "sr" else "pre-start_timing regime"

    HOST_BOUND = 0.85

    def _enable_graph(self):
        from semireward_amd.core.stepgraph import StepGraph
        self.graph = StepGraph(self.alg, warm=1)

    def host_enqueue_ms(self, steps=6):
        """Host time to ENQUEUE a step (perf_counter around step() with the device drained before and nothing waited for inside) and the step's
        device time (events), over ``steps`` steps.  Returns (host ms per step, device ms per step)."""
        import torch
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        host = 1e3 * (time.perf_counter() - t0) / steps
        e1.record()
        e1.synchronize()
        return host, e0.elapsed_time(e1) / steps

    def step(self):
        m = self.alg
        if self.graph is not None:         # train_step + ParamUpdateHook as ONE captured HIP graph per step variant (core/stepgraph.py)
            self.graph.step(**self.batch)
        else:
            m.out_dict, m.log_dict = m.train_step(**self.batch)
            m.call_hook("after_train_step")
        m.it += 1

    def fence(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.ctx["world"] > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, steps):
        """Exactly ``steps`` steps between two fences; max over ranks.  Returns seconds."""
        import torch
        import torch.distributed as dist
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        dt = time.perf_counter() - t0
        if self.ctx["world"] > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    def run(self, steps, warmup, repeats, roofline=True):
        import torch
        from semireward_amd import ops
        a, world = self.a, self.ctx["world"]
        m = self.alg
        # set-up, untimed like building the model: the start-up autotune of the step schedule (share of the inference rows on the second
        # stream, srflexmatch._DeferTuner) runs its candidates as real training steps; the W warm-up and the K timed steps follow it
        # (data parallel: the gradient exchange is selected first -- distributed.ExchangeTuner -- then the step schedule)
        tune_steps = 0
        tuning = lambda: (not m.dp.settled) or bool(getattr(m, "_tuners", None)) or bool(getattr(m, "_untuned", None))   # noqa: E731
        while (getattr(m, "_tuners", None) is not None or not m.dp.settled) and (tune_steps == 0 or tuning()) and tune_steps < 96:
            self.step()
            tune_steps += 1
        # host-boundness of the step, every rank its own: enqueue time against device time of the same steps
        host_ms, dev_ms = self.host_enqueue_ms()
        host_bound = host_ms > self.HOST_BOUND * dev_ms
        if self.graph is None and self.graph_mode == "auto" and self.graph_ok and host_bound:
            self._enable_graph()
        # ... and the capture of the step's HIP graphs: every variant that occurs in the steady regime (with / without the SemiReward update of
        # every N_k-th step) is run eagerly once and captured at its next occurrence
        cap_steps = 0
        # (data parallel: the variant with the rewarder update -- a collective inside train_step -- stays eager, one variant is captured)
        while self.graph is not None and len(self.graph.graphs) < (2 if (self.regime == "sr" and not self.dp_on) else 1) and cap_steps < 3 * NS["N_k"]:
            self.step()
            cap_steps += 1
        for _ in range(warmup):
            self.step()
        m.dp.comm_events = [] if self.dp_on else None          # event pairs around the gradient all-reduce of every timed step
        syncs0 = m.dp.agreement_syncs
        dts = [self.timed(steps) for _ in range(repeats)]
        syncs_timed = m.dp.agreement_syncs - syncs0
        ar_ms = None
        if self.dp_on:
            torch.cuda.synchronize()
            ev = m.dp.comm_events
            m.dp.comm_events = None
            if ev:
                ar_ms = sum(x.elapsed_time(y) for x, y in ev) / len(ev)
        dt = sorted(dts)[len(dts) // 2]
        K = m.sr_decay() if self.regime == "sr" else 0
        bl, bu = self.bl, self.bu
        out = {"metric": self.metric, "value": world * bu * steps / dt, "unit": self.unit, "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic",
               "repeats": {"n": len(dts), "ms_per_step": [round(1e3 * d / steps, 4) for d in dts], "value_is": "median",
                           "spread_pct": round(100.0 * (max(dts) - min(dts)) / dt, 2)},
               # `config` is FLAT (scalars and strings only): the driver's BENCH_rNN.json keeps those and drops nested values
               "config": {"workload": self.workload, "algorithm": self.alg_name, "bl": bl, "bu_w": bu, "bu_s": 0 if self.net == "wrn" else bu,
                          "K_passes": K,
                          "forward_image_passes_per_step": (bl + (1 + K) * bu) if self.net == "wrn" else
                          (1 + K) * (bl + 2 * bu) - (K * bl if self.net != "vit" else 0) -
                          ((K * bl if self.net == "vit" else 0) + max(K - 1, 0) * bu if self.elide else 0),
                          "unread_rows": "ELIDED (opt-in extension: fewer forward rows than the reference executes, results identical)"
                          if self.elide else "computed, as in the reference",
                          "backward_images_per_step": bl + bu, "rewarder_update_every": NS["N_k"], "parallelism": "dp%d" % world,
                          "grad_allreduce": "flat fp32 block, exchange selected at start-up (grad_exchange)" if self.dp_on else "none"}}
        if os.environ.get("SR_BENCH_HEAD"):                   # (tools/round_evidence.sh: the code commit the line was measured on)
            out["head"] = os.environ["SR_BENCH_HEAD"]
        rep = list(getattr(m, "defer_report", {}).values())
        if rep:
            out["config"].update(deferred_share=rep[-1]["chosen"], deferred_images=rep[-1]["deferred_images"], autotune_steps=tune_steps)
            out["step_schedule"] = {"deferred_share": rep[-1]["chosen"], "deferred_images": rep[-1]["deferred_images"],
                                    "autotune_ms_per_step_by_share": rep[-1]["ms_per_step"], "autotune_steps": tune_steps}
        out["config"]["launch"] = ("HIP graph replay of train_step + optimizer (%d variants captured; %d replays, %d eager steps incl. tuning / capture)" % (
            len(self.graph.graphs), self.graph.replays, self.graph.eager_steps)) if self.graph is not None else "eager launches"
        # host enqueue time of a step per rank (8 feeder processes share one host on a node) and whether the step is host-bound there
        hosts = [host_ms]
        if world > 1:
            import torch.distributed as dist
            t = torch.zeros(world, dtype=torch.float64, device="cuda")
            t[self.ctx["rank"]] = host_ms
            dist.all_reduce(t)
            hosts = [float(x) for x in t.cpu()]
        out["host_enqueue_ms_per_step"] = {"per_rank": [round(h, 4) for h in hosts], "max": max(hosts), "device_ms_per_step_this_rank": dev_ms,
                                           "host_bound": bool(max(hosts) > self.HOST_BOUND * dev_ms),
                                           "measured_over": "6 steps after the schedule tuning, device drained before, nothing waited for inside"}
        out["config"]["host_enqueue_ms_per_step_max"] = max(hosts)
        out["config"]["hip_graph"] = "%s -> %s" % (self.graph_mode, "replay" if self.graph is not None else "eager")
        if self.dp_on:
            out["config"]["backend"] = self.ctx["backend_note"]
            out["rccl_ranks"] = self.ctx["rccl_ranks"]
            out["devices"] = self.ctx["ndev_used"]
            out["allreduce_ms_per_step"] = ar_ms
            # which gradient exchange ran: chosen at start-up from measurements on this backend (or pinned by SR_GRAD_EXCHANGE); rank agreements
            # (blocking) happen while the schedules are tuned -- none inside the timed regions
            out["grad_exchange"] = dict(m.dp.exchange_report or {}, running=m.dp.exchange)
            out["config"]["grad_exchange"] = m.dp.exchange
            out["rank_agreement_syncs"] = {"during_tuning": syncs0 + (m.dp.exchange_report or {}).get("agreement_syncs", 0),
                                           "in_timed_regions": syncs_timed}
        if roofline:
            # roofline pass: the SAME steps again, in this process, with HIP events (on the launch stream) around every GEMM-class launch.
            # Kept out of the timed regions because 2 x 145 event records per step cost ~0.9 ms of host time.  Every rank runs the pass
            # (the steps contain collectives); rank 0 reports.
            prof = ops.enable_gemm_profile()
            replay, self.graph = self.graph, None          # (the pass instruments single launches: eager steps, also when the leg replays a graph)
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            self.graph = replay
            ops.disable_gemm_profile()
            if self.ctx["rank"] == 0:
                out["roofline"] = self.roofline(prof, steps)
        return out

    def roofline(self, prof, prof_steps):
        """Roofline object of the dominant kernel.  Time = the kernel's own execution time per launch, measured live with an event pair bound
        to each dispatch (srhip_prof_*, csrc/prof.hip) -- the quantity `rocprofv3 --kernel-trace --stats` averages for the same kernel
        (tools/round_evidence.sh asserts the two agree within 3 %).  The torch event pair recorded AROUND the call on the launch stream
        (execution + dispatch latency + event packets) is reported beside it and never enters a fraction."""
        pk = prof.per_kernel()
        name, (fl, ms, n, nbytes, ms_pair) = max(pk.items(), key=lambda kv: kv[1][1])   # dominant kernel = most execution time in the pass
        tfl, tms, tn = sum(v[0] for v in pk.values()), sum(v[1] for v in pk.values()), sum(v[2] for v in pk.values())
        # which roof bounds this kernel?  arithmetic intensity of its launches vs the ridge of the machine
        ridge = MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12 / (HBM_PEAK_TBS * 1e12)
        intensity = fl / nbytes
        traffic, traffic_source = None, None
        tf = os.path.join(ROOT, TRAFFIC_JSON)
        headline = self.bu == 8 and self.regime == "sr" and self.ctx["world"] == 1 and self.img == 32 and self.net == "vit" and not self.elide
        if os.path.exists(tf) and headline:
            traffic = (json.load(open(tf)).get(name.split("<")[0]) or json.load(open(tf)).get(name) or {}).get("hbm_bytes_per_launch")
            traffic_source = "%s (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/traffic.sh; not measured in this run)" % TRAFFIC_JSON

        def against_roof(kfl, kms, kbytes):
            if kfl / kbytes < ridge:
                a_ = kbytes / (kms * 1e-3) / 1e9
                return {"bound": "hbm", "achieved": a_, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": a_ / (HBM_PEAK_TBS * 1e3)}
            a_ = kfl / (kms * 1e-3) / 1e12
            return {"bound": "mfma", "achieved": a_, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a_ / MFMA_BF16_DENSE_PEAK_TFLOPS}
        roof = against_roof(fl, ms, nbytes)
        roof.update({"kernel": name, "traffic": traffic, "traffic_source": traffic_source, "launches": n, "avg_launch_us": 1e3 * ms / n,
                     "avg_launch_us_is": "kernel execution time of the dispatch (start/stop events bound to the launch: what rocprofv3 --kernel-trace reports)",
                     "avg_launch_us_event_pair": 1e3 * ms_pair / n,
                     "algorithmic_bytes_per_launch": nbytes / n, "flop_per_launch": fl / n, "flop_per_byte": intensity,
                     "ridge_flop_per_byte": ridge, "tflops": fl / (ms * 1e-3) / 1e12, "ms_per_step_in_kernel": ms / prof_steps,
                     "measured_over": "%d instrumented steps run right after the timed regions (same process, same inputs)" % prof_steps,
                     "all_instrumented_kernels": {"tflops": tfl / (tms * 1e-3) / 1e12, "launches": tn, "ms_per_step": tms / prof_steps}})
        # the other instrumented kernels of the step, same clock, each against the roof that bounds it: the row-streaming attention launch,
        # the GEMMs / attention / LayerNorm backward of the gradient rows, the grouped weight-gradient launch
        others = []
        for kn, (kfl, kms, kn_l, kbytes, _) in sorted(pk.items(), key=lambda kv: -kv[1][1]):
            if kn == name or kn_l == 0 or kbytes <= 0 or kms <= 0:
                continue
            o = against_roof(kfl, kms, kbytes)
            o.update({"kernel": kn, "launches": kn_l, "avg_launch_us": 1e3 * kms / kn_l, "ms_per_step_in_kernel": kms / prof_steps})
            others.append(o)
        roof["other_kernels"] = others[:9]
        return roof


def _order_config(out):
    """The driver's record keeps the first ~25 scalar keys of ``config``: the workload's identity first, then ONE summary string per leg (the three
    BASELINE.json configs lead), then the legs' numbers, then the rest."""
    cfg = out["config"]
    first = [k for k in ("workload", "algorithm", "bl", "bu_w", "bu_s", "K_passes", "parallelism") if k in cfg]
    summ = [k for k in cfg if k.startswith("leg_") and k.endswith("_summary")]
    nums = [k for k in cfg if k.startswith("leg_") and not k.endswith("_summary")]
    rest = [k for k in cfg if k not in first and not k.startswith("leg_")]
    out["config"] = {k: cfg[k] for k in first + summ + nums + rest}


def worker(a):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (start it as `python bench.py --gpus N`, or under torch.distributed.run "
                         "with --nproc-per-node equal to --gpus)" % (world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libsrhip has no CPU path)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("SR_DIST_BACKEND", "nccl")                                  # "nccl" is RCCL on ROCm
    if backend == "nccl" and ndev < world:
        # never a silent fall-back: N ranks over fewer devices would print a "scaling" line measured through host memory (a mis-set
        # HIP_VISIBLE_DEVICES on a real 8-GPU node).  The functional check with ranks sharing devices has to be asked for.
        raise SystemExit("bench.py: --gpus %d but only %d visible device(s).  One rank per GPU over RCCL needs %d devices (check HIP_VISIBLE_DEVICES / "
                         "ROCR_VISIBLE_DEVICES); for a functional check of the multi-rank path with ranks sharing devices set SR_DIST_BACKEND=gloo "
                         "explicitly (its line is labelled as such and is not a scaling measurement)." % (world, ndev, world))
    ctx = {"world": world, "rank": rank, "local": local % max(ndev, 1), "rccl_ranks": 0, "ndev_used": min(ndev, world), "backend_note": None,
           "force_dp": bool(a.force_dp) and world == 1}
    torch.cuda.set_device(ctx["local"])
    if ctx["force_dp"]:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    if world > 1 or ctx["force_dp"]:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", ctx["local"]))
            ctx["rccl_ranks"] = dist.get_world_size()
            ctx["backend_note"] = "nccl (RCCL), one rank per GPU" + (" -- ONE rank, data-parallel path forced (--force-dp)" if ctx["force_dp"] else "")
        else:
            # fewer devices than ranks (or SR_DIST_BACKEND=gloo): the ranks share devices and the collectives go through host memory --
            # a functional check of the N-rank path, NOT a scaling measurement
            dist.init_process_group(backend, rank=rank, world_size=world)
            ctx["backend_note"] = "%s: %d ranks on %d device(s) -- functional check of the multi-rank path, not a scaling measurement" % (
                backend, world, ctx["ndev_used"])
            if rank == 0:
                print("bench.py: %s" % ctx["backend_note"], file=sys.stderr)

    head = Leg(a, ctx, net=a.net, img=a.img, bu=a.bu, bl=a.bl, regime=a.regime, alg=a.alg, elide=a.elide_unread_rows)
    out = head.run(a.steps, a.warmup, max(1, a.repeats), roofline=not a.no_roofline)
    if rank == 0 and os.environ.get("SR_PHASES", "0") != "0":
        print("phases (ms since step start: gpu, host):", {k: (round(v[0], 2), round(v[1], 2)) for k, v in head.alg.phase_report().items()},
              file=sys.stderr)
    default_headline = a.net == "vit" and a.img == 32 and a.bu == 8 and a.bl in (0, 8) and a.regime == "sr" and a.alg in ("auto", "srflexmatch") \
        and not a.elide_unread_rows
    del head
    torch.cuda.empty_cache()
    # The headline is measured.  What follows (secondary legs, the gradient-exchange A/B for N > 1) must never cost the line: if those phases
    # do not finish within --extra-budget seconds (a collective one rank never reaches does not raise, it waits), every rank's watchdog
    # ends its process and rank 0 prints the line with what is complete.
    import threading
    budget = float(a.extra_budget)
    state = {"phase": "secondary legs", "done": False}

    def _give_up():
        if state["done"]:
            return
        if rank == 0:
            out["truncated"] = "watchdog: phase '%s' did not finish within %.0f s after the headline; the headline fields are complete" % (
                state["phase"], budget)
            print(json.dumps(out), flush=True)
        os._exit(0)
    watchdog = threading.Timer(budget, _give_up)
    watchdog.daemon = True
    watchdog.start()
    if default_headline and not a.no_also:
        # the other workloads north_star names, short legs in the same process: fewer steps, one timed region each
        also = out["also"] = []                          # (filled in place: a truncated line carries the legs that finished)
        # (BASELINE.json configs[3] and [4] -- the usb_nlp BERT-base and usb_audio Wav2Vec2-base + FreeMatch steps -- ride along with their own
        # metric names and fewer steps; a leg that fails is reported as such and never costs the line)
        # order: the BASELINE.json configs first (configs[0] WRN, [3] BERT, [4] Wav2Vec2), so that a line cut short by the watchdog -- and the flat
        # ``leg_*`` keys the driver's record keeps from ``config`` -- carry them before the north_star's secondary ViT workloads
        legs = (("classic_cv_wrn_28_2_srpseudolabel", dict(net="wrn", bu=64), max(4, a.steps // 2), True),
                ("usb_nlp_bert_base_srsoftmatch", dict(net="bert"), max(3, a.steps // 5), True),
                ("usb_audio_wave2vecv2_base_srfreematch", dict(net="wave2vec", alg="srfreematch"), max(3, a.steps // 5), True),
                ("vit_s16_224", dict(img=224), max(4, a.steps // 2), True), ("scaled_batch_bu64", dict(bu=64), max(4, a.steps // 2), True),
                ("pre_start_timing_K0", dict(regime="pre"), max(4, a.steps // 2), True))
        only = [t for t in a.legs.split(",") if t]        # (--legs: a subset of the secondary legs)
        for tag, kw, nsteps, roof in legs:
            if only and tag not in only:
                continue
            leg = None
            try:
                leg = Leg(a, ctx, **kw)
                o = leg.run(nsteps, 2, 3, roofline=roof and not a.no_roofline)
                keep = {k: o[k] for k in ("metric", "value", "unit", "ms_per_step", "repeats", "n_gpus") if k in o}
                keep.update(leg=tag, workload=o["config"]["workload"], K_passes=o["config"]["K_passes"],
                            per_gpu_batch={k: o["config"][k] for k in ("bl", "bu_w", "bu_s")})
                if "forward_image_passes_per_step" in o["config"]:
                    keep["forward_image_passes_per_step"] = o["config"]["forward_image_passes_per_step"]
                if "roofline" in o:
                    keep["roofline"] = {k: o["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "launches", "avg_launch_us",
                                                                      "flop_per_launch", "algorithmic_bytes_per_launch")}
                if "allreduce_ms_per_step" in o:
                    keep["allreduce_ms_per_step"] = o["allreduce_ms_per_step"]
                if "host_enqueue_ms_per_step" in o:       # a leg whose host cannot keep up shows here (round 6: a blocking table upload in the audio legs)
                    keep["host_enqueue_ms_per_step"] = round(o["host_enqueue_ms_per_step"]["max"], 3)
                    keep["host_bound"] = o["host_enqueue_ms_per_step"]["host_bound"]
                if tag.startswith("classic_cv") and rank == 0 and world == 1 and not a.no_cpu_baseline:
                    keep["cpu_baseline"] = cpu_baseline_wrn(64, 64)       # the configuration BASELINE.json labels "CPU reference"
                also.append(keep)
                # the driver keeps the scalar keys of `config`: every leg's value / ms / roofline rides there as flat keys
                rf = keep.get("roofline", {})
                out["config"].update({"leg_%s_summary" % tag: "%.1f %s, %.3f ms/step, dominant kernel %s at %s of the %s roof%s" % (
                                          keep["value"], keep["unit"], keep["ms_per_step"], rf.get("kernel"),
                                          ("%.3f" % rf["frac"]) if rf.get("frac") is not None else "n/a", rf.get("bound"),
                                          (", host enqueue %.2f ms/step%s" % (keep["host_enqueue_ms_per_step"], " (HOST-BOUND)" if keep.get("host_bound") else ""))
                                          if "host_enqueue_ms_per_step" in keep else ""),
                                      "leg_%s_value" % tag: keep["value"], "leg_%s_ms_per_step" % tag: keep["ms_per_step"],
                                      "leg_%s_roofline_frac" % tag: rf.get("frac")})
                _order_config(out)
            except Exception as e:                       # noqa: BLE001
                also.append({"leg": tag, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
                out["config"]["leg_%s_error" % tag] = "%s: %s" % (type(e).__name__, str(e)[:200])
            del leg
            torch.cuda.empty_cache()
    state["phase"] = "gradient-exchange A/B"
    if (world > 1 or ctx["force_dp"]) and default_headline and not a.no_allreduce_ab:
        # Every gradient exchange pinned once in the SAME run, beside the headline's start-up selection (distributed.ExchangeTuner): the evidence
        # table for the first run on RCCL over xGMI
        ab = out["grad_exchange_legs"] = {"headline": {"exchange": out["config"].get("grad_exchange"), "ms_per_step": out["ms_per_step"],
                                                       "value": out["value"], "allreduce_ms_per_step": out.get("allreduce_ms_per_step")}}
        for tag, env in (("allreduce", {"SR_GRAD_EXCHANGE": "allreduce"}), ("overlap", {"SR_GRAD_EXCHANGE": "overlap"}),
                         ("allreduce_bf16", {"SR_GRAD_EXCHANGE": "allreduce_bf16"}), ("rs_ag", {"SR_GRAD_EXCHANGE": "rs_ag"}),
                         ("rs_ag_overlap", {"SR_GRAD_EXCHANGE": "rs_ag_overlap"})):
            leg, old_env = None, {k: os.environ.get(k) for k in env}
            try:
                os.environ.update(env)
                leg = Leg(a, ctx, net=a.net, img=a.img, bu=a.bu, bl=a.bl, regime=a.regime, alg=a.alg)
                # same step schedule as the headline leg (its tuned deferred share): the legs differ in the gradient exchange only, and none
                # of them spends ~30 tuning steps (each with a blocking rank agreement) before its timed region
                leg.alg.defer_share = out["config"].get("deferred_share")
                leg.graph_mode = "0"
                o = leg.run(max(4, a.steps // 2), 2, 3, roofline=False)
                ab[tag] = {"ms_per_step": o["ms_per_step"], "value": o["value"], "allreduce_ms_per_step": o.get("allreduce_ms_per_step"),
                           "note": {"allreduce": "one all-reduce of the flat fp32 block after the backward",
                                    "overlap": "allreduce_ms_per_step = what is left exposed behind the backward (event pair around all_reduce_grads)",
                                    "allreduce_bf16": "gradient block exchanged as bf16 (a rounded sum: NOT the reference's fp32 DDP buckets)",
                                    "rs_ag": "reduce-scatter + all-gather of the flat fp32 block instead of one all-reduce",
                                    "rs_ag_overlap": "reduce-scatter + all-gather per layer-group slice under the backward"}[tag]}
            except Exception as e:                       # noqa: BLE001
                ab[tag] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            finally:
                for k, v in old_env.items():
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            del leg
            torch.cuda.empty_cache()
    if "grad_exchange_legs" in out:
        # the pinned legs beside the start-up selection of the headline: on gloo / shared devices they say nothing about xGMI
        ab = out["grad_exchange_legs"]
        timed = {k: v["ms_per_step"] for k, v in ab.items() if k not in ("headline", "allreduce_bf16") and isinstance(v, dict) and v.get("ms_per_step")}
        if timed:
            best = min(timed, key=timed.get)
            ab["fastest_fp32"] = {"leg": best, "ms_per_step": timed[best], "measured_on": ctx["backend_note"]}
    state["phase"] = "cpu baseline"
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline and default_headline:
            out["cpu_baseline"] = cpu_baseline(a.bl or a.bu, a.bu)
        state["done"] = True
        watchdog.cancel()
        print(json.dumps(out), flush=True)
    state["done"] = True
    watchdog.cancel()
    if world > 1 or ctx["force_dp"]:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    worker(a)


if __name__ == "__main__":
    main()
