#!/usr/bin/env python
"""bench.py -- unlabeled images/s of the SemiReward hot path (SRFlexMatch, ViT-S/2, CIFAR-100 shapes) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = one reference-semantics training iteration in the steady SR regime (it > start_timing, K = sr_decay() = 8
extra backbone passes, rewarder update every N_k = 10 steps): SRFlexMatch.train_step (batched (1+K)-pass forward,
score filter, rewarder scoring, masked losses, hand-written backward) + ParamUpdateHook (gradient all-reduce when
N > 1, fused AdamW / scheduler / zero_grad).  Inputs are synthetic (seeded N(0,1) images), resident
in HBM before the timed region; weights are random-init of the reference architecture.
Workload = BASELINE.json configs[1]: config/SemiReward/usb_cv/flexmatch/flexmatch_cifar100_200_0.yaml
(per-GPU batch 8 labelled + 8 weak + 8 strong).  ``--bu`` scales the per-GPU batch for the throughput variant.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NS = dict(algorithm="srflexmatch", num_classes=100, num_train_iter=204800, epoch=200, ema_m=0.0, ulb_loss_ratio=1.0,
          use_cat=True, amp=False, lr=5e-4, weight_decay=5e-4, layer_decay=0.5, num_warmup_iter=5120, optim="AdamW",
          T=0.5, p_cutoff=0.95, hard_label=True, thresh_warmup=True, ulb_dest_len=50000, N_k=10, start_timing=20000,
          feature_dim=384, sr_lr=5e-4, sr_ema=False, sr_ema_m=0.99)
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (the 5 PF headline is 2:1 sparse)
HBM_PEAK_TBS = 8.0                        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 measured with a float4 copy)
START_IT = 30000                          # it >= 25601 -> sr_decay() == 8 (88 % of the reference run)


def cpu_baseline(bl, bu):
    """Reference-semantics step on the host cores, timed on a bounded sample with the CPU oracle (kind = 'port'):
    one (Bl+2Bu)-image ViT-S/2 pass with autograd graph is timed (fwd) together with the backward of two such graphs
    and one AdamW sweep; the K=8 step time is (1+K)*t_fwd + t_bwd2 + t_opt, exactly the reference's work per step."""
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
    V.vit_forward(P, x[:2], cfg, dp[:, :, :2])                       # warm the allocator / threads
    t0 = time.perf_counter()
    o1 = V.vit_forward(P, x, cfg, dp)
    o2 = V.vit_forward(P, x, cfg, dp)
    t1 = time.perf_counter()
    (H.ce_loss_mean(o1["logits"], y) + H.ce_loss_mean(o2["logits"], y)).backward()
    t2 = time.perf_counter()
    hp = O.vit_param_hparams(V.param_shapes(cfg), cfg.depth, 5e-4, 5e-4, 0.5)
    with torch.no_grad():
        for k, p in P.items():
            O.adamw_step(p, p.grad, torch.zeros_like(p), torch.zeros_like(p), 1, hp[k][0], hp[k][1])
    t3 = time.perf_counter()
    t_fwd, t_bwd2, t_opt = (t1 - t0) / 2, t2 - t1, t3 - t2
    K = 8
    t_step = (1 + K) * t_fwd + t_bwd2 + t_opt
    return {"value": bu / t_step, "unit": "unlabeled images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle ViT-S/2 fp32, Bt=%d: 2 graph forwards (%.2fs each) + backward of both (%.2fs) + AdamW (%.2fs); "
                      "K=8 step = 9*fwd + bwd + opt = %.1fs" % (x.shape[0], t_fwd, t_bwd2, t_opt, t_step)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bu", type=int, default=8, help="per-GPU unlabeled batch (reference yaml: 8, uratio 1)")
    ap.add_argument("--bl", type=int, default=0, help="per-GPU labelled batch (default = --bu, uratio 1)")
    ap.add_argument("--regime", choices=["sr", "pre"], default="sr", help="sr: it > start_timing (K=8); pre: K=0")
    ap.add_argument("--img", type=int, choices=[32, 224], default=32,
                    help="32: ViT-S/2 on 32x32 (north-star config); 224: ViT-S/16 on 224x224 (vit_small_patch16_224, 197 tokens)")
    ap.add_argument("--net", choices=["vit", "bert", "hubert"], default="vit",
                    help="vit: the headline workload (BASELINE.json metric); bert: SRSoftMatch + bert_base_uncased on [B, --seq-len] token "
                         "batches (usb_nlp shapes, BASELINE.json configs[3]); hubert: SRSoftMatch + hubert_base (the Wav2Vec2 architecture) on "
                         "[B, --samples] waveforms (usb_audio shapes, configs[4]) -- both reported under their own metric names")
    ap.add_argument("--samples", type=int, default=64000, help="waveform length (usb_audio: 4 s at 16 kHz)")
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--infer-chunk", type=int, default=0)
    ap.add_argument("--elide-unread-rows", action="store_true",
                    help="NOT the reference's work (never the default, flagged in config): skip the (pass, image) rows whose outputs nothing reads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()
    bl = a.bl or a.bu
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1 and a.gpus == 1, "launch N>1 with torch.distributed.run --nproc-per-node N"
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)            # (debug: several ranks on one device with SR_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SR_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from semireward_amd import ops
    from semireward_amd.algorithms import get_algorithm
    from semireward_amd.nets import vit
    from semireward_amd.utils import synth

    if a.net == "bert":
        # config/SemiReward/usb_nlp/softmatch/softmatch_ag_news_40_0.yaml: bert_base_uncased, 4 classes, batch 8 / uratio 1, max_length 512,
        # use_cat False, AdamW lr 5e-5 wd 5e-4 layer_decay 0.65, dist_align uniform, ema_p 0.999; num_train_iter 102400, start_timing 10000
        from semireward_amd.nets import bert
        nlp = dict(NS, algorithm="srsoftmatch", num_classes=4, num_train_iter=102400, start_timing=10000, lr=5e-5, layer_decay=0.65,
                   use_cat=False, feature_dim=768, dist_align=True, dist_uniform=True, ema_p=0.999, n_sigma=2, per_class=False)
        args = argparse.Namespace(gpu=local, rank=rank, world_size=world, distributed=world > 1, infer_chunk=a.infer_chunk, **nlp)
        alg = get_algorithm(args, bert.bert_base_uncased)                       # random init (no network for the checkpoint)
        alg.dp.broadcast_params(alg.model, alg.rewarder, alg.generator)
        alg.model.seed = 1234 + rank
        g = torch.Generator().manual_seed(100 + rank)
        tok = lambda n: {"input_ids": torch.randint(1, 30522, (n, a.seq_len), generator=g),   # noqa: E731
                         "attention_mask": torch.ones(n, a.seq_len, dtype=torch.int64)}      # full-length rows
        batch = alg.process_batch(x_lb=tok(bl), y_lb=torch.randint(0, 4, (bl,), generator=g), x_ulb_w=tok(a.bu), x_ulb_s=tok(a.bu))
        START = 90001                                                           # sr_decay(): max(8, 1 + 102400 / it) = 8
    elif a.net == "hubert":
        # config/SemiReward/usb_audio/softmatch/softmatch_urbansound8k_100_0.yaml: hubert_base, 10 classes, batch 8 / uratio 1, 4 s at 16 kHz,
        # use_cat False, AdamW lr 5e-5 wd 5e-4 layer_decay 0.75
        from semireward_amd.nets import hubert
        au = dict(NS, algorithm="srsoftmatch", num_classes=10, num_train_iter=102400, start_timing=10000, lr=5e-5, layer_decay=0.75,
                  use_cat=False, feature_dim=768, dist_align=True, dist_uniform=True, ema_p=0.999, n_sigma=2, per_class=False)
        args = argparse.Namespace(gpu=local, rank=rank, world_size=world, distributed=world > 1, infer_chunk=a.infer_chunk, **au)
        alg = get_algorithm(args, hubert.hubert_base)
        alg.dp.broadcast_params(alg.model, alg.rewarder, alg.generator)
        alg.model.seed = 1234 + rank
        g = torch.Generator().manual_seed(100 + rank)
        wv = lambda n: torch.randn(n, a.samples, generator=g)   # noqa: E731
        batch = alg.process_batch(x_lb=wv(bl), y_lb=torch.randint(0, 10, (bl,), generator=g), x_ulb_w=wv(a.bu), x_ulb_s=wv(a.bu))
        START = 90001
    else:
        args = argparse.Namespace(gpu=local, rank=rank, world_size=world, distributed=world > 1, infer_chunk=a.infer_chunk, **NS)
        alg = get_algorithm(args, vit.vit_small_patch2_32 if a.img == 32 else vit.vit_small_patch16_224)
        P = synth.synth_params(alg.model.names_shapes, 0)
        alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
        alg.dp.broadcast_params(alg.model, alg.rewarder, alg.generator)
        alg.model.seed = 1234 + rank
        b = synth.synth_batch(100 + rank, bl, a.bu, a.img, 100, 50000)         # each rank: its own shard of the unlabeled stream
        batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
        START = START_IT
    alg.elide_unread_rows = bool(a.elide_unread_rows)
    alg.it = START if a.regime == "sr" else 1000
    alg.optimizer.sched_step = alg.it
    alg.model.train()

    def step():
        alg.out_dict, alg.log_dict = alg.train_step(**batch)
        alg.call_hook("after_train_step")
        alg.it += 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    # roofline pass: the SAME steps again, in this process, with HIP events around every GEMM launch on the launch
    # stream.  Kept out of the timed region above because 2 x 145 event records per step cost ~0.9 ms of host time.
    prof = None
    if not a.no_roofline:                    # every rank runs the pass (the steps contain collectives); rank 0 reports
        prof = ops.enable_gemm_profile()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ops.disable_gemm_profile()
        prof_steps = a.steps
    if world > 1:
        dist.barrier()
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    K = alg.sr_decay() if a.regime == "sr" else 0
    if rank == 0 and os.environ.get("SR_PHASES", "0") != "0":
        print("phases (ms since step start: gpu, host):", {k: (round(v[0], 2), round(v[1], 2)) for k, v in alg.phase_report().items()}, file=sys.stderr)
    if rank == 0:
        out = {"metric": "unlabeled images/sec/node (FlexMatch+SR, ViT-S CIFAR-100)" if a.net == "vit" else
               "unlabeled sequences/sec/node (SoftMatch+SR, BERT-base, L=%d)" % a.seq_len if a.net == "bert" else
               "unlabeled clips/sec/node (SoftMatch+SR, HuBERT-base, %d samples)" % a.samples, "value": world * a.bu * a.steps / dt,
               "unit": {"vit": "unlabeled images/s", "bert": "unlabeled sequences/s", "hubert": "unlabeled clips/s"}[a.net], "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": ("SRSoftMatch bert_base_uncased, [B, %d] token batches, softmatch_ag_news_40_0.yaml shapes, use_cat False, " % a.seq_len
                                       if a.net == "bert" else
                                       "SRSoftMatch hubert_base, [B, %d] waveforms, softmatch_urbansound8k_100_0.yaml shapes, use_cat False, " % a.samples
                                       if a.net == "hubert" else
                                       "SRFlexMatch ViT-S/2@32 CIFAR-100 shapes, flexmatch_cifar100_200_0.yaml, " if a.img == 32 else
                                       "SRFlexMatch ViT-S/16@224 (vit_small_patch16_224) 224x224x3 batches, 100 classes, ") +
                                      ("steady SR regime" if a.regime == "sr" else "pre-start_timing regime"),
                          "per_gpu_batch": {"lb": bl, "ulb_w": a.bu, "ulb_s": a.bu}, "K_passes": K,
                          "forward_image_passes_per_step": (1 + K) * (bl + 2 * a.bu) - (K * bl if a.net != "vit" else 0) -
                          ((K * bl if a.net == "vit" else 0) + max(K - 1, 0) * a.bu if a.elide_unread_rows else 0),
                          "unread_rows": "ELIDED (opt-in extension: fewer forward rows than the reference executes, results identical)"
                          if a.elide_unread_rows else "computed, as in the reference",
                          "backward_images_per_step": bl + a.bu,
                          "rewarder_update_every": NS["N_k"], "parallelism": "dp%d" % world,
                          "grad_allreduce": "flat fp32 block, 1 RCCL all-reduce/step" if world > 1 else "none"}}
        if prof is not None:
            pk = prof.per_kernel()
            name, (fl, ms_raw, n, nbytes) = max(pk.items(), key=lambda kv: kv[1][1])   # dominant kernel = most time in the pass
            tfl, tms, tn = prof.totals()
            # An event pair brackets the kernel's dispatch and the event packets themselves, not only its execution (rocprofv3 reports the
            # execution alone).  Calibration: the same pair around a one-workgroup kernel on the same stream, median of 200.
            tiny_in, tiny_out = torch.zeros(8, device="cuda"), torch.empty(8, dtype=torch.bfloat16, device="cuda")
            pairs = []
            for _ in range(200):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.cast_f32_bf16(tiny_in, tiny_out, 8); e1.record()
                pairs.append((e0, e1))
            torch.cuda.synchronize()
            # (minus the ~2 us the one-workgroup kernel itself runs according to rocprofv3)
            ovh_ms = max(sorted(x.elapsed_time(y) for x, y in pairs)[100] - 2.0e-3, 0.0)
            ms = max(ms_raw - n * ovh_ms, 0.5 * ms_raw)
            # which roof bounds this kernel?  arithmetic intensity of its launches vs the ridge of the machine
            ridge = MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12 / (HBM_PEAK_TBS * 1e12)
            intensity = fl / nbytes
            traffic = None
            tf = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")          # PMC pass of this same command (tools/pmc.sh)
            if os.path.exists(tf) and a.bu == 8 and a.regime == "sr" and world == 1 and a.img == 32 and a.net == "vit":
                traffic = (json.load(open(tf)).get(name) or {}).get("hbm_bytes_per_launch")
            if intensity < ridge:
                ach = nbytes / (ms * 1e-3) / 1e9
                roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": ach / (HBM_PEAK_TBS * 1e3)}
            else:
                ach = fl / (ms * 1e-3) / 1e12
                roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_BF16_DENSE_PEAK_TFLOPS}
            roof.update({"kernel": name, "traffic": traffic, "launches": n, "avg_launch_us": 1e3 * ms / n,
                         "avg_launch_us_raw_event_pair": 1e3 * ms_raw / n, "event_pair_overhead_us": 1e3 * ovh_ms,
                         "algorithmic_bytes_per_launch": nbytes / n, "flop_per_launch": fl / n, "flop_per_byte": intensity,
                         "ridge_flop_per_byte": ridge, "tflops": fl / (ms * 1e-3) / 1e12, "ms_per_step_in_kernel": ms / prof_steps,
                         "measured_over": "%d instrumented steps run right after the timed region (same process, same inputs)" % prof_steps,
                         "all_gemm_kernels": {"tflops": tfl / (tms * 1e-3) / 1e12, "launches": tn, "ms_per_step": tms / prof_steps}})
            out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline and a.img == 32 and a.net == "vit":
            out["cpu_baseline"] = cpu_baseline(bl, a.bu)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
