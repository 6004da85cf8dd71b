"""Run with 2 ranks on one GPU (gloo):  SR_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1
--master-port 29534 tools/dp_global_threshold_check.py.  One SRFlexMatch step per rank (its own shard of the unlabeled stream) through the HIP
engine with ``global_reward_threshold`` on and off -- BASELINE.json configs[2]'s "global reward-threshold all-reduce":
  on : mask2 of every pass == (reward >= mean over BOTH ranks' rewards of that pass), bit for bit, identical threshold on both ranks;
  off: mask2 == (reward >= the rank's own mean) -- the reference under DDP (srflexmatch.py:100-101)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import bench
from semireward_amd.algorithms import get_algorithm
from semireward_amd.nets import vit
from semireward_amd.utils import synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("SR_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
NSt = dict(bench.NS, num_classes=10, ulb_dest_len=256, feature_dim=128, num_train_iter=2000, start_timing=100, num_warmup_iter=0)
out = {}
for flag in (True, False):
    args = argparse.Namespace(gpu=0, rank=rank, world_size=world, distributed=True, global_reward_threshold=flag, **NSt)
    alg = get_algorithm(args, vit.vit_tiny_test)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.dp.broadcast_params(alg.model, alg.rewarder, alg.generator)
    assert alg.dp.active and alg.dp.global_reward_threshold == flag
    alg.model.seed = 77 + rank
    b = synth.synth_batch(500 + rank, 4, 4, 8, 10, 256)                 # each rank its own shard
    alg.it = 250                                                        # K = sr_decay() = 9
    alg.optimizer.sched_step = alg.it
    alg.trace = {}
    alg.out_dict, alg.log_dict = alg.train_step(**alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()}))
    alg.call_hook("after_train_step")                                   # gradient all-reduce + optimizer: the step completes on both ranks
    torch.cuda.synchronize()
    K, nu = alg.trace["K"], 4
    r = alg.trace["reward"].float().cpu().view(K, nu)
    m2 = alg.trace["mask2"].cpu().view(K, nu).numpy()
    both = [torch.empty_like(r) for _ in range(world)]
    dist.all_gather(both, r)
    # the engine's threshold: (sum over ranks of the per-rank fp32 sums) / (world * nu), fp32 -- restated here with the same operation order
    sums = torch.stack([x.sum(dim=1) for x in both]).sum(dim=0)
    gmean = (sums / float(world * nu)).numpy()
    lmean = (r.sum(dim=1) / float(nu)).numpy()
    want = (r.numpy() >= (gmean if flag else lmean)[:, None]).astype(np.float32)
    ok = bool(np.array_equal(m2, want))
    differs = bool((want != (r.numpy() >= (lmean if flag else gmean)[:, None])).any())
    out[flag] = (ok, differs)
    print("rank %d: global_reward_threshold=%s K=%d mask2 == expected: %s (the other statistic would give a different mask: %s)" % (
        rank, flag, K, ok, differs), flush=True)
    assert ok, (m2, want)
    params = alg.model.flat.clone()
    other = params.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(other, params), "ranks diverged after the step"
dist.destroy_process_group()
