"""Run with 2 ranks on one GPU (gloo):  SR_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1
--master-port 29533 tools/dp_overlap_check.py.  One data-parallel step with the all-reduce under the backward (layer-group slices on a
communication stream) and one with the single all-reduce after it, from the same state: the reduced gradient blocks must agree."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from semireward_amd.algorithms import get_algorithm
from semireward_amd.nets import vit
from semireward_amd.utils import synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("SR_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
res = []
for overlap in ("1", "0"):
    os.environ["SR_GRAD_EXCHANGE"] = "overlap" if overlap == "1" else "allreduce"
    args = argparse.Namespace(gpu=0, rank=rank, world_size=world, distributed=True, **bench.NS)
    alg = get_algorithm(args, vit.vit_small_patch2_32)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.dp.broadcast_params(alg.model, alg.rewarder, alg.generator)
    assert (alg.model.grad_ready_cb is not None) == (overlap == "1")
    alg.model.seed = 1234 + rank
    b = synth.synth_batch(100 + rank, 8, 8, 32, 100, 50000)
    batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
    alg.it = bench.START_IT
    alg.optimizer.sched_step = alg.it
    alg.model.train()
    alg.out_dict, alg.log_dict = alg.train_step(**batch)          # one step, no optimizer: compare the all-reduced GRADIENTS
    local = alg.model.grad.clone() if overlap == "0" else None     # (overlap = 0: nothing is reduced before all_reduce_grads)
    alg.dp.all_reduce_grads(alg.model)
    torch.cuda.synchronize()
    res.append((alg.model.grad.clone(), local))
(ga, _), (gb, local) = res
want = local.clone()
dist.all_reduce(want)
torch.cuda.synchronize()
rel = lambda a, b: float((a - b).double().norm() / b.double().norm())   # noqa: E731
other = ga.clone()
dist.broadcast(other, src=0)
print("rank %d: |g| %.3e; overlapped vs single all-reduce rel %.2e; single vs sum of local rel %.2e; ranks agree: %s; overlapped = 2 x local? %.2e" % (
    rank, float(ga.norm()), rel(ga, gb), rel(gb, want), bool(torch.equal(other, ga)), rel(ga, 2 * local)), flush=True)
assert rel(ga, gb) < 1e-4 and rel(gb, want) == 0.0 and torch.equal(other, ga)
dist.destroy_process_group()
