"""Where does the data-parallel step lose time on ONE rank?  (round 6: bench.py --force-dp 5.66-6.77 ms against 4.89 ms without data parallel,
with an all-reduce that takes 15 us.)  Same process, same algorithm set-up, 20-step regions:
  A  no process group, no data parallel                         (the headline)
  B  1-rank nccl process group created, algorithm NOT data parallel   (is it the communicator's mere existence?)
  C  forced data parallel, SR_GRAD_EXCHANGE=allreduce
  D  as C with dist.all_reduce replaced by a no-op                   (is it the collective call?)
  E  as C with a 1-element all_reduce                                 (is it the size?)"""
import argparse
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SR_DEFER_FRACTION"] = os.environ.get("SR_DEFER_FRACTION", "0.475")
os.environ["SR_GRAD_EXCHANGE"] = "allreduce"
import torch                                              # noqa: E402
import torch.distributed as dist                          # noqa: E402
import bench                                              # noqa: E402
from semireward_amd.algorithms import get_algorithm       # noqa: E402
from semireward_amd.nets import vit                       # noqa: E402
from semireward_amd.utils import synth                    # noqa: E402

torch.cuda.set_device(0)


def make(force):
    args = argparse.Namespace(gpu=0, rank=0, world_size=1, distributed=force, force_dp=force, **bench.NS)
    alg = get_algorithm(args, vit.vit_small_patch2_32)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.model.seed = 1234
    alg.it = bench.START_IT
    alg.optimizer.sched_step = alg.it
    alg.model.train()
    b = synth.synth_batch(100, 8, 8, 32, 100, 50000)
    return alg, alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})


def timed(alg, batch, n=20, warm=5):
    for _ in range(warm):
        alg.out_dict, alg.log_dict = alg.train_step(**batch); alg.call_hook("after_train_step"); alg.it += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        alg.out_dict, alg.log_dict = alg.train_step(**batch); alg.call_hook("after_train_step"); alg.it += 1
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, 1e3 * th / n


alg, batch = make(False)
print("A  no process group, no DP:            %.3f ms/step (host %.3f)" % timed(alg, batch), flush=True)
del alg
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0))
    os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.zeros(8, device="cuda")
dist.all_reduce(t)                                         # (communicator really created)
torch.cuda.synchronize()
alg, batch = make(False)
print("B  process group alive, no DP:         %.3f ms/step (host %.3f)" % timed(alg, batch), flush=True)
del alg
alg, batch = make(True)
assert alg.dp.active
print("C  forced DP, allreduce:               %.3f ms/step (host %.3f)" % timed(alg, batch), flush=True)
if os.environ.get("SR_PHASES", "0") != "0":
    print("   phases:", {k: (round(v[0], 2), round(v[1], 2)) for k, v in alg.phase_report().items()}, flush=True)
real = dist.all_reduce
dist.all_reduce = lambda t_, *a, **k: None
print("D  forced DP, all_reduce = no-op:      %.3f ms/step (host %.3f)" % timed(alg, batch), flush=True)
one = torch.zeros(1, device="cuda")
dist.all_reduce = lambda t_, *a, **k: real(one)
print("E  forced DP, 1-element all_reduce:    %.3f ms/step (host %.3f)" % timed(alg, batch), flush=True)
dist.all_reduce = real
print("C' forced DP, allreduce (again):       %.3f ms/step (host %.3f)" % timed(alg, batch), flush=True)
dist.destroy_process_group()
