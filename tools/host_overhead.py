"""How long does the host need to ENQUEUE one step (no sync) vs the GPU to execute it?"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from semireward_amd.algorithms import get_algorithm
from semireward_amd.nets import vit
from semireward_amd.utils import synth

args = argparse.Namespace(gpu=0, rank=0, world_size=1, distributed=False, **bench.NS)
alg = get_algorithm(args, vit.vit_small_patch2_32)
alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
b = synth.synth_batch(100, 8, 8, 32, 100, 50000)
batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
alg.it = bench.START_IT
alg.optimizer.sched_step = alg.it
def step():
    alg.out_dict, alg.log_dict = alg.train_step(**batch)
    alg.call_hook("after_train_step")
    alg.it += 1
for _ in range(3):
    step()
torch.cuda.synchronize()
# (a) enqueue-only time: GPU kept busy by a long dummy kernel first so the host never waits on a full queue
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
