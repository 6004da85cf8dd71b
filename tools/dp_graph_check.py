"""HIP-graph replay of the step UNDER data parallel (core/stepgraph.py _split): train_step replayed as a graph, the gradient exchange and the
one-launch optimizer step issued eagerly behind it -- against the same data-parallel steps launched eagerly, from equal states.
    SR_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/dp_graph_check.py
    python tools/dp_graph_check.py --one-rank-nccl        (one forced rank on a 1-rank RCCL communicator)
Per step (both engines start it from the eager engine's state, copied in place): forward features bit-equal, FlexMatch table equal, parameters
within the round-off of one AdamW step, identical on all ranks; the variant with the rewarder update (a collective inside train_step) runs
eagerly, every other step is a replay."""
import argparse
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SR_DEFER_FRACTION"] = "0.475"
os.environ["SR_GRAD_EXCHANGE"] = "allreduce"
import torch                                              # noqa: E402
import torch.distributed as dist                          # noqa: E402
import bench                                              # noqa: E402
from semireward_amd.algorithms import get_algorithm       # noqa: E402
from semireward_amd.core.stepgraph import StepGraph       # noqa: E402
from semireward_amd.nets import vit                       # noqa: E402
from semireward_amd.utils import synth                    # noqa: E402

one = "--one-rank-nccl" in sys.argv
if one:
    rank, world, backend = 0, 1, "nccl"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
else:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("SR_DIST_BACKEND", "nccl")
    torch.cuda.set_device(0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend, rank=rank, world_size=world)


def make(graphed):
    args = argparse.Namespace(gpu=torch.cuda.current_device(), rank=rank, world_size=world, distributed=True, force_dp=one, **bench.NS)
    alg = get_algorithm(args, vit.vit_small_patch2_32)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.dp.broadcast_params(alg.model, alg.rewarder, alg.generator)
    alg.model.seed = 99 + rank
    alg.it = bench.START_IT + 5
    alg.optimizer.sched_step = alg.it
    alg.optimizer.step_count = 7
    alg.model.train()
    assert alg.dp.active
    return alg, (StepGraph(alg, warm=1) if graphed else None)


def copy_state(dst, src):
    dst.model.load_state_dict(src.model.state_dict())
    dst.optimizer.load_state_dict(src.optimizer.state_dict())
    dst.rewarder.load_state_dict(src.rewarder.state_dict()); dst.generator.load_state_dict(src.generator.state_dict())
    dst.rewarder_optimizer.load_state_dict(src.rewarder_optimizer.state_dict())
    dst.max_reward.copy_(src.max_reward)
    hs, hd = src.hooks_dict["MaskingHook"], dst.hooks_dict["MaskingHook"]
    hd.selected_label.copy_(hs.selected_label); hd.classwise_acc.copy_(hs.classwise_acc); hd.hist.copy_(hs.hist)
    dst.model._rng_calls = src.model._rng_calls
    torch.cuda.synchronize()


a0, _ = make(False)
a1, sg = make(True)
n = 12
batches = [a0.process_batch(**{k: torch.from_numpy(v) for k, v in synth.synth_batch(700 + 10 * rank + i, 8, 8, 32, 100, 50000).items()}) for i in range(n)]
ok = True
for i in range(n):
    before = a0.model.flat.clone()
    out0, _ = a0.train_step(**batches[i])
    a0.out_dict = out0
    a0.hooks_dict["ParamUpdateHook"].after_train_step(a0)
    a0.it += 1
    out1, _ = sg.step(**batches[i])
    a1.it += 1
    torch.cuda.synchronize()
    upd = float((a0.model.flat - before).abs().max())
    h0, h1 = a0.hooks_dict["MaskingHook"], a1.hooks_dict["MaskingHook"]
    same_fwd = bool(torch.equal(out0["feat"]["x_ulb_w"], out1["feat"]["x_ulb_w"])) and bool(torch.equal(h0.selected_label, h1.selected_label))
    dmax = float((a0.model.flat - a1.model.flat).abs().max())
    other = a1.model.flat.clone()
    dist.broadcast(other, src=0)
    same_ranks = bool(torch.equal(other, a1.model.flat))
    good = same_fwd and dmax <= 2.1 * upd and same_ranks and a1.optimizer.step_count == a0.optimizer.step_count
    ok = ok and good
    if not good:
        print("rank %d step %d: forward equal %s, |dp| %.3e (one update %.3e), ranks agree %s" % (rank, i, same_fwd, dmax, upd, same_ranks), flush=True)
    copy_state(a1, a0)
ok = ok and len(sg.graphs) == 1 and sg.replays >= n - 4 and sg.eager_steps >= 2
print("rank %d: %d replays, %d eager steps (warm-up + the rewarder-update variant), %d captured variant(s) on %s: graph replay under data parallel == eager: %s"
      % (rank, sg.replays, sg.eager_steps, len(sg.graphs), backend, ok), flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
