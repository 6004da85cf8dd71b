"""The start-up selection of the gradient exchange (distributed.ExchangeTuner, SR_GRAD_EXCHANGE=auto) walked through ALL of its phases with
two ranks on one GPU, the exchanged gradient checked at EVERY step -- in particular at the step where the schedule is decided: by then that
step's backward has already reduced its layer-group ranges on the communication stream, and the exchange that follows must neither reduce
them a second time (overlap loses: the callback is uninstalled at that very call) nor skip what was not reported.
    SR_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/dp_tuner_decision_check.py
Expected value of a step: the SUM over the ranks of the local gradients, computed by a second, non-data-parallel instance per rank that steps in
lockstep (same parameters, batch, DropPath seeds; no optimizer, ``it`` fixed so that no rewarder update runs).  Both outcomes are forced in turn
(``_after`` = 0: overlap loses; inf: overlap wins)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SR_GRAD_EXCHANGE"] = "auto"
os.environ["SR_DEFER_FRACTION"] = "0.475"                 # (read at import; the step-schedule tuner is not what is under test)
import torch                                              # noqa: E402
import torch.distributed as dist                          # noqa: E402
import bench                                              # noqa: E402
from semireward_amd.algorithms import get_algorithm       # noqa: E402
from semireward_amd.distributed import ExchangeTuner      # noqa: E402
from semireward_amd.nets import vit                       # noqa: E402
from semireward_amd.utils import synth                    # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("SR_DIST_BACKEND", "gloo")
torch.cuda.set_device(0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group(backend, rank=rank, world_size=world)
rel = lambda a, b: float((a - b).double().norm() / (b.double().norm() + 1e-30))   # noqa: E731


def make(dp):
    args = argparse.Namespace(gpu=torch.cuda.current_device(), rank=rank if dp else 0, world_size=world if dp else 1, distributed=dp, **bench.NS)
    alg = get_algorithm(args, vit.vit_small_patch2_32)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.model.seed = 1234 + rank
    alg.it = bench.START_IT + 1                           # K = 8, not a multiple of N_k: no rewarder update (its gradient is averaged under DP)
    alg.optimizer.sched_step = alg.it
    alg.model.train()
    return alg


ok_all = True
for outcome, after in (("overlap loses", 0.0), ("overlap wins", float("inf"))):
    alg, shadow = make(True), make(False)
    assert alg.dp.active and alg.dp.tuner is not None and not shadow.dp.active
    b = synth.synth_batch(100 + rank, 8, 8, 32, 100, 50000)
    batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
    decided_at, worst, phases = None, 0.0, []
    per = ExchangeTuner.WARM + ExchangeTuner.TIMED
    for step in range(2 * (per + 1) + 4):
        alg.train_step(**batch)
        shadow.train_step(**batch)
        want = shadow.model.grad.clone()
        dist.all_reduce(want)
        t = alg.dp.tuner
        phases.append(t.phase if t is not None else 3)
        deciding = t is not None and t.phase == 2 and len(t.marks) == per      # this call records the last mark of phase 2 and decides
        if deciding:
            t._after = after
            decided_at = step
            assert alg.model.grad_ready_cb is not None and len(alg.dp._done) > 0, "the deciding step's backward reported no ranges"
        alg.dp.all_reduce_grads(alg.model)
        torch.cuda.synchronize()
        e = rel(alg.model.grad, want)
        worst = max(worst, e)
        if deciding or e > 1e-4:
            print("rank %d [%s] step %d (phase %d%s): exchanged vs sum of local gradients rel %.2e" % (
                rank, outcome, step, phases[-1], ", DECIDING" if deciding else "", e), flush=True)
        alg.model.grad.zero_()
        shadow.model.grad.zero_()
    rep = alg.dp.exchange_report or {}
    chosen = rep.get("chosen")
    good = (decided_at is not None and worst < 1e-4 and alg.dp.tuner is None and
            chosen in (("allreduce", "rs_ag") if after == 0.0 else ("overlap", "rs_ag_overlap")) and
            (alg.model.grad_ready_cb is None) == (after == 0.0) and "rs_ag_refused" not in rep)
    ok_all = ok_all and good
    print("rank %d [%s]: phases %s decided at step %d, chosen %s, worst rel %.2e over %d steps: every step's gradient is the sum over the ranks: %s"
          % (rank, outcome, "".join(str(p) for p in phases), decided_at if decided_at is not None else -1, chosen, worst, len(phases), good), flush=True)
    del alg, shadow
    torch.cuda.empty_cache()
assert ok_all
dist.barrier()
dist.destroy_process_group()
