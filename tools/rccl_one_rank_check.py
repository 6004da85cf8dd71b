"""Every collective the data-parallel engine issues, put through RCCL on ONE GPU: a 1-rank ``nccl`` communicator (what
``init_process_group('nccl', ...)`` of semilearn/train.py:374-379 creates per rank) with the data-parallel path forced on
(distributed.DataParallel.force: the reference wraps the model in DDP whenever args.distributed is set, misc.py:55-58, whatever the world size).
    python tools/rccl_one_rank_check.py            (started by tests/test_gpu_rccl_one_rank.py; prints one verdict line per section)
Sections:
  grad      the five pinned gradient exchanges (allreduce | rs_ag in place | overlap | rs_ag_overlap | allreduce_bf16) on the real 85.7 MB block
            inside real SRFlexMatch steps (ViT-S/2, 100 classes, 8 / 8 / 8, K = 8), side by side with a non-data-parallel instance that steps
            in lockstep: the block behind the exchange == the block in front of it, BIT FOR BIT (bf16 exchange: == that block rounded to bf16;
            the exchanges under the backward have no "in front": their result is compared with the other instance's gradient), the exchanged
            gradient and the parameters after every optimizer step against the other instance's (<= 1e-6: the backward's LayerNorm / patch
            atomics are not run-to-run deterministic at 1e-9, with or without data parallel), and every element of the block travels exactly
            ONCE per step (counted at the torch.distributed entry points -- with one rank a range reduced twice has the same VALUE);
  auto      SR_GRAD_EXCHANGE=auto: ExchangeTuner through all of its phases on RCCL until settled, no refusal of the reduce-scatter form;
  bcast     broadcast_params of model / rewarder / generator;
  schedule  the data-parallel step (exchange under the backward, its own communication stream) takes the time of the step without data parallel:
            the step's streams really run beside each other (ops.streams_overlap; HIP hardware-queue aliasing, profiles/r06_hw_queue_aliasing.txt);
  reward    the global reward threshold (packed (sum, n) all-reduce per step, reward_means) against the rank-local mean;
  stats     SoftMatch / FreeMatch / DistAlign statistics (gather_stats: all_reduce of column sums + histogram, all_gather of the max-probs);
  syncbn    the WideResNet's SyncBatchNorm exchanges (forward accumulators incl. the row-count cells, single-pass and shared-launch forwards;
            the two column sums of every BatchNorm backward).
With one rank every sum over the ranks is the identity, so "data parallel == not data parallel" is an exact statement; what the section proves
is that the calls (dtypes, in-place aliasing of reduce_scatter_tensor / all_gather_into_tensor, slices at 256-byte shard boundaries, the
communication stream's event ordering, device_id= initialisation) are accepted and completed by the backend this engine is written for."""
import argparse
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SR_DEFER_FRACTION"] = "0.475"                 # (read at import; the step-schedule tuner is not under test here)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402
import torch.distributed as dist                          # noqa: E402
import bench                                              # noqa: E402
from semireward_amd.algorithms import get_algorithm       # noqa: E402
from semireward_amd.distributed import DataParallel       # noqa: E402
from semireward_amd.nets import vit, wrn                  # noqa: E402
from semireward_amd.utils import synth                    # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
report = {"backend": dist.get_backend(), "world_size": 1, "torch": torch.__version__, "hip": torch.version.hip}


class Counter:
    """Elements handed to the collectives, counted at the torch.distributed entry points the engine calls."""

    def __init__(self):
        self.n = {"all_reduce": 0, "reduce_scatter_tensor": 0, "all_gather_into_tensor": 0, "all_gather": 0, "broadcast": 0}
        self.calls = dict(self.n)
        self.real = {k: getattr(dist, k) for k in self.n}
        self.on_comm_stream = 0

        def wrap(name, size_of):
            def f(*a, **k):
                self.n[name] += size_of(*a, **k)
                self.calls[name] += 1
                return self.real[name](*a, **k)
            return f
        dist.all_reduce = wrap("all_reduce", lambda t, *a, **k: t.numel())
        dist.reduce_scatter_tensor = wrap("reduce_scatter_tensor", lambda out, inp, *a, **k: inp.numel())
        dist.all_gather_into_tensor = wrap("all_gather_into_tensor", lambda out, inp, *a, **k: out.numel())
        dist.all_gather = wrap("all_gather", lambda outs, t, *a, **k: t.numel())
        dist.broadcast = wrap("broadcast", lambda t, *a, **k: t.numel())

    def snap(self):
        return dict(self.n), dict(self.calls)

    def restore(self):
        for k, v in self.real.items():
            setattr(dist, k, v)


cnt = Counter()
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))   # noqa: E731


def make(force, exchange=None, **kw):
    if exchange is not None:
        os.environ["SR_GRAD_EXCHANGE"] = exchange
    ns = dict(bench.NS)
    ns.update(kw)
    args = argparse.Namespace(gpu=0, rank=0, world_size=1, distributed=bool(force), force_dp=bool(force), **ns)
    alg = get_algorithm(args, vit.vit_small_patch2_32)
    alg.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(alg.model.names_shapes, 0).items()})
    alg.model.seed = 1234
    alg.it = bench.START_IT + 7                           # K = 8; the rewarder update of every N_k-th step falls inside the run
    alg.optimizer.sched_step = alg.it
    alg.model.train()
    return alg


def full_step(alg, batch, grab, ident=None):
    """train_step + ParamUpdateHook; ``grab``: list that receives the gradient block as the optimizer is about to read it; ``ident``: list that
    receives (block in front of the exchange, block behind it) when the exchange runs after the backward."""
    real = alg.optimizer.step

    def step(*a, **k):
        grab.append(alg.model.grad.clone())
        return real(*a, **k)
    alg.optimizer.step = step
    real_x = alg.dp._all_reduce_grads
    if ident is not None:
        def exchange(model):
            pre = model.grad.clone()
            real_x(model)
            ident.append((pre, model.grad.clone()))
        alg.dp._all_reduce_grads = exchange
    try:
        alg.out_dict, alg.log_dict = alg.train_step(**batch)
        alg.call_hook("after_train_step")
    finally:
        alg.optimizer.step = real
        if ident is not None:
            del alg.dp._all_reduce_grads
    alg.it += 1


TOL = 1e-6
ok_all = True
b = synth.synth_batch(100, 8, 8, 32, 100, 50000)
# ---- grad: the five pinned exchanges -----------------------------------------------------------------------------------------------
report["grad"] = {}
for exch in ("allreduce", "rs_ag", "overlap", "rs_ag_overlap", "allreduce_bf16"):
    alg, ref = make(True, exch), make(False, "allreduce")
    assert alg.dp.active and alg.dp.force and alg.dp.exchange == ("allreduce" if exch == "allreduce_bf16" else exch) and not ref.dp.active
    assert (alg.model.grad_ready_cb is not None) == exch.endswith("overlap")
    batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
    numel = alg.model.grad.numel()
    worst_g = worst_p = 0.0
    once, exact = True, True
    under = exch.endswith("overlap")
    for step in range(4):                                 # it = 30007 .. 30010: the last one carries a rewarder update (its gradient is all-reduced too)
        n0, c0 = cnt.snap()
        ga, gr, idt = [], [], (None if under else [])
        full_step(alg, batch, ga, idt)
        full_step(ref, batch, gr)
        torch.cuda.synchronize()
        n1, c1 = cnt.snap()
        if idt is not None:                               # sum over ONE rank = identity, exactly (bf16 exchange: the rounding to bf16, exactly)
            pre, post = idt[0]
            exact = exact and bool(torch.equal(post, pre.to(torch.bfloat16).float() if exch == "allreduce_bf16" else pre))
        if exch != "allreduce_bf16" or step == 0:          # (a bf16-rounded gradient moves the parameters: later steps are other steps)
            want = gr[0].to(torch.bfloat16).float() if exch == "allreduce_bf16" else gr[0]
            worst_g = max(worst_g, rel(ga[0], want))
            worst_p = max(worst_p, rel(alg.model.flat, ref.model.flat))
        rew = alg.rewarder.grad.numel() if (alg.it - 1) % alg.N_k == 0 else 0
        moved_ar = n1["all_reduce"] - n0["all_reduce"] - rew
        moved_rs = n1["reduce_scatter_tensor"] - n0["reduce_scatter_tensor"]
        moved_ag = n1["all_gather_into_tensor"] - n0["all_gather_into_tensor"]
        once = once and (moved_ar + moved_rs == numel) and (moved_ag == moved_rs) and (("rs_ag" in exch) == (moved_rs > 0))
    tol_g, tol_p = (1e-4, 1e-3) if exch == "allreduce_bf16" else (TOL, TOL)      # (bf16: values on a rounding boundary may fall the other way)
    good = once and exact and worst_g <= tol_g and worst_p <= tol_p and alg.dp.exchange_report.get("chosen") == exch
    ok_all = ok_all and good
    report["grad"][exch] = dict(ok=good, every_element_exactly_once=once, exchange_is_exact=(None if under else exact), grad_rel=worst_g,
                                params_rel=worst_p, steps=4, block_bytes=4 * numel)
    print("grad[%s]: every element of the %.1f MB block exactly once per step: %s; block behind the exchange == block in front of it%s: %s; "
          "gradient vs the non-DP instance rel %.1e, parameters after the optimizer steps rel %.1e: %s" % (
              exch, 4e-6 * numel, once, " rounded to bf16" if exch == "allreduce_bf16" else "", "(under the backward: n/a)" if under else exact,
              worst_g, worst_p, "OK" if good else "FAILED"), flush=True)
    del alg, ref
    torch.cuda.empty_cache()

# ---- auto: the start-up selection on the live backend ----------------------------------------------------------------------------------
alg, ref = make(True, "auto"), make(False, "allreduce")
batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
def copy_state(dst, src):
    """Every step of the pair starts from the SAME state (two free-running trajectories part ways after a handful of steps once a reward within
    round-off of its pass mean flips a 0/1 mask -- with or without data parallel, tests/test_gpu_stepgraph.py): the non-DP instance's, in place."""
    dst.model.load_state_dict(src.model.state_dict())
    dst.optimizer.load_state_dict(src.optimizer.state_dict())
    dst.rewarder.load_state_dict(src.rewarder.state_dict())
    dst.rewarder_optimizer.load_state_dict(src.rewarder_optimizer.state_dict())
    dst.max_reward.copy_(src.max_reward)
    hs, hd = src.hooks_dict["MaskingHook"], dst.hooks_dict["MaskingHook"]
    hd.selected_label.copy_(hs.selected_label); hd.classwise_acc.copy_(hs.classwise_acc); hd.hist.copy_(hs.hist)
    dst.model._rng_calls = src.model._rng_calls
    torch.cuda.synchronize()


steps, worst = 0, 0.0
while not alg.dp.settled and steps < 40:
    ga, gr = [], []
    full_step(alg, batch, ga)
    full_step(ref, batch, gr)
    steps += 1
    torch.cuda.synchronize()
    worst = max(worst, rel(alg.model.flat, ref.model.flat), rel(ga[0], gr[0]))
    copy_state(alg, ref)
for _ in range(3):                                        # ... and on the selected exchange
    ga, gr = [], []
    full_step(alg, batch, ga)
    full_step(ref, batch, gr)
    torch.cuda.synchronize()
    worst = max(worst, rel(alg.model.flat, ref.model.flat), rel(ga[0], gr[0]))
    copy_state(alg, ref)
rep = alg.dp.exchange_report or {}
good = alg.dp.settled and rep.get("chosen") in ("allreduce", "rs_ag", "overlap", "rs_ag_overlap") and "rs_ag_refused" not in rep and worst <= TOL and \
    "step_ms_exchange_under_backward" in rep and rep["collective_ms"]["rs_ag"] is not None
ok_all = ok_all and good
report["auto"] = dict(ok=good, tuning_steps=steps, worst_rel_vs_non_dp=worst, **{k: v for k, v in rep.items()})
print("auto: ExchangeTuner settled after %d steps on %s: %s; gradient and parameters of every step vs the non-DP instance rel <= %.1e: %s" % (
    steps, rep.get("chosen"), json.dumps(rep), worst, "OK" if good else "FAILED"), flush=True)
# ---- bcast -----------------------------------------------------------------------------------------------------------------------------
n0, c0 = cnt.snap()
before = [m.flat.clone() for m in (alg.model, alg.rewarder, alg.generator)]
alg.dp.broadcast_params(alg.model, alg.rewarder, alg.generator)
torch.cuda.synchronize()
n1, c1 = cnt.snap()
good = c1["broadcast"] - c0["broadcast"] == 3 and all(torch.equal(x, m.flat) for x, m in zip(before, (alg.model, alg.rewarder, alg.generator)))
ok_all = ok_all and good
report["bcast"] = dict(ok=good, elements=n1["broadcast"] - n0["broadcast"])
print("bcast: model / rewarder / generator blocks (%d elements) broadcast from rank 0, unchanged: %s" % (n1["broadcast"] - n0["broadcast"], "OK" if good else "FAILED"),
      flush=True)
del alg, ref
torch.cuda.empty_cache()

# ---- schedule: the step under data parallel is as fast as without ---------------------------------------------------------------------------
# (profiles/r06_hw_queue_aliasing.txt: with the communicator's streams in the process the step's second stream once shared a hardware queue with
# the step's own stream and the two launch trains serialised: 6.8 instead of 4.9 ms.  Streams are now chosen by measured overlap.)
import time                                                # noqa: E402


def ms_per_step(a_, n=15, warm=4):
    for i_ in range(warm + n):
        if i_ == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        a_.out_dict, a_.log_dict = a_.train_step(**batch)
        a_.call_hook("after_train_step")
        a_.it += 1
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


alg, ref = make(True, "overlap"), make(False, "allreduce")
batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
from semireward_amd import ops as _ops                      # noqa: E402
main_ = torch.cuda.current_stream()
overlaps = dict(side=_ops.streams_overlap(main_, alg._side_stream), comm=_ops.streams_overlap(main_, alg.dp._comm),
                side_comm=_ops.streams_overlap(alg._side_stream, alg.dp._comm))
t_ref, t_dp = ms_per_step(ref), ms_per_step(alg)
t_ref2 = ms_per_step(ref)
good = all(overlaps.values()) and t_dp <= 1.2 * min(t_ref, t_ref2)
ok_all = ok_all and good
report["schedule"] = dict(ok=good, ms_per_step_dp_overlap_exchange=t_dp, ms_per_step_no_dp=min(t_ref, t_ref2), streams_overlap=overlaps)
print("schedule: step with the exchange under the backward on RCCL %.3f ms, without data parallel %.3f ms; second / communication streams run beside the "
      "step's stream and beside each other: %s: %s" % (t_dp, min(t_ref, t_ref2), overlaps, "OK" if good else "FAILED"), flush=True)
del alg, ref
torch.cuda.empty_cache()

# ---- reward: global reward threshold -----------------------------------------------------------------------------------------------------
alg, ref = make(True, "allreduce", global_reward_threshold=True), make(False, "allreduce")
assert alg.dp.global_reward_threshold
batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in b.items()})
alg.trace, ref.trace = {}, {}
n0, c0 = cnt.snap()
alg.train_step(**batch)
n1, c1 = cnt.snap()
ref.train_step(**batch)
torch.cuda.synchronize()
r, m2 = alg.trace["reward"].view(8, 8), alg.trace["mask2"].view(8, 8)
expect = (r >= r.mean(dim=1, keepdim=True)).float()
good = bool(torch.equal(m2, expect)) and bool(torch.equal(alg.trace["mask2"], ref.trace["mask2"])) and bool(torch.equal(alg.trace["reward"], ref.trace["reward"])) \
    and (c1["all_reduce"] - c0["all_reduce"]) == 1 and (n1["all_reduce"] - n0["all_reduce"]) == 9 and 0 < float(m2.sum()) < 64
ok_all = ok_all and good
report["reward"] = dict(ok=good, packed_floats=n1["all_reduce"] - n0["all_reduce"], rows_kept=int(m2.sum()))
print("reward: packed (sum per pass, n) all-reduce of %d floats, mask2 == (reward >= global mean) == rank-local mask (one rank), %d of 64 rows kept: %s"
      % (n1["all_reduce"] - n0["all_reduce"], int(m2.sum()), "OK" if good else "FAILED"), flush=True)
del alg, ref
torch.cuda.empty_cache()

# ---- stats: SoftMatch (+ DistAlign) and FreeMatch statistics -----------------------------------------------------------------------------
report["stats"] = {}
for name, extra in (("srsoftmatch", dict(dist_align=True, dist_uniform=True, ema_p=0.999, n_sigma=2, per_class=False)),
                    ("srfreematch", dict(ema_p=0.999, use_quantile=True, clip_thresh=False, ent_loss_ratio=0.001))):
    algs = []
    for force in (True, False):
        ns = dict(bench.NS, algorithm=name, **extra)
        args = argparse.Namespace(gpu=0, rank=0, world_size=1, distributed=force, force_dp=force, **ns)
        os.environ["SR_GRAD_EXCHANGE"] = "allreduce"
        a = get_algorithm(args, vit.vit_small_patch2_32)
        a.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(a.model.names_shapes, 0).items()})
        a.model.seed, a.it = 1234, bench.START_IT + 7
        a.optimizer.sched_step = a.it
        a.model.train()
        a.trace = {}
        algs.append(a)
    alg, ref = algs
    bb = {k: v for k, v in b.items() if k != "idx_ulb"}
    batch = alg.process_batch(**{k: torch.from_numpy(v) for k, v in bb.items()})
    n0, c0 = cnt.snap()
    first = []
    for a_ in (alg, ref):
        for i_ in range(2):
            a_.out_dict, a_.log_dict = a_.train_step(**batch)
            if i_ == 0:
                torch.cuda.synchronize()
                first.append([m_.clone() for m_ in a_.trace["masks"]])
            a_.call_hook("after_train_step")
        torch.cuda.synchronize()
        if a_ is alg:
            n1, c1 = cnt.snap()
    same = all(torch.equal(x, y) for x, y in zip(*first)) and rel(alg.model.flat, ref.model.flat) <= TOL
    gathered = c1["all_gather"] - c0["all_gather"]
    good = bool(same) and gathered >= 2 * 9                # one gather per masking call, 1 + K calls per step
    ok_all = ok_all and good
    report["stats"][name] = dict(ok=good, all_gather_calls=gathered, all_reduce_calls=c1["all_reduce"] - c0["all_reduce"],
                                 params_rel=rel(alg.model.flat, ref.model.flat))
    print("stats[%s]: %d all_gather + %d all_reduce calls in 2 steps, the 9 masks of the first step bit-equal to the non-DP instance's, parameters rel %.1e: %s" % (
        name, gathered, c1["all_reduce"] - c0["all_reduce"], rel(alg.model.flat, ref.model.flat), "OK" if good else "FAILED"), flush=True)
    del alg, ref, algs
    torch.cuda.empty_cache()

# ---- syncbn: WideResNet ------------------------------------------------------------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(11))
B, HW, C = 8, 16, 10
x = torch.from_numpy(rng.standard_normal((B, 3, HW, HW)).astype(np.float32)).to(dev)
dl = torch.from_numpy((rng.standard_normal((B, C)) / B).astype(np.float32)).to(dev)


def build_wrn():
    m = wrn.WideResNet(num_classes=C, depth=10, widen_factor=2, first_stride=1, device=dev)
    m.init_weights(seed=3)
    m.refresh_operands()
    m.train()
    return m


loc, syn = build_wrn(), build_wrn()
syn.dp = DataParallel(1, 0, force=True)
assert syn.stat_sync and syn.stat_ranks == 1 and not loc.stat_sync
n0, c0 = cnt.snap()
outs = []
for m in (syn, loc):
    lg, ft, ctx = m.forward_features(x, save=True, update_stats=True, tag="a")
    m.backward(ctx, dl)
    lg3, _, _ = m.forward_features(x, save=False, update_stats=False, tag="b", passes=3)
    m.check_equal_rows()
    outs.append((lg.clone(), m.grad.clone(), lg3.clone(), {k: v.clone() for k, v in m.buffers.items()}))
    if m is syn:
        torch.cuda.synchronize()
        n1, c1 = cnt.snap()
torch.cuda.synchronize()
(la, ga_, l3a, ba), (lb, gb_, l3b, bb_) = outs
nbn = len(syn.bn)
calls = c1["all_reduce"] - c0["all_reduce"]
same = torch.equal(la, lb) and torch.equal(l3a, l3b) and all(torch.equal(ba[k], bb_[k]) for k in ba) and rel(ga_, gb_) < 1e-6
# forward: one exchange per BatchNorm (the first carries the row-count cells); the 3 shared-launch passes: one per BatchNorm for all passes; backward:
# one per BatchNorm a gradient flows through (the blocks whose conv1 takes the raw input leave their bn1 without one, wrn.py:50)
good = bool(same) and 2 * nbn < calls <= 3 * nbn
report["syncbn"] = dict(ok=good, batchnorms=nbn, all_reduce_calls=calls, logits_equal=bool(torch.equal(la, lb)), shared_pass_logits_equal=bool(torch.equal(l3a, l3b)),
                        grad_rel=rel(ga_, gb_))
ok_all = ok_all and good
print("syncbn: %d BatchNorms, %d statistics exchanges over RCCL (forward, backward, 3 shared-launch passes), logits / running statistics bit-equal to the "
      "unsynchronised model, gradient rel %.1e: %s" % (nbn, calls, rel(ga_, gb_), "OK" if good else "FAILED"), flush=True)

cnt.restore()
report["ok"] = bool(ok_all)
print("RCCL_ONE_RANK " + json.dumps(report), flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok_all else 1)
