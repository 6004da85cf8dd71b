"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: flat-gradient all-reduce + 1/world scaling equals the
single-process gradient of the concatenated batch; rank-stride sharding; packed reward-statistic all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Flat:
    def __init__(self, n):
        self.flat = torch.zeros(n)
        self.grad = torch.zeros(n)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import hooks_ref as H
    from semireward_amd.distributed import DataParallel, shard_indices
    dp = DataParallel(world, rank, global_reward_threshold=True)
    assert dp.active
    # (1) parameters: rank 0's block wins, like DDP's construction-time broadcast
    m = _Flat(1000)
    m.flat += rank + 1
    dp.broadcast_params(m)
    assert float(m.flat.min()) == 1.0 and float(m.flat.max()) == 1.0
    # (2) gradients: per-rank mean-loss gradients, SUM all-reduce, 1/world scale == gradient of the global-batch mean loss
    rng = np.random.Generator(np.random.PCG64(5))
    W = torch.from_numpy(rng.standard_normal((10, 32)).astype(np.float32))
    X = torch.from_numpy(rng.standard_normal((16, 32)).astype(np.float32))
    Y = torch.from_numpy(rng.integers(0, 10, size=(16,), dtype=np.int64))
    mine = shard_indices(16, rank, world)
    assert mine == list(range(rank, 16, world))                       # reference DistributedSampler: perm[rank::world]
    w = W.clone().requires_grad_(True)
    H.ce_loss_mean(X[mine] @ w.t(), Y[mine]).backward()
    m.grad = w.grad.reshape(-1).clone()
    dp.all_reduce_grads(m)
    got = m.grad / world
    w2 = W.clone().requires_grad_(True)
    H.ce_loss_mean(X @ w2.t(), Y).backward()
    ok_grad = torch.allclose(got, w2.grad.reshape(-1), rtol=1e-5, atol=1e-7)
    # (2b) opt-in bf16 exchange of the gradient block: the same sum to bf16 precision (every rank rounds its block, the sum is rounded)
    dp16 = DataParallel(world, rank, exchange="allreduce_bf16")
    m.grad = w.grad.reshape(-1).clone()
    dp16.all_reduce_grads(m)
    ok_grad = ok_grad and m.grad.dtype == torch.float32 and torch.allclose(m.grad / world, w2.grad.reshape(-1), rtol=2e-2, atol=1e-4)
    # (3) global reward threshold extension: packed (sum..., n) all-reduce == mean over the concatenated ranks
    r_all = torch.from_numpy(rng.random((world, 3, 8)).astype(np.float32))      # [rank, groups, B]
    means = dp.reward_means(r_all[rank].reshape(-1), 3)
    ok_mean = torch.allclose(means, r_all.permute(1, 0, 2).reshape(3, -1).mean(1), rtol=1e-6)
    # (4) FreeMatch statistics: all-reduced column sums / histogram + gathered max-probs == statistics of the concatenated batch
    pr = torch.softmax(torch.from_numpy(rng.standard_normal((world, 8, 10)).astype(np.float32)), -1)      # [rank, Bu, C]
    mine_p = pr[rank]
    colsum, hist = mine_p.sum(0).clone(), torch.bincount(mine_p.argmax(-1), minlength=10).float()
    allp, n_all = dp.gather_stats(mine_p.max(-1).values.contiguous(), colsum, hist)
    full = pr.reshape(-1, 10)
    ok_fm = (n_all == 16 and torch.allclose(colsum, full.sum(0), rtol=1e-6) and torch.equal(hist, torch.bincount(full.argmax(-1), minlength=10).float())
             and torch.equal(allp, full.max(-1).values))
    q.put((rank, bool(ok_grad), bool(ok_mean and ok_fm)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]


def _auto_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semireward_amd.distributed import EXCHANGES, DataParallel
    dp = DataParallel(world, rank)                     # SR_GRAD_EXCHANGE unset: auto
    assert dp.requested == "auto" and not dp.settled
    n = 3 * 64 * 40 + 17
    rng = np.random.Generator(np.random.PCG64(9))
    blocks = torch.from_numpy(rng.standard_normal((world, n)).astype(np.float32))
    m = _Flat(n)
    ok = True
    for step in range(3):
        m.grad = blocks[rank].clone() * (step + 1)
        dp.all_reduce_grads(m)                          # step 0 times both collective forms on a scratch block first; the gradient is summed ONCE
        ok = ok and torch.allclose(m.grad.double(), blocks.double().sum(0) * (step + 1), rtol=1e-6, atol=1e-6)
    rep = dp.exchange_report
    # a host-memory block cannot be reduced under a backward: the selection is over after the collective forms
    ok = ok and dp.settled and rep["chosen"] in EXCHANGES and not rep["chosen"].endswith("overlap") and set(rep["collective_ms"]) == {"allreduce", "rs_ag"}
    q.put((rank, bool(ok), rep["chosen"], rep["agreement_syncs"]))
    dist.barrier()
    dist.destroy_process_group()


def _refused_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semireward_amd.distributed import DataParallel

    def refuse(*a, **k):                                 # an argument check of the backend: raised before anything is enqueued, on every rank alike
        raise RuntimeError("reduce_scatter_tensor: in-place shards are not supported by this backend (test stand-in)")
    dist.reduce_scatter_tensor = refuse
    dp = DataParallel(world, rank)
    n = 2 * 64 * 40 + 5
    rng = np.random.Generator(np.random.PCG64(10))
    blocks = torch.from_numpy(rng.standard_normal((world, n)).astype(np.float32))
    m = _Flat(n)
    ok = True
    for step in range(2):
        m.grad = blocks[rank].clone()
        dp.all_reduce_grads(m)
        ok = ok and torch.allclose(m.grad.double(), blocks.double().sum(0), rtol=1e-6, atol=1e-6)
    rep = dp.exchange_report
    q.put((rank, bool(ok), rep["chosen"], "rs_ag_refused" in rep, rep["collective_ms"]["rs_ag"]))
    dist.barrier()
    dist.destroy_process_group()


def test_a_refused_reduce_scatter_form_is_agreed_on_a_tiny_block_before_anything_large_is_timed():
    """The reduce-scatter + all-gather form is probed ONCE on a tiny block and the refusal agreed between the ranks (all-reduce MAX of a flag)
    before the full-size block is timed: every rank then skips the form alike, keeps the all-reduce, reports the refusal -- and the gradients
    of that very step are summed exactly once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_refused_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] == "allreduce" and r[3] and r[4] is None for r in res), res


def test_gradient_exchange_is_selected_at_start_up_and_agreed_between_the_ranks():
    """SR_GRAD_EXCHANGE=auto (the default): the first exchange call probes the reduce-scatter form on a tiny block and AGREES on whether any
    rank was refused (one blocking all-reduce), times one all-reduce against reduce-scatter + all-gather on a scratch block, the ranks agree
    on the maxima over the ranks (a second blocking all-reduce) and ALL keep the same form; the gradients are summed exactly once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_auto_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res) and len({r[2] for r in res}) == 1 and all(r[3] == 2 for r in res), res


def _rs_ag_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semireward_amd.distributed import DataParallel
    ok = True
    # block sizes: one that divides into aligned shards with a tail (the ViT-S block: 21 436 900 elements on 8 ranks leaves a tail too), one
    # smaller than world * SHARD_ALIGN (everything goes through the tail all-reduce), one exactly divisible
    for n in (3 * 64 * 5 + 37, 100, 3 * 64 * 2):
        rng = np.random.Generator(np.random.PCG64(100 + n))
        blocks = torch.from_numpy(rng.standard_normal((world, n)).astype(np.float32))
        want = blocks.double().sum(0)
        for bf16 in (False, True):
            dp = DataParallel(world, rank, exchange="rs_ag")
            dp.bf16_grads = bf16
            m = _Flat(n)
            m.grad = blocks[rank].clone()
            dp.all_reduce_grads(m)
            ref = DataParallel(world, rank, exchange="allreduce_bf16" if bf16 else "allreduce")
            m2 = _Flat(n)
            m2.grad = blocks[rank].clone()
            ref.all_reduce_grads(m2)
            tol = dict(rtol=3e-2, atol=3e-2) if bf16 else dict(rtol=1e-6, atol=1e-6)
            ok = ok and torch.allclose(m.grad.double(), want, **tol) and torch.allclose(m.grad, m2.grad, **tol)
            # every rank holds the same block afterwards (bit for bit: the shards are gathered, not re-summed)
            g0 = m.grad.clone()
            dist.broadcast(g0, src=0)
            ok = ok and torch.equal(g0, m.grad)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_scatter_all_gather_exchange_equals_the_all_reduce():
    """exchange 'rs_ag' (SR_GRAD_EXCHANGE=rs_ag; distributed.DataParallel._sum_over_ranks): in-place reduce-scatter into aligned shards + all-gather + tail
    all-reduce == the single all-reduce of the flat gradient block, on 3 ranks (uneven division), fp32 and the opt-in bf16 exchange."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_rs_ag_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True), (2, True)]


def test_local_mode_is_reference_semantics():
    from semireward_amd.distributed import DataParallel
    dp = DataParallel(1, 0)
    assert not dp.active
    r = torch.arange(16, dtype=torch.float32)
    assert torch.equal(dp.reward_means(r, 2), torch.tensor([3.5, 11.5]))     # per-rank local mean (srflexmatch.py:100)


class _FakeEvent:
    """Stand-in for torch.cuda.Event in the tuner test: 'time' advances by a step cost that depends on the rank and on the share the
    tuner handed out for the step that just ran (set by the worker)."""
    clock = 0.0
    cost = 1.0

    def __init__(self, enable_timing=True):
        self.t = None

    def record(self):
        _FakeEvent.clock += _FakeEvent.cost
        self.t = _FakeEvent.clock

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return other.t - self.t


def _tuner_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semireward_amd.algorithms import srflexmatch as SF
    from semireward_amd.distributed import DataParallel
    dp = DataParallel(world, rank)
    torch.cuda.Event = _FakeEvent
    # the two ranks disagree about the best coarse share (rank 0: 0.42, rank 1: 1.0, which has no neighbours to refine): left alone they
    # would queue different refinements, run a different number of tuning steps and issue a different number of gradient all-reduces
    best_of = {0: 0.42, 1: 1.0}[rank]
    cost = lambda f: (1.0 + abs(f - best_of)) * (1.0 + 0.5 * rank)      # noqa: E731
    tuner = SF._DeferTuner(SF._DeferTuner.CANDIDATES, agree=lambda ms: dp.max_over_ranks(ms, "cpu"))
    seq, prev = [], None
    while not tuner.done and len(seq) < 200:
        _FakeEvent.cost = cost(prev) if prev is not None else 1.0       # the step that just ended ran with the previous share
        prev = tuner.fraction()
        seq.append(prev)
        g = torch.ones(4)
        dist.all_reduce(g)                                               # the step's gradient exchange: hangs if the ranks' step counts differ
    q.put((rank, seq, tuner.best, tuner.report))
    dist.barrier()
    dist.destroy_process_group()


def test_deferred_share_tuner_takes_the_same_decisions_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_tuner_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, b0, r0), (_, s1, b1, r1) = res
    assert s0 == s1 and b0 == b1 and r0 == r1 and len(s0) > len(_FakeEvent.__mro__)       # same shares step for step, same choice, same table
    from semireward_amd.algorithms.srflexmatch import _DeferTuner
    assert len(s0) >= len(_DeferTuner.CANDIDATES) * (_DeferTuner.WARM + _DeferTuner.TIMED)
