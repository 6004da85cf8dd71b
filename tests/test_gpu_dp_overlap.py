"""Data parallel on the GPU engine: two ranks on ONE device (gloo), the all-reduce of the gradient block issued in layer-group slices under
the backward (distributed.DataParallel.install_overlap) against the single all-reduce after it -- tools/dp_overlap_check.py asserts that the
reduced blocks agree, equal the sum of the local gradients, and are identical on both ranks."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sliced_all_reduce_under_the_backward_equals_single_all_reduce():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "dp_overlap_check.py")],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr                  # (the two ranks' lines may interleave on one line: count the verdicts, not the lines)
    assert r.returncode == 0 and out.count("ranks agree: True") == 2, (r.stdout[-2000:], r.stderr[-3000:])


def test_exchange_tuner_deciding_step_reduces_every_range_exactly_once():
    """distributed.ExchangeTuner decides "exchange under the backward or after it" at the top of all_reduce_grads -- AFTER that step's backward
    has already reduced its layer-group ranges.  tools/dp_tuner_decision_check.py walks two ranks through every phase with both outcomes forced
    and compares the exchanged gradient of EVERY step (the deciding one included) with the sum of the ranks' local gradients."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "dp_tuner_decision_check.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and out.count("every step's gradient is the sum over the ranks: True") == 4, (r.stdout[-3000:], r.stderr[-3000:])
    assert out.count("DECIDING") == 4


def test_step_graph_replay_under_data_parallel():
    """core/stepgraph.py under data parallel: train_step replayed as a HIP graph, exchange + optimizer launch eager behind it
    (tools/dp_graph_check.py): two gloo ranks on one GPU, and one forced rank on a 1-rank RCCL communicator."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "dp_graph_check.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and out.count("graph replay under data parallel == eager: True") == 2, (r.stdout[-3000:], r.stderr[-3000:])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SR_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_graph_check.py"), "--one-rank-nccl"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "on nccl: graph replay under data parallel == eager: True" in out, (r.stdout[-3000:], r.stderr[-3000:])


def test_global_reward_threshold_through_the_engine_with_two_ranks():
    """BASELINE.json configs[2]: ``global_reward_threshold`` through SRFlexMatch.train_step on the HIP engine with two gloo ranks on one GPU
    (tools/dp_global_threshold_check.py): mask2 == (reward >= mean over both ranks' rewards) bit for bit with the flag on, the rank-local mask
    (reference semantics) with it off, parameters identical on both ranks after the step."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "dp_global_threshold_check.py")],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr                  # (the two ranks' lines may interleave on one line: count the verdicts, not the lines)
    assert r.returncode == 0 and out.count("mask2 == expected: True") == 4 and "expected: False" not in out, (r.stdout[-2000:], r.stderr[-3000:])
    assert out.count("global_reward_threshold=True") == 2 and out.count("global_reward_threshold=False") == 2


def test_wideresnet_under_data_parallel_is_syncbatchnorm():
    """The reference converts the WideResNet's BatchNorms to SyncBatchNorm under DDP (core/utils/misc.py:55).  tools/dp_syncbn_check.py: two
    ranks with half a batch each == one rank with the whole batch (logits, running statistics, the sum over ranks of the gradients), and the
    same comparison fails by > 10x when the statistics stay per-rank; unequal per-rank batches raise on EVERY rank (no rank-local collective)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "dp_syncbn_check.py")],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and out.count("syncbn == whole batch: True") == 2, (r.stdout[-2000:], r.stderr[-3000:])
    assert out.count("unequal per-rank batches raise on every rank: True") == 2, (r.stdout[-2000:], r.stderr[-3000:])
    assert out.count("unequal per-rank batches raise on every rank (shared launches): True") == 2, (r.stdout[-2000:], r.stderr[-3000:])


def test_allreduce_streams_are_ordered_by_events_not_by_the_backend():
    """The all-reduce under the backward runs on a communication stream (distributed.DataParallel.install_overlap).  gloo's all_reduce
    synchronises the stream it is called on, so the two-rank tests cannot see a missing event; RCCL does not.  Here the collective is replaced by
    an asynchronous stand-in on the communication stream and the compute stream is made slow with a spin kernel:
      * _reduce_range(lo, hi): the collective must see the gradient slice as the compute stream leaves it (comm stream waited on the event
        recorded after the producing launch), although that launch sits behind ~50 ms of queued work when the collective is enqueued;
      * all_reduce_grads: the compute stream must see the collective's result (it waits on the event recorded behind the last slice), although
        the stand-in spins before it writes; the not-yet-reported ranges are reduced there, each exactly once."""
    import torch
    import torch.distributed as dist
    from semireward_amd.distributed import DataParallel

    class FakeDP(DataParallel):
        active = True

    dev = torch.device("cuda:0")
    n = 1 << 20
    model = type("M", (), {})()
    model.grad, model.grad_ready_cb = torch.zeros(n, device=dev), None
    dp = FakeDP(world_size=2, rank=0, exchange="overlap")        # pinned: the start-up selection would time collectives on a scratch block first
    assert dp.install_overlap(model) and model.grad_ready_cb is not None
    seen, calls = [], []
    spin = int(1e8)                                        # ~50 ms at 2 GHz

    def fake_all_reduce(t, op=None):
        cur = torch.cuda.current_stream()
        assert cur == dp._comm, "the collective of a slice belongs on the communication stream"
        calls.append((t.data_ptr() - model.grad.data_ptr()) // 4)
        seen.append(t.clone())                             # what the collective reads (stream-ordered on the comm stream)
        torch.cuda._sleep(spin)
        t.mul_(2.0)                                        # "sum over two ranks holding the same gradients"
    real = dist.all_reduce
    dist.all_reduce = fake_all_reduce
    try:
        lo, hi = n // 2, n
        torch.cuda._sleep(spin)                            # the backward is still busy ...
        model.grad[lo:hi].fill_(3.0)                       # ... and only then produces the slice
        model.grad_ready_cb(lo, hi)                        # = dp._reduce_range: enqueued while the producer has not run yet
        torch.cuda._sleep(spin)
        model.grad[:lo].fill_(5.0)                         # the rest of the backward
        dp.all_reduce_grads(model)                         # reduces [0, lo) and joins the communication stream
        got = model.grad.clone()                           # compute stream: must be ordered behind both collectives
        torch.cuda.synchronize()
    finally:
        dist.all_reduce = real
    assert calls == [lo, 0], calls                         # every range exactly once
    assert float(seen[0].min()) == 3.0 and float(seen[0].max()) == 3.0, "the comm stream did not wait for the producer of the slice"
    assert float(seen[1].min()) == 5.0 and float(seen[1].max()) == 5.0
    assert float(got[lo:].min()) == 6.0 and float(got[:lo].min()) == 10.0 and float(got.max()) == 10.0, "the compute stream did not wait for the collectives"
