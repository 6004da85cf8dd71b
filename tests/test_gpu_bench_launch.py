"""bench.py starts its own ranks (the reference's train.py:344 mp.spawn): the plain driver command ``python bench.py --gpus 2 ...`` must print
one valid JSON line with n_gpus == 2.  On a one-GPU box the two ranks share the device and the collectives run over gloo (bench.py picks
that backend when there are fewer devices than ranks) -- a functional proof of the multi-rank path, not a scaling measurement; the same
command under torch.distributed.run (how the driver launches N > 1) must give the same line shape."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-also", "--no-roofline"]


def _check(line):
    o = json.loads(line)
    assert o["n_gpus"] == 2 and o["steps"] == 2 and o["value"] > 0 and o["config"]["parallelism"] == "dp2"
    assert o["config"]["per_gpu_batch"] == {"lb": 8, "ulb_w": 8, "ulb_s": 8} and o["scaling"] == "weak"
    assert o["allreduce_ms_per_step"] is not None and o["allreduce_ms_per_step"] > 0
    assert ("rccl_ranks" in o) and ("backend" in o["config"])
    return o


def test_plain_command_launches_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    _check(lines[0])


def test_same_command_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py")] + ARGS, env=env, capture_output=True, text=True,
                       timeout=420, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    _check(lines[0])


def test_watchdog_prints_the_headline_when_the_secondary_phases_do_not_finish():
    """After the headline is measured nothing may cost the line: with a budget the secondary legs cannot meet, every rank's watchdog ends its
    process and rank 0 prints the line with the headline fields complete (a collective that one rank never reaches does not raise)."""
    env = dict(os.environ, SR_BENCH_EXTRA_BUDGET="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "1", "--no-roofline"], env=env,
                       capture_output=True, text=True, timeout=420, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    o = json.loads(lines[0])
    assert "truncated" in o and o["value"] > 0 and o["n_gpus"] == 1 and o["steps"] == 3 and o["ms_per_step"] > 0
