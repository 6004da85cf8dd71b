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
    lines = [l for l in r.stdout.splitlines() + r.stderr.splitlines() if l.startswith("rank ")]
    assert r.returncode == 0 and len(lines) == 2, (r.stdout[-2000:], r.stderr[-3000:])
