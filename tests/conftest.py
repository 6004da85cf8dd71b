import os
import sys

import numpy as np
import pytest

os.environ.setdefault("SRHIP_CHECK_ARGS", "1")      # every libsrhip call of the test suite validates its tensor arguments (ops._p)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __getitem__(self, k):
        return self.z[k]

    def keys(self, prefix=""):
        return [k for k in self.z.files if k.startswith(prefix)]

    def samp(self, prefix):
        return dict(sample=self.z[prefix + "/sample"], stride=int(self.z[prefix + "/stride"]),
                    sum=float(self.z[prefix + "/sum"]), abssum=float(self.z[prefix + "/abssum"]))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def check_samp(arr, g, rtol, atol, what="", exclude=None):
    """Compare a full tensor against a strided-sample fixture written by oracle/gen_golden.py:samp.
    exclude: optional fn(flat_index_array) -> bool mask of elements left out of the comparison."""
    a = np.asarray(arr, dtype=np.float32).ravel()
    s = a[::g["stride"]]
    if exclude is not None:
        keep = ~exclude(np.arange(a.size)[::g["stride"]])
        np.testing.assert_allclose(s[keep], g["sample"][keep], rtol=rtol, atol=atol, err_msg=what)
        return
    np.testing.assert_allclose(s, g["sample"], rtol=rtol, atol=atol, err_msg=what)
    scale = max(g["abssum"], 1e-30)
    assert abs(float(a.astype(np.float64).sum()) - g["sum"]) <= max(rtol * scale * 4, atol * a.size), what
