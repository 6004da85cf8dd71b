"""The whole path on one GPU: uint8 dataset in HBM -> device-side weak / strong views -> SRFlexMatch steps (crossing start_timing, so the
K-pass scoring loop and the rewarder updates run) -> evaluate().  A synthetic 10-class task is learned to > 90 % in 100 steps."""
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_train_synthetic_cifar_learns(monkeypatch):
    sys.path.insert(0, "examples")
    import train_synthetic_cifar as ex
    monkeypatch.setattr(sys, "argv", ["train_synthetic_cifar.py", "--steps", "100"])
    ev = ex.main()
    assert ev["eval/top-1-acc"] > 0.9 and ev["eval/loss"] < 1.0
