"""The yaml contract on the GPU: for every (algorithm, backbone family) pair that occurs in config/SemiReward/**.yaml one config is taken from
tests/golden/sr_configs.json -- the namespace the REFERENCE's get_config() produces for the unchanged yaml -- the algorithm is constructed
from exactly that namespace (+ the four keys the reference's main_worker / set_dataset add at run time: gpu, rank, distributed,
ulb_dest_len) through the registry and get_net_builder, and two steps run at the yaml's own batch layout (small synthetic inputs): one
before start_timing (stage-1 rewarder update) and one in the SR regime (K = sr_decay() data_generator passes)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from semireward_amd import nets, ops
from semireward_amd.algorithms import get_algorithm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "sr_configs.json")))


def _pairs():
    seen, out = set(), []
    for path, c in sorted(G["configs"].items()):
        fam = c["net"]
        if (c["algorithm"], fam) in seen or fam == "vit_tiny_patch2_32":
            continue
        seen.add((c["algorithm"], fam))
        out.append((path, c))
    return out


PAIRS = _pairs()


def _batch(a, alg, rng, B=2):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    if a.net.startswith("bert"):
        def x(n):
            L = int(rng.integers(20, 33))                          # every batch padded to its own longest row (nlp_collactor.py:63-69)
            return {"input_ids": torch.randint(1, 30522, (n, L), generator=g), "attention_mask": torch.ones(n, L, dtype=torch.int64)}
    elif a.net.startswith("hubert") or a.net.startswith("wave2vec"):
        x = lambda n: torch.randn(n, 16000, generator=g)           # noqa: E731  (1 s at 16 kHz instead of max_length_seconds)
    else:
        x = lambda n: torch.randn(n, 3, a.img_size, a.img_size, generator=g)   # noqa: E731
    d = dict(x_lb=x(B), y_lb=torch.randint(0, a.num_classes, (B,), generator=g), x_ulb_w=x(B * a.uratio), x_ulb_s=x(B * a.uratio),
             idx_ulb=torch.from_numpy(rng.permutation(a.ulb_dest_len)[:B * a.uratio].astype(np.int64)))
    return alg.process_batch(**d)                                  # signature-driven filter (algorithmbase.py:287-306)


@pytest.mark.parametrize("path,cfg", PAIRS, ids=["%s+%s" % (c["algorithm"], c["net"]) for _, c in PAIRS])
def test_reference_yaml_namespace_constructs_and_steps(path, cfg):
    assert len(PAIRS) == 14
    a = argparse.Namespace(**cfg)
    a.gpu, a.rank, a.world_size, a.distributed = 0, 0, 1, False    # main_worker (train.py:347-379) on one GPU
    a.ulb_dest_len = 1000                                          # set_dataset (algorithmbase.py:164): size of the unlabeled set
    alg = get_algorithm(a, nets.get_net_builder(a.net, a.net_from_name))
    assert type(alg).__name__.lower() == a.algorithm and alg.num_classes == a.num_classes
    assert alg.optimizer.base_lr == a.lr and alg.start_timing == a.start_timing and alg.N_k == a.N_k
    assert alg.rewarder.flat.numel() > 0 and alg.use_cat == a.use_cat and alg.ema_m == a.ema_m
    rng = np.random.Generator(np.random.PCG64(3))
    alg.model.train()
    for it in (1, a.start_timing + 10):
        alg.it = it
        alg.optimizer.sched_step = it
        before = alg.model.flat.clone()
        rbefore = alg.rewarder.flat.clone()
        alg.out_dict, alg.log_dict = alg.train_step(**_batch(a, alg, rng))
        alg.call_hook("after_train_step")
        torch.cuda.synchronize()
        ops.check_label_errors()
        for k in ("train/sup_loss", "train/unsup_loss", "train/total_loss", "train/util_ratio"):
            assert np.isfinite(float(alg.log_dict[k])), (it, k)
        assert not torch.equal(before, alg.model.flat) and bool(torch.isfinite(alg.model.flat).all())
        assert not torch.equal(rbefore, alg.rewarder.flat)          # it = 1: stage-1 update; it = start_timing + 10: it % N_k == 0
        assert float(alg.model.grad.abs().max()) == 0.0
