"""Headless import of the upstream reference (ONLY usable in the build container).

TEST INFRASTRUCTURE -- never imported by the product path, never run on the GPU box
(/root/reference does not exist there).  Used solely by oracle/gen_golden.py to
produce the committed fixtures under tests/golden/ and to cross-check the oracle.

The reference package cannot be imported normally here (torchvision/timm/ruamel/...
are absent), so the package __init__ side effects are skipped by registering
path-only stub packages, as described in SURVEY.md Appendix B.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("SEMIREWARD_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "semilearn"))


class _DropPath(nn.Module):
    """timm.models.layers.DropPath semantics, with an optional injected per-sample
    scale (value already divided by keep-prob) so that golden vectors are reproducible."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = p
        self.injected = None  # tensor [B] of scales (0 or 1/keep)

    def forward(self, x):
        if self.injected is not None:
            return x * self.injected.view(-1, *([1] * (x.ndim - 1))).to(x.dtype)
        if self.p == 0.0 or not self.training:
            return x
        k = 1 - self.p
        r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(k).div_(k)
        return x * r


def _stub(name, path=None, **kw):
    m = types.ModuleType(name)
    if path:
        m.__path__ = [path]
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


_done = False


def install():
    global _done
    if _done:
        return
    sys.dont_write_bytecode = True
    for n in ("wandb", "aim", "ruamel", "ruamel.yaml", "timm", "timm.models"):
        _stub(n)
    _stub("torch.utils.tensorboard", SummaryWriter=object)
    _stub("timm.models.layers", DropPath=_DropPath,
          to_2tuple=lambda v: v if isinstance(v, (tuple, list)) else (v, v))
    S = REF + "/semilearn"
    _stub("semilearn", S)
    _stub("semilearn.nets", S + "/nets")
    _stub("semilearn.nets.vit", S + "/nets/vit")
    _stub("semilearn.nets.wrn", S + "/nets/wrn")
    _stub("semilearn.nets.bert", S + "/nets/bert")
    _stub("semilearn.nets.wave2vecv2", S + "/nets/wave2vecv2")
    _stub("semilearn.nets.hubert", S + "/nets/hubert")
    _stub("semilearn.algorithms", S + "/algorithms")
    for a in ("srflexmatch", "srfixmatch", "srpseudolabel", "srsoftmatch", "srfreematch",
              "flexmatch", "freematch", "softmatch"):
        _stub("semilearn.algorithms." + a, S + "/algorithms/" + a)
    _stub("semilearn.datasets", get_collactor=None, name2sampler={},
          DistributedSampler=type("DS", (), {}))
    core = _stub("semilearn.core", S + "/core")
    core.AlgorithmBase = importlib.import_module("semilearn.core.algorithmbase").AlgorithmBase
    torch.Tensor.cuda = lambda self, *a, **k: self  # CPU shim: reference calls .cuda(gpu)
    _done = True


def mod(name):
    install()
    return importlib.import_module(name)
