"""Oracle: WideResNet backbone of the classic_cv configs (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional torch-CPU fp32 restatement of ``semilearn/nets/wrn/wrn.py``:
  BasicBlock.forward :45-55 (pre-activation residual block, 1x1 shortcut when the shape changes, activate_before_residual only
  for block1), WideResNet.extract :131-137 / forward :112-129, BatchNorm2d(momentum 0.001; eps 1e-5, final bn1 eps 1e-3),
  LeakyReLU(0.1).  Parameter / buffer names are the reference's state_dict keys.
BatchNorm follows torch semantics: training mode normalises with the statistics of THIS forward's batch and (unless frozen,
core/utils/misc.py:105-129 Bn_Controller) moves running_mean / running_var (unbiased) with momentum 0.001.
"""
from collections import OrderedDict, namedtuple

import torch
import torch.nn.functional as F

WrnCfg = namedtuple("WrnCfg", "num_classes depth widen first_stride")
WRN_28_2 = dict(depth=28, widen=2, first_stride=1)
WRN_TINY_TEST = dict(depth=10, widen=2, first_stride=1)          # one BasicBlock per group: every code path, 8x8 images in the tests
MOMENTUM, SLOPE = 0.001, 0.1


def channels(cfg):
    return [16, 16 * cfg.widen, 32 * cfg.widen, 64 * cfg.widen]


def blocks(cfg):
    """[(prefix, cin, cout, stride, activate_before_residual)] in forward order."""
    ch, n = channels(cfg), (cfg.depth - 4) // 6
    out = []
    for g, stride in enumerate((cfg.first_stride, 2, 2)):
        for i in range(n):
            out.append(("block%d.layer.%d." % (g + 1, i), ch[g] if i == 0 else ch[g + 1], ch[g + 1], stride if i == 0 else 1, g == 0))
    return out


def param_shapes(cfg):
    """Reference ``named_parameters()`` order (wrn.py:75-101)."""
    ch = channels(cfg)
    s = [("conv1.weight", (ch[0], 3, 3, 3)), ("conv1.bias", (ch[0],))]
    for p, cin, cout, stride, _ in blocks(cfg):
        s += [(p + "bn1.weight", (cin,)), (p + "bn1.bias", (cin,)), (p + "conv1.weight", (cout, cin, 3, 3)),
              (p + "bn2.weight", (cout,)), (p + "bn2.bias", (cout,)), (p + "conv2.weight", (cout, cout, 3, 3))]
        if cin != cout:
            s.append((p + "convShortcut.weight", (cout, cin, 1, 1)))
    s += [("bn1.weight", (ch[3],)), ("bn1.bias", (ch[3],)), ("classifier.weight", (cfg.num_classes, ch[3])),
          ("classifier.bias", (cfg.num_classes,))]
    return s


def bn_names(cfg):
    """[(prefix of the BatchNorm, channels, eps)] in forward order."""
    out = []
    for p, cin, cout, _, _ in blocks(cfg):
        out += [(p + "bn1", cin, 1e-5), (p + "bn2", cout, 1e-5)]
    out.append(("bn1", channels(cfg)[3], 1e-3))
    return out


def init_buffers(cfg):
    B = OrderedDict()
    for n, c, _ in bn_names(cfg):
        B[n + ".running_mean"], B[n + ".running_var"] = torch.zeros(c), torch.ones(c)
    return B


def _bn(x, P, BUF, name, eps, train, update):
    rm, rv = BUF[name + ".running_mean"], BUF[name + ".running_var"]
    if train and not update:                       # Bn_Controller.freeze_bn ... unfreeze_bn: statistics restored afterwards
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, P[name + ".weight"], P[name + ".bias"], train, MOMENTUM, eps)


def _rb(t):
    """bf16 rounding with a straight-through gradient (models the engine's bf16 GEMM operands)."""
    return t + (t.to(torch.bfloat16).float() - t).detach()


def wrn_forward(P, BUF, x, cfg, train=True, update_stats=True, bf16_operands=False):
    """Returns dict(logits [B,C], feat [B, 64*widen]).  BUF (running statistics) is updated in place when train and update_stats.
    bf16_operands: round every convolution's input and filter to bf16 (fp32 accumulation) -- the arithmetic of the HIP engine.  The
    LeakyReLU kink makes the GRADIENT of a deep ReLU-type net sensitive to which side of zero a pre-activation falls, so a bf16
    forward is compared with this variant (tight) and with the fp32 reference (loose); see tests/test_gpu_wrn.py."""
    act = lambda t: F.leaky_relu(t, SLOPE)   # noqa: E731
    if bf16_operands:
        conv2d = lambda a, w, b, s, p: F.conv2d(_rb(a), _rb(w), b, s, p)   # noqa: E731
    else:
        conv2d = F.conv2d
    out = conv2d(x, P["conv1.weight"], P["conv1.bias"], 1, 1)                               # :132
    for p, cin, cout, stride, abr in blocks(cfg):
        equal = cin == cout
        o = act(_bn(out, P, BUF, p + "bn1", 1e-5, train, update_stats))                     # :46-49
        if not equal and abr:
            out = o                                                                         # x = relu1(bn1(x))  (:47)
        h = conv2d(o if equal else out, P[p + "conv1.weight"], None, stride, 1)             # :50 (note: raw x when not abr)
        h = act(_bn(h, P, BUF, p + "bn2", 1e-5, train, update_stats))
        h = conv2d(h, P[p + "conv2.weight"], None, 1, 1)                                    # :53
        sc = out if equal else conv2d(out, P[p + "convShortcut.weight"], None, stride, 0)
        out = sc + h                                                                        # :54
    out = act(_bn(out, P, BUF, "bn1", 1e-3, train, update_stats))                           # :136
    feat = out.mean(dim=(2, 3))                                                             # :121-122
    return {"logits": feat @ P["classifier.weight"].t() + P["classifier.bias"], "feat": feat}


def sgd_nesterov_step(p, g, buf, lr, momentum, wd, first):
    """torch.optim.SGD(nesterov=True) as built by core/utils/build.py:193-224 for optim 'SGD'."""
    with torch.no_grad():
        d = g + wd * p if wd != 0 else g.clone()
        if first:
            buf.copy_(d)
        else:
            buf.mul_(momentum).add_(d)
        p.add_(d + momentum * buf, alpha=-lr)
