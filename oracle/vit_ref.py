"""Oracle: ViT backbone forward (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional fp32 restatement of reference ``semilearn/nets/vit/vit.py``:
  PatchEmbed.forward :39-44   Attention.forward :91-107   Mlp.forward :69-75
  Block.forward :163-166      VisionTransformer.extract :277-283 / forward :285-306
DropPath (timm semantics, vit.py:148,161) is made reproducible by INJECTING the
per-sample scale (0 or 1/keep): ``droppath[depth, 2, B]``; ``None`` == eval mode.
Parameters: dict keyed by the reference's ``named_parameters()`` names.
"""
import math

import torch


class VitCfg:
    def __init__(self, img_size=32, patch_size=2, embed_dim=384, depth=12, num_heads=6,
                 mlp_ratio=4.0, num_classes=100, drop_path_rate=0.2, in_chans=3):
        self.img_size, self.patch_size, self.embed_dim = img_size, patch_size, embed_dim
        self.depth, self.num_heads, self.mlp_ratio = depth, num_heads, mlp_ratio
        self.num_classes, self.drop_path_rate, self.in_chans = num_classes, drop_path_rate, in_chans

    @property
    def grid(self):
        return self.img_size // self.patch_size

    @property
    def num_tokens(self):
        return self.grid * self.grid + 1

    @property
    def hidden(self):
        return int(self.embed_dim * self.mlp_ratio)


VIT_SMALL_P2_32 = dict(img_size=32, patch_size=2, embed_dim=384, depth=12, num_heads=6, drop_path_rate=0.2)
VIT_SMALL_P16_224 = dict(img_size=224, patch_size=16, embed_dim=384, depth=12, num_heads=6, drop_path_rate=0.2)   # vit.py:358-371
VIT_SMALL_P16_224 = dict(img_size=224, patch_size=16, embed_dim=384, depth=12, num_heads=6, drop_path_rate=0.2)
VIT_BASE_P16_96 = dict(img_size=96, patch_size=16, embed_dim=768, depth=12, num_heads=12, drop_path_rate=0.2)      # vit.py:374-390
VIT_TINY_TEST = dict(img_size=8, patch_size=2, embed_dim=128, depth=2, num_heads=2, drop_path_rate=0.2)


def param_shapes(cfg):
    """Names / shapes in the reference's named_parameters() order (vit.py:228-275)."""
    D, H, p, C = cfg.embed_dim, cfg.hidden, cfg.patch_size, cfg.num_classes
    out = [("cls_token", (1, 1, D)), ("pos_embed", (1, cfg.num_tokens, D)),
           ("patch_embed.proj.weight", (D, cfg.in_chans, p, p)), ("patch_embed.proj.bias", (D,))]
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        out += [(b + "norm1.weight", (D,)), (b + "norm1.bias", (D,)),
                (b + "attn.qkv.weight", (3 * D, D)), (b + "attn.qkv.bias", (3 * D,)),
                (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,)),
                (b + "norm2.weight", (D,)), (b + "norm2.bias", (D,)),
                (b + "mlp.fc1.weight", (H, D)), (b + "mlp.fc1.bias", (H,)),
                (b + "mlp.fc2.weight", (D, H)), (b + "mlp.fc2.bias", (D,))]
    out += [("norm.weight", (D,)), ("norm.bias", (D,)), ("head.weight", (C, D)), ("head.bias", (C,))]
    return out


def drop_path_probs(cfg):
    """vit.py:247-249: linspace(0, rate, depth)."""
    return [float(x) for x in torch.linspace(0, cfg.drop_path_rate, cfg.depth)]


def _ln(x, w, b, eps=1e-6):               # vit.py:222 eps=1e-6
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def _gelu(x):                             # nn.GELU() default == exact erf form
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def patchify(x, p):
    """[B,C,H,W] -> [B, gh*gw, C*p*p] with (c, i, j) minor order == Conv2d weight.flatten(1)."""
    B, C, H, W = x.shape
    gh, gw = H // p, W // p
    x = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * p * p)


def vit_forward(P, x, cfg, droppath=None, return_tokens=False, aten_ops=False):
    """Returns dict(logits [B,C], feat [B,D]).  droppath: None or [depth,2,B] scales.
    aten_ops: LayerNorm / GELU through the ATen kernels the reference's nn.LayerNorm / nn.GELU modules call (vit.py:63, :222) instead of the
    op-by-op restatement above -- same values to fp32 round-off, but one fused CPU kernel each instead of 5-6 elementwise passes with their
    autograd temporaries: the restatement ran 3.5x slower than the reference on the same cores (tools/ref_cpu_step.py), which would have
    made bench.py's cpu_baseline slow by construction.  bench.py times aten_ops=True."""
    if aten_ops:
        import torch.nn.functional as F
        ln = lambda t_, w, b_: F.layer_norm(t_, (t_.shape[-1],), w, b_, 1e-6)   # noqa: E731
        gelu, lin = F.gelu, F.linear
    else:
        ln, gelu = _ln, _gelu
        lin = lambda t_, w, b_: t_ @ w.t() + b_                                  # noqa: E731
    B = x.shape[0]
    D, nh = cfg.embed_dim, cfg.num_heads
    hd = D // nh
    t = patchify(x, cfg.patch_size) @ P["patch_embed.proj.weight"].reshape(D, -1).t() \
        + P["patch_embed.proj.bias"]                                     # vit.py:40-42
    t = torch.cat((P["cls_token"].expand(B, -1, -1), t), dim=1) + P["pos_embed"]   # :278-279
    N = t.shape[1]
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        h = ln(t, P[b + "norm1.weight"], P[b + "norm1.bias"])
        qkv = lin(h, P[b + "attn.qkv.weight"], P[b + "attn.qkv.bias"]) \
            .reshape(B, N, 3, nh, hd).permute(2, 0, 3, 1, 4)            # :93-98
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = torch.softmax((q @ k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)   # :100-101
        o = (a @ v).transpose(1, 2).reshape(B, N, D)                    # :104
        o = lin(o, P[b + "attn.proj.weight"], P[b + "attn.proj.bias"])
        if droppath is not None:
            o = o * droppath[i, 0].view(B, 1, 1)
        t = t + o                                                        # :164
        h = ln(t, P[b + "norm2.weight"], P[b + "norm2.bias"])
        h = gelu(lin(h, P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"]))
        h = lin(h, P[b + "mlp.fc2.weight"], P[b + "mlp.fc2.bias"])
        if droppath is not None:
            h = h * droppath[i, 1].view(B, 1, 1)
        t = t + h                                                        # :165
    t = ln(t, P["norm.weight"], P["norm.bias"])                          # :282
    feat = t[:, 0]                                                       # :299 global_pool='token'
    logits = feat @ P["head.weight"].t() + P["head.bias"]                # :304
    out = {"logits": logits, "feat": feat}
    if return_tokens:
        out["tokens"] = t
    return out


def vit_forward_engine_rounding(P, x, cfg, droppath=None):
    """The SAME network with the ENGINE's operand rounding points restated (fp32 everywhere else) -- not the reference's arithmetic, a CPU
    model of libsrhip's inference path (semireward_amd/csrc/attn_block.hip, mlp_fused.hip): bf16 GEMM operands (LayerNorm outputs, the four
    weight matrices of a block, q / k / v, the softmax probabilities -- normalised by the sum of the UNROUNDED ones --, the attention output
    with its DropPath factor applied before the rounding, the GELU output with its DropPath factor), fp32 accumulation, fp32 residual stream /
    LayerNorm / softmax, fp32 patch embedding and classifier head.  Used to separate the two reasons the engine can deviate from the fp32
    reference: operand ROUNDING (this function shares it) and kernel ERROR (summation order, the exp2 / polynomial-GELU approximations, bugs:
    this function has none of the engine's code).  It predicts the STATISTICS of the engine's deviation, not its numbers: the direction an
    operand rounds in depends on its value to ~1e-4 relative, so two implementations of the same rounding points draw nearly independent
    noise from the second block on (tools/rounding_model_probe.py).  tests/test_gpu_srflexmatch.py compares the engine with both."""
    r = lambda t_: t_.to(torch.bfloat16).to(torch.float32)   # noqa: E731
    B = x.shape[0]
    D, nh = cfg.embed_dim, cfg.num_heads
    hd = D // nh
    t = patchify(x, cfg.patch_size) @ P["patch_embed.proj.weight"].reshape(D, -1).t() + P["patch_embed.proj.bias"]
    t = torch.cat((P["cls_token"].expand(B, -1, -1), t), dim=1) + P["pos_embed"]
    N = t.shape[1]
    one = torch.ones(B)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s1 = (droppath[i, 0] if droppath is not None else one).view(B, 1, 1)
        s2 = (droppath[i, 1] if droppath is not None else one).view(B, 1, 1)
        h = r(_ln(t, P[b + "norm1.weight"], P[b + "norm1.bias"]))
        qkv = r(h @ r(P[b + "attn.qkv.weight"]).t() + P[b + "attn.qkv.bias"]).reshape(B, N, 3, nh, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        sc = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
        p = torch.exp(sc - sc.max(dim=-1, keepdim=True).values)
        o = (r(p) @ v) / p.sum(dim=-1, keepdim=True)
        o = r(o.transpose(1, 2).reshape(B, N, D) * s1)
        t = t + o @ r(P[b + "attn.proj.weight"]).t() + s1 * P[b + "attn.proj.bias"]
        h = r(_ln(t, P[b + "norm2.weight"], P[b + "norm2.bias"]))
        h = r(_gelu(h @ r(P[b + "mlp.fc1.weight"]).t() + P[b + "mlp.fc1.bias"]) * s2)
        t = t + h @ r(P[b + "mlp.fc2.weight"]).t() + s2 * P[b + "mlp.fc2.bias"]
    t = _ln(t, P["norm.weight"], P["norm.bias"])
    feat = t[:, 0]
    return {"logits": feat @ P["head.weight"].t() + P["head.bias"], "feat": feat}


def flops_per_image(cfg):
    """depth*N*(24 D^2 + 4 N D) + patch-embed + head (forward)."""
    D, N = cfg.embed_dim, cfg.num_tokens
    pe = 2 * (N - 1) * (cfg.in_chans * cfg.patch_size ** 2) * D
    return cfg.depth * N * (24 * D * D + 4 * N * D) + pe + 2 * D * cfg.num_classes
