"""Oracle: Wav2Vec2 classification backbone of the usb_audio configs (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional torch-CPU fp32 restatement of ``semilearn/nets/wave2vecv2/wave2vecv2.py`` (ClassificationWave2Vec.extract :42-49 / forward
:23-40: HF ``Wav2Vec2Model`` on the raw waveform, no attention mask -> Dropout(0.1) on last_hidden_state -> mean over the frames ->
Linear / GELU / Linear) and of the third-party encoder it calls: ``transformers.Wav2Vec2Model`` (requirement ``transformers>=4.30.0``,
unpinned; the build container has 5.15.0), checkpoint family ``facebook/wav2vec2-base-960h``: feat_extract_norm 'group',
do_stable_layer_norm False (post-LN encoder layers), conv_bias False.  Published algorithm (Baevski et al. 2020; HF modeling_wav2vec2.py):
  feature encoder : 7 x Conv1d(k, stride, no bias) over the waveform, GroupNorm(C groups == per-channel over time) after the first, GELU
  projection      : LayerNorm(conv_dim) -> Linear -> Dropout(feat_proj_dropout)
  SpecAugment     : train mode only: ~mask_time_prob of the frames, in spans of mask_time_length (>= min_masks spans per clip), are REPLACED by
                    the learned ``masked_spec_embed`` vector
  encoder         : x + GELU(grouped Conv1d(k=128, pad 64, groups 16, weight-normed over dim 2)(x)[..., :-1]) -> LayerNorm -> Dropout;
                    layers: x = LN(x + drop(attn(x)));  x = LN(x + drop(W2 drop(GELU(W1 x))));  LayerDrop skips a whole layer in train mode
Parameter names are the reference module's state_dict keys (``model.`` + HF names, ``classifier.0/2``).

Randomness (dropout masks, SpecAugment spans, LayerDrop decisions) is drawn by torch / numpy global RNGs in the reference, which nothing else
can reproduce: oracle and HIP engine take them as INPUTS (counter-based dropout generator of oracle/bert_ref.py; ``spec_augment_mask`` /
explicit skip flags below) and the golden generator injects the same values into the HF modules.
"""
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

from .bert_ref import keep_mask

W2vCfg = namedtuple("W2vCfg", "hidden layers heads inter conv_dim conv_kernel conv_stride pos_k pos_groups num_classes "
                              "p_hidden p_act p_attn p_featproj p_head layerdrop mask_time_prob mask_time_length mask_time_min_masks",
                    defaults=(0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.05, 10, 2))        # facebook/wav2vec2-base-960h config.json
W2V_BASE = dict(hidden=768, layers=12, heads=12, inter=3072, conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2),
                conv_stride=(5, 2, 2, 2, 2, 2, 2), pos_k=128, pos_groups=16)
W2V_TINY_TEST = dict(hidden=128, layers=2, heads=2, inter=256, conv_dim=(128, 128, 128), conv_kernel=(10, 3, 2), conv_stride=(5, 2, 2),
                     pos_k=16, pos_groups=4)
LN_EPS = 1e-5
SITE_EMB, SITE_HEAD, SITE_FEATPROJ = 0x7FFFFFF0, 0x7FFFFFF1, 0x7FFFFFF2
SITE_PROBS, SITE_ATTN_OUT, SITE_FFN_OUT, SITE_ACT = 0, 1, 2, 3


def frames(cfg, samples):
    """Frame counts after every conv layer."""
    out, t = [], samples
    for k, s in zip(cfg.conv_kernel, cfg.conv_stride):
        t = (t - k) // s + 1
        out.append(t)
    return out


def param_shapes(cfg):
    """named_parameters() order of the reference ClassificationWave2Vec."""
    D, I, C = cfg.hidden, cfg.inter, cfg.conv_dim
    s = [("model.masked_spec_embed", (D,))]
    for i, (c, k) in enumerate(zip(C, cfg.conv_kernel)):
        s.append(("model.feature_extractor.conv_layers.%d.conv.weight" % i, (c, C[i - 1] if i else 1, k)))
        if i == 0:
            s += [("model.feature_extractor.conv_layers.0.layer_norm.weight", (c,)), ("model.feature_extractor.conv_layers.0.layer_norm.bias", (c,))]
    s += [("model.feature_projection.layer_norm.weight", (C[-1],)), ("model.feature_projection.layer_norm.bias", (C[-1],)),
          ("model.feature_projection.projection.weight", (D, C[-1])), ("model.feature_projection.projection.bias", (D,)),
          ("model.encoder.pos_conv_embed.conv.bias", (D,)),
          ("model.encoder.pos_conv_embed.conv.parametrizations.weight.original0", (1, 1, cfg.pos_k)),
          ("model.encoder.pos_conv_embed.conv.parametrizations.weight.original1", (D, D // cfg.pos_groups, cfg.pos_k)),
          ("model.encoder.layer_norm.weight", (D,)), ("model.encoder.layer_norm.bias", (D,))]
    for i in range(cfg.layers):
        p = "model.encoder.layers.%d." % i
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s += [(p + "attention.%s.weight" % n, (D, D)), (p + "attention.%s.bias" % n, (D,))]
        s += [(p + "layer_norm.weight", (D,)), (p + "layer_norm.bias", (D,)),
              (p + "feed_forward.intermediate_dense.weight", (I, D)), (p + "feed_forward.intermediate_dense.bias", (I,)),
              (p + "feed_forward.output_dense.weight", (D, I)), (p + "feed_forward.output_dense.bias", (D,)),
              (p + "final_layer_norm.weight", (D,)), (p + "final_layer_norm.bias", (D,))]
    s += [("classifier.0.weight", (D, D)), ("classifier.0.bias", (D,)), ("classifier.2.weight", (cfg.num_classes, D)),
          ("classifier.2.bias", (cfg.num_classes,))]
    return s


def spec_augment_mask(seed, B, T, mask_prob, mask_length, min_masks):
    """_compute_mask_indices (modeling_wav2vec2.py:101-233, no attention mask) with an explicit numpy Generator: bool [B, T]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    eps = float(rng.random())
    n = max(int(mask_prob * T / mask_length + eps), min_masks)
    if n * mask_length > T:
        n = T // mask_length
    if T - (mask_length - 1) < n:
        n = max(T - (mask_length - 1), 0)
    m = np.zeros((B, T), dtype=bool)
    if n == 0:
        return m
    for b in range(B):
        starts = rng.choice(np.arange(T - (mask_length - 1)), n, replace=False)
        idx = np.minimum((starts[:, None] + np.arange(mask_length)[None]).reshape(-1), T - 1)
        m[b, idx] = True
    return m


def frame_pitch(T):
    """Row pitch of the engine's [clip, frame, channel] activations at the encoder: >= T + 1, multiple of 8 (nets/wave2vec.py geometry)."""
    return (T + 1 + 7) // 8 * 8


def pitched_keep(seed, site, shape, p, pitch):
    """keep mask of a [B, T, X] (or [B, H, T, T]) tensor whose elements are indexed as in the engine's frame-pitched layout [B, pitch, X]
    ([B, H, pitch, pitch]): the masks are INPUTS of the model, any fixed assignment of generator outputs to elements is as good as another."""
    if pitch is None:
        return keep_mask(seed, site, shape, p)
    if len(shape) == 3:
        return keep_mask(seed, site, (shape[0], pitch, shape[2]), p)[:, :shape[1]]
    return keep_mask(seed, site, (shape[0], shape[1], pitch, pitch), p)[:, :, :shape[2], :shape[3]]


def _drop(x, seed, site, p, pitch=None):
    if seed is None or p <= 0.0:
        return x
    return x * torch.from_numpy(np.ascontiguousarray(pitched_keep(seed, site, tuple(x.shape), p, pitch)).astype(np.float32) / np.float32(1.0 - p))


def pos_conv_weight(P):
    """nn.utils.parametrizations.weight_norm(conv, dim=2): w = g * v / ||v||, the norm over every dim except 2 (per kernel tap)."""
    g = P["model.encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v = P["model.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    return g * v / v.norm(dim=(0, 1), keepdim=True)


def w2v_forward(P, wave, cfg, seed=None, spec_mask=None, skip=None, pitched=True):
    """wave fp32 [B, samples].  Eval mode: seed = spec_mask = skip = None.  Train mode: seed = 64-bit dropout seed of the call,
    spec_mask bool [B, T] (SpecAugment), skip = per-layer LayerDrop flags; pitched: dropout masks indexed in the engine's frame-pitched
    layout (pitched_keep).  Returns dict(logits, feat, hidden)."""
    D, H = cfg.hidden, cfg.heads
    x = wave[:, None]
    for i, s in enumerate(cfg.conv_stride):
        x = F.conv1d(x, P["model.feature_extractor.conv_layers.%d.conv.weight" % i], stride=s)
        if i == 0:
            x = F.group_norm(x, x.shape[1], P["model.feature_extractor.conv_layers.0.layer_norm.weight"],
                             P["model.feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5)
        x = F.gelu(x)
    x = x.transpose(1, 2)                                                                         # [B, T, conv_dim]
    B, T, _ = x.shape
    pt = frame_pitch(T) if pitched else None
    x = F.layer_norm(x, (x.shape[-1],), P["model.feature_projection.layer_norm.weight"], P["model.feature_projection.layer_norm.bias"], LN_EPS)
    x = F.linear(x, P["model.feature_projection.projection.weight"], P["model.feature_projection.projection.bias"])
    x = _drop(x, seed, SITE_FEATPROJ, cfg.p_featproj, pt)
    if spec_mask is not None:
        x = torch.where(torch.as_tensor(spec_mask)[:, :, None], P["model.masked_spec_embed"], x)
    pc = F.conv1d(x.transpose(1, 2), pos_conv_weight(P), P["model.encoder.pos_conv_embed.conv.bias"], padding=cfg.pos_k // 2,
                  groups=cfg.pos_groups)
    if cfg.pos_k % 2 == 0:
        pc = pc[:, :, :-1]
    x = x + F.gelu(pc).transpose(1, 2)
    x = F.layer_norm(x, (D,), P["model.encoder.layer_norm.weight"], P["model.encoder.layer_norm.bias"], LN_EPS)
    x = _drop(x, seed, SITE_EMB, cfg.p_hidden, pt)
    for i in range(cfg.layers):
        if skip is not None and skip[i]:
            continue
        q_ = "model.encoder.layers.%d." % i
        lin = lambda t, n: F.linear(t, P[q_ + n + ".weight"], P[q_ + n + ".bias"])   # noqa: E731
        heads = lambda t: t.view(B, T, H, D // H).transpose(1, 2)   # noqa: E731
        q, k, v = heads(lin(x, "attention.q_proj")), heads(lin(x, "attention.k_proj")), heads(lin(x, "attention.v_proj"))
        probs = _drop(torch.softmax(q @ k.transpose(-1, -2) * (D // H) ** -0.5, dim=-1), seed, 4 * i + SITE_PROBS, cfg.p_attn, pt)
        ctx = (probs @ v).transpose(1, 2).reshape(B, T, D)
        x = F.layer_norm(x + _drop(lin(ctx, "attention.out_proj"), seed, 4 * i + SITE_ATTN_OUT, cfg.p_hidden, pt), (D,),
                         P[q_ + "layer_norm.weight"], P[q_ + "layer_norm.bias"], LN_EPS)
        h = _drop(F.gelu(lin(x, "feed_forward.intermediate_dense")), seed, 4 * i + SITE_ACT, cfg.p_act, pt)
        x = F.layer_norm(x + _drop(lin(h, "feed_forward.output_dense"), seed, 4 * i + SITE_FFN_OUT, cfg.p_hidden, pt), (D,),
                         P[q_ + "final_layer_norm.weight"], P[q_ + "final_layer_norm.bias"], LN_EPS)
    feat = _drop(x, seed, SITE_HEAD, cfg.p_head, pt).mean(1)                                          # wave2vecv2.py:47-48
    hcls = F.gelu(F.linear(feat, P["classifier.0.weight"], P["classifier.0.bias"]))
    logits = F.linear(hcls, P["classifier.2.weight"], P["classifier.2.bias"])
    return dict(logits=logits, feat=feat, hidden=x)


def synth_params(cfg, seed):
    """HF-style magnitudes (kaiming conv filters, normal(0, 0.02)-like dense weights, LayerNorm 1 +- 0.1) with every gradient path alive."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for n, shp in param_shapes(cfg):
        if n.endswith("layer_norm.weight") or n.endswith("LayerNorm.weight"):
            out[n] = (1.0 + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif n.endswith("original0"):
            out[n] = (0.5 + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif n == "model.masked_spec_embed":
            out[n] = rng.random(shp).astype(np.float32)
        elif len(shp) == 1:
            out[n] = (0.02 * rng.standard_normal(shp)).astype(np.float32)
        elif len(shp) == 3:
            out[n] = (rng.standard_normal(shp) * np.sqrt(2.0 / (shp[1] * shp[2]))).astype(np.float32)
        elif n.startswith("classifier"):
            out[n] = (rng.standard_normal(shp) / np.sqrt(shp[1])).astype(np.float32)
        else:
            out[n] = (0.05 * rng.standard_normal(shp)).astype(np.float32)
    return out
