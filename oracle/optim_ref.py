"""Oracle: optimizer / scheduler semantics (TEST INFRASTRUCTURE, see oracle/__init__.py).

  get_optimizer                    semilearn/core/utils/build.py:193-224
  param_groups_layer_decay         semilearn/nets/utils.py:143-204 (+ group_matcher vit.py:311-320)
  get_cosine_schedule_with_warmup  semilearn/core/utils/build.py:227-251
  ParamUpdateHook.after_train_step semilearn/core/hooks/param_update.py:21-45
  EMA.update                       semilearn/core/utils/misc.py:152-155
"""
import math
import re

import torch


def vit_layer_id(name, depth):
    """group_matcher (vit.py:311-320) + group_with_matcher(reverse=True): stem -> 0,
    blocks.i -> i+1, final ``norm`` joins the previous group (MATCH_PREV_GROUP) == depth,
    anything unmatched (head) -> layer_max == depth+1 (nets/utils.py:178)."""
    if name in ("cls_token", "pos_embed") or name.startswith("patch_embed"):
        return 0
    m = re.match(r"^blocks\.(\d+)", name)
    if m:
        return int(m.group(1)) + 1
    if name.startswith("norm"):
        return depth
    return depth + 1


def vit_param_hparams(names_shapes, depth, lr, weight_decay, layer_decay,
                      no_weight_decay=("pos_embed", "cls_token")):
    """Per-tensor (lr, weight_decay) exactly as the 28 param groups of SURVEY A.11."""
    layer_max = depth + 1
    out = {}
    for name, shape in names_shapes:
        lid = vit_layer_id(name, depth)
        scale = layer_decay ** (layer_max - lid)
        wd = 0.0 if (len(shape) == 1 or name in no_weight_decay) else weight_decay
        out[name] = (scale * lr, wd)
    return out


def bert_param_hparams(names_shapes, layers, lr, weight_decay, layer_decay):
    """ClassificationBert.group_matcher (bert.py:54-56: stem = ^bert.embeddings, blocks = ^bert.encoder.layer.(\\d+)) through
    group_with_matcher(reverse=True) (nets/utils.py:208-270): embeddings -> 0, encoder layer i -> i + 1, everything unmatched (pooler,
    classifier) -> layer_max = layers + 1; lr scale layer_decay ** (layer_max - id), no weight decay for 1-D tensors (nets/utils.py:170-176).
    The pooler feeds nothing on this path: its .grad stays None and torch.optim.AdamW skips it entirely -> (0, 0)."""
    layer_max = layers + 1
    out = {}
    for name, shape in names_shapes:
        if name.startswith("bert.embeddings"):
            lid = 0
        elif name.startswith("bert.encoder.layer."):
            lid = int(name.split(".")[3]) + 1
        else:
            lid = layer_max
        scale = layer_decay ** (layer_max - lid) if layer_decay != 1.0 else 1.0
        wd = 0.0 if len(shape) == 1 else weight_decay
        out[name] = (0.0, 0.0) if name.startswith("bert.pooler") else (scale * lr, wd)
    return out


def w2v_param_hparams(names_shapes, layers, lr, weight_decay, layer_decay, hubert=False):
    """ClassificationWave2Vec.group_matcher (wave2vecv2.py:51-53: stem = feature_projection | feature_extractor, blocks = encoder.layers.(\\d+));
    ClassificationHubert's (hubert.py:52-54) also puts encoder.pos_conv_embed into the stem.  Unmatched names (masked_spec_embed, encoder
    layer_norm, classifier, [pos_conv_embed]) -> layer_max = layers + 1."""
    layer_max = layers + 1
    out = {}
    for name, shape in names_shapes:
        if name.startswith("model.feature_projection") or name.startswith("model.feature_extractor") or \
                (hubert and name.startswith("model.encoder.pos_conv_embed")):
            lid = 0
        elif name.startswith("model.encoder.layers."):
            lid = int(name.split(".")[3]) + 1
        else:
            lid = layer_max
        scale = layer_decay ** (layer_max - lid) if layer_decay != 1.0 else 1.0
        out[name] = (scale * lr, 0.0 if len(shape) == 1 else weight_decay)
    return out


def cosine_warmup_factor(step, num_training_steps, num_warmup_steps=0, num_cycles=7.0 / 16.0):
    """build.py:237-249.  LambdaLR applies factor(it) at 0-based iteration ``it``."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    t = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, math.cos(math.pi * num_cycles * t))


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor math (decoupled decay first).  step is 1-based.  In place."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def ema_update(shadow, p, decay):
    """misc.py:152-155: shadow = (1-m)*p + m*shadow."""
    return (1.0 - decay) * p + decay * shadow
