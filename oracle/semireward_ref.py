"""Oracle: Rewarder / Generator / SR losses (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates reference ``semilearn/algorithms/semireward/semireward.py``:
  Generator.forward   :21-24      Rewarder.forward :52-72
  cosine_similarity_n :130-139    label_dim        :147-148
and the SR update block ``semilearn/algorithms/srflexmatch/srflexmatch.py:180-208``.
Parameters are plain dicts keyed by the reference's ``named_parameters()`` names.
"""
import math

import numpy as np
import torch

REWARDER_KEYS = (
    "feature_fc.weight", "feature_fc.bias", "feature_norm.weight", "feature_norm.bias",
    "label_embedding.weight", "label_norm.weight", "label_norm.bias",
    "cross_attention_fc.weight", "cross_attention_fc.bias",
    "mlp_fc1.weight", "mlp_fc1.bias", "mlp_fc2.weight", "mlp_fc2.bias",
    "ffn_fc1.weight", "ffn_fc1.bias", "ffn_fc2.weight", "ffn_fc2.bias",
)
GENERATOR_KEYS = tuple(f"fc_layers.{i}.{w}" for i in (0, 2, 4, 6) for w in ("weight", "bias"))


def label_dim(num_classes, default_dim=100):
    """semireward.py:147-148."""
    return int(max(default_dim, num_classes))


def rewarder_shapes(feature_dim, num_classes, emb=128):
    L = label_dim(num_classes)
    return {
        "feature_fc.weight": (128, feature_dim), "feature_fc.bias": (128,),
        "feature_norm.weight": (128,), "feature_norm.bias": (128,),
        "label_embedding.weight": (L, emb), "label_norm.weight": (emb,), "label_norm.bias": (emb,),
        "cross_attention_fc.weight": (1, 128), "cross_attention_fc.bias": (1,),
        "mlp_fc1.weight": (256, 128), "mlp_fc1.bias": (256,),
        "mlp_fc2.weight": (128, 256), "mlp_fc2.bias": (128,),
        "ffn_fc1.weight": (64, 128), "ffn_fc1.bias": (64,),
        "ffn_fc2.weight": (1, 64), "ffn_fc2.bias": (1,),
    }


def generator_shapes(feature_dim):
    dims = [feature_dim, 256, 128, 64, 1]
    out = {}
    for li, i in enumerate((0, 2, 4, 6)):
        out[f"fc_layers.{i}.weight"] = (dims[li + 1], dims[li])
        out[f"fc_layers.{i}.bias"] = (dims[li + 1],)
    return out


def _ln(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def rewarder_forward(p, features, labels):
    """semireward.py:52-72.  features [B,F] f32, labels [B] i64 -> [B,1]."""
    h = _ln(features @ p["feature_fc.weight"].t() + p["feature_fc.bias"],
            p["feature_norm.weight"], p["feature_norm.bias"])
    e = _ln(p["label_embedding.weight"][labels], p["label_norm.weight"], p["label_norm.bias"])
    z = torch.cat((h, e), dim=0)                                   # [2B,128]
    s = z @ p["cross_attention_fc.weight"].t() + p["cross_attention_fc.bias"]  # [2B,1]
    a = torch.softmax(s, dim=0)                                    # over the BATCH rows (:61)
    c = (a * z).sum(dim=0)                                         # [128]
    u = c.unsqueeze(0) + e
    m1 = torch.relu(u @ p["mlp_fc1.weight"].t() + p["mlp_fc1.bias"])
    m2 = m1 @ p["mlp_fc2.weight"].t() + p["mlp_fc2.bias"]
    f1 = torch.relu(m2 @ p["ffn_fc1.weight"].t() + p["ffn_fc1.bias"])
    return torch.sigmoid(f1 @ p["ffn_fc2.weight"].t() + p["ffn_fc2.bias"])


def generator_forward(p, x):
    """semireward.py:21-24: F->256->128->64->1, ReLU after every layer (incl. the last)."""
    for i in (0, 2, 4, 6):
        x = torch.relu(x @ p[f"fc_layers.{i}.weight"].t() + p[f"fc_layers.{i}.bias"])
    return x


def generated_labels(p, x):
    """srflexmatch.py:158-159 -- ``generator(feats).long()`` (truncation toward zero)."""
    return generator_forward(p, x).to(torch.int64).squeeze(1)


def cosine_target(gen_labels, ref_labels, num_classes):
    """srflexmatch.py:180-182 / :195-197 + semireward.py:130-139 on one-hots:
    (cos+1)/2 == 1.0 where the labels agree, 0.5 otherwise.  Computed the long way."""
    a = torch.nn.functional.one_hot(gen_labels, num_classes).float()
    b = torch.nn.functional.one_hot(ref_labels, num_classes).float()
    dot = (a * b).sum(-1)
    den = torch.clamp(a.norm(dim=-1) * b.norm(dim=-1), min=1e-8)
    return ((dot / den + 1) / 2).view(-1, 1)


def sr_losses(reward, target):
    """srflexmatch.py:183-184: generator_loss = MSE(r,1), rewarder_loss = MSE(r,t)."""
    return ((reward - 1.0) ** 2).mean(), ((reward - target) ** 2).mean()


def rewarder_update_grads(p, features, gen_labels, target):
    """Gradient that the two ``backward()`` calls (srflexmatch.py:189-190 / :204-205)
    accumulate into the rewarder: d(MSE(r,1) + MSE(r,t))/dtheta.  Returns (reward, grads)."""
    q = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    r = rewarder_forward(q, features, gen_labels)
    lg, lr_ = sr_losses(r, target)
    (lg + lr_).backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in q.items()}
    return r.detach(), grads, float(lg.detach()), float(lr_.detach())


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (srflexmatch.py:54: ``Adam(rewarder.parameters(), lr=sr_lr)``),
    default betas/eps, no weight decay, no amsgrad.  ``step`` is 1-based.  In place."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k in p:
        m[k].mul_(beta1).add_(g[k], alpha=1 - beta1)
        v[k].mul_(beta2).addcmul_(g[k], g[k], value=1 - beta2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p[k].addcdiv_(m[k], denom, value=-lr / bc1)


def reward_mask2(reward):
    """srflexmatch.py:100-101: mask2 = (reward >= reward.mean()).float(), local-batch fp32 mean."""
    r = reward.reshape(-1)
    return (r >= r.mean()).to(torch.float32)


def to_numpy(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}
