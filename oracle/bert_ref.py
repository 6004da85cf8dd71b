"""Oracle: BERT classification backbone of the usb_nlp configs (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional torch-CPU fp32 restatement of ``semilearn/nets/bert/bert.py`` (ClassificationBert.forward :22-48: HF ``BertModel`` ->
Dropout(0.1) on last_hidden_state -> mean over ALL positions, padding included (:37) -> Linear / GELU / Linear) and of the third-party
encoder it calls: ``transformers.BertModel`` (requirement ``transformers>=4.30.0``, unpinned; the build container has 5.15.0).  The
published algorithm (Devlin et al. 2018; HF modeling_bert.py) restated here:
  embeddings = LayerNorm(word[ids] + position[0..L) + token_type[0]), dropout      (word row 0 = [PAD] is a padding_idx row: zero gradient)
  layer:  q,k,v = Linear(x);  probs = softmax(q k^T / sqrt(64) + additive key mask), dropout;  ctx = probs v
          x = LayerNorm(x + dropout(Linear(ctx)));  x = LayerNorm(x + dropout(Linear(GELU_erf(Linear(x)))))       (post-LN, eps 1e-12)
Parameter names are the state_dict keys of the reference module (``bert.`` + HF names, ``classifier.0/2``); the pooler is a parameter
of the module but feeds nothing on this path (its gradient is None in the reference).

Dropout: the reference draws torch RNG masks, which no other implementation can reproduce.  Oracle and HIP kernels share the
counter-based generator below instead (``keep_mask``); the golden generator injects the same masks into the HF modules, so train-mode
parity is exact in the masks and the reference's own arithmetic is what gets compared.
"""
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

BertCfg = namedtuple("BertCfg", "vocab hidden layers heads inter max_pos num_classes p_drop", defaults=(0.1,))
BERT_BASE = dict(vocab=30522, hidden=768, layers=12, heads=12, inter=3072, max_pos=512)          # bert-base-uncased
BERT_TINY_TEST = dict(vocab=120, hidden=128, layers=2, heads=2, inter=512, max_pos=64)
LN_EPS = 1e-12
PAD_ID = 0

# dropout sites of one forward, in call order (HF eager attention): site id = 4 * layer + k, k in SITE_*; embeddings / head separately
SITE_EMB, SITE_HEAD = 0x7FFFFFF0, 0x7FFFFFF1
SITE_PROBS, SITE_ATTN_OUT, SITE_FFN_OUT = 0, 1, 2


def param_shapes(cfg):
    """named_parameters() order of the reference ClassificationBert."""
    D, I = cfg.hidden, cfg.inter
    s = [("bert.embeddings.word_embeddings.weight", (cfg.vocab, D)), ("bert.embeddings.position_embeddings.weight", (cfg.max_pos, D)),
         ("bert.embeddings.token_type_embeddings.weight", (2, D)), ("bert.embeddings.LayerNorm.weight", (D,)),
         ("bert.embeddings.LayerNorm.bias", (D,))]
    for i in range(cfg.layers):
        p = "bert.encoder.layer.%d." % i
        for n in ("query", "key", "value"):
            s += [(p + "attention.self.%s.weight" % n, (D, D)), (p + "attention.self.%s.bias" % n, (D,))]
        s += [(p + "attention.output.dense.weight", (D, D)), (p + "attention.output.dense.bias", (D,)),
              (p + "attention.output.LayerNorm.weight", (D,)), (p + "attention.output.LayerNorm.bias", (D,)),
              (p + "intermediate.dense.weight", (I, D)), (p + "intermediate.dense.bias", (I,)),
              (p + "output.dense.weight", (D, I)), (p + "output.dense.bias", (D,)),
              (p + "output.LayerNorm.weight", (D,)), (p + "output.LayerNorm.bias", (D,))]
    s += [("bert.pooler.dense.weight", (D, D)), ("bert.pooler.dense.bias", (D,)),
          ("classifier.0.weight", (D, D)), ("classifier.0.bias", (D,)), ("classifier.2.weight", (cfg.num_classes, D)),
          ("classifier.2.bias", (cfg.num_classes,))]
    return s


# ---- counter-based dropout generator shared with semireward_amd/csrc/enc_ops.hip / attention.hip ----------------------------------
def _fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16); h *= np.uint32(0x85EBCA6B); h ^= h >> np.uint32(13); h *= np.uint32(0xC2B2AE35); h ^= h >> np.uint32(16)
    return h


def site_key(seed, site):
    """32-bit key of one dropout site of one forward call (seed: 64-bit call seed)."""
    with np.errstate(over="ignore"):
        lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
        a = _fmix32(np.array([np.uint32(hi + np.uint32(0x9E3779B9) * np.uint32(site & 0xFFFFFFFF))], dtype=np.uint32))[0]
        return int(_fmix32(np.array([lo ^ a], dtype=np.uint32))[0])


def keep_mask(seed, site, shape, p):
    """Boolean keep mask of one dropout site (semireward_amd/csrc/common.h drop_keep / drop_pair_hash): ONE hash decides TWO neighbouring
    elements.  With key = site_key(seed, site) and t16 = floor(p * 2^32) >> 16:
      * element with row-major linear index i belongs to pair i >> 1; h = fmix32(pair * 0x9E3779B1 + key); the even element is kept iff
        (h & 0xFFFF) >= t16, the odd one iff (h >> 16) >= t16;
      * 4-D shapes [B, H, N, M] = attention probabilities (the one 4-D site; M = N, or the frame pitch of the Wav2Vec2 layout): the pairs are
        taken within a query row -- pair index = ((b H + h) N + q) ceil(M / 2) + (key >> 1) -- so that a lane's four consecutive keys are two
        whole pairs whatever the parity of M.
    (The hash -- two quarter-rate 32-bit multiplies -- was 74 % of the attention forward's vector-ALU work at L = 512 and a quarter of a GEMM
    tile's time where the epilogue carries a dropout; one decision per 16 bits resolves p to 2^-16.)"""
    n = int(np.prod(shape))
    assert n < 2 ** 32
    t16 = np.uint32(int(p * 4294967296.0) >> 16)
    key = np.uint32(site_key(seed, site))
    if len(shape) == 4:
        B, H, N, M = shape
        nh = (M + 1) // 2
        with np.errstate(over="ignore"):
            rows = np.arange(B * H * N, dtype=np.uint64).astype(np.uint32)
            pair = rows[:, None] * np.uint32(nh) + np.arange(nh, dtype=np.uint32)[None, :]
            h = _fmix32(pair * np.uint32(0x9E3779B1) + key)
        keep = np.empty((B * H * N, 2 * nh), dtype=bool)
        keep[:, 0::2] = (h & np.uint32(0xFFFF)) >= t16
        keep[:, 1::2] = (h >> np.uint32(16)) >= t16
        return keep[:, :M].reshape(shape)
    npair = (n + 1) // 2
    with np.errstate(over="ignore"):
        h = _fmix32(np.arange(npair, dtype=np.uint64).astype(np.uint32) * np.uint32(0x9E3779B1) + key)
    keep = np.empty(2 * npair, dtype=bool)
    keep[0::2] = (h & np.uint32(0xFFFF)) >= t16
    keep[1::2] = (h >> np.uint32(16)) >= t16
    return keep[:n].reshape(shape)


def _drop(x, seed, site, p):
    if seed is None or p <= 0.0:
        return x
    return x * torch.from_numpy(keep_mask(seed, site, tuple(x.shape), p).astype(np.float32) / np.float32(1.0 - p))


def bert_forward(P, ids, mask, cfg, seed=None):
    """ids int64 [B, L], mask int64 [B, L] (1 = token, 0 = padding).  seed None: no dropout (eval mode); else the 64-bit call seed of
    the shared generator (train mode).  Returns dict(logits [B, C], feat [B, hidden], hidden [B, L, hidden])."""
    B, L = ids.shape
    D, H, p = cfg.hidden, cfg.heads, cfg.p_drop
    e = "bert.embeddings."
    # nn.Embedding(vocab, hidden, padding_idx=pad_token_id = 0): the [PAD] row never receives a gradient
    x = F.embedding(ids, P[e + "word_embeddings.weight"], padding_idx=PAD_ID) + P[e + "position_embeddings.weight"][:L][None] \
        + P[e + "token_type_embeddings.weight"][0]
    x = F.layer_norm(x, (D,), P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"], LN_EPS)
    x = _drop(x, seed, SITE_EMB, p)
    neg = torch.where(mask.bool(), 0.0, float("-inf"))[:, None, None, :]                       # additive key mask
    for i in range(cfg.layers):
        q_ = "bert.encoder.layer.%d." % i
        lin = lambda t, n: F.linear(t, P[q_ + n + ".weight"], P[q_ + n + ".bias"])   # noqa: E731
        heads = lambda t: t.view(B, L, H, D // H).transpose(1, 2)   # noqa: E731
        q, k, v = heads(lin(x, "attention.self.query")), heads(lin(x, "attention.self.key")), heads(lin(x, "attention.self.value"))
        probs = torch.softmax(q @ k.transpose(-1, -2) * (D // H) ** -0.5 + neg, dim=-1)
        probs = _drop(probs, seed, 4 * i + SITE_PROBS, p)
        ctx = (probs @ v).transpose(1, 2).reshape(B, L, D)
        x = F.layer_norm(x + _drop(lin(ctx, "attention.output.dense"), seed, 4 * i + SITE_ATTN_OUT, p), (D,),
                         P[q_ + "attention.output.LayerNorm.weight"], P[q_ + "attention.output.LayerNorm.bias"], LN_EPS)
        h = F.gelu(lin(x, "intermediate.dense"))
        x = F.layer_norm(x + _drop(lin(h, "output.dense"), seed, 4 * i + SITE_FFN_OUT, p), (D,),
                         P[q_ + "output.LayerNorm.weight"], P[q_ + "output.LayerNorm.bias"], LN_EPS)
    feat = _drop(x, seed, SITE_HEAD, p).mean(1)                                                   # bert.py:36-37: padding included
    hcls = F.gelu(F.linear(feat, P["classifier.0.weight"], P["classifier.0.bias"]))
    logits = F.linear(hcls, P["classifier.2.weight"], P["classifier.2.bias"])
    return dict(logits=logits, feat=feat, hidden=x)


def synth_params(cfg, seed):
    """HF-style init (normal std 0.02, LayerNorm 1/0, zero biases) perturbed so that no gradient path is degenerate."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for n, shp in param_shapes(cfg):
        if "LayerNorm.weight" in n:
            out[n] = (1.0 + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif len(shp) == 1:
            out[n] = (0.02 * rng.standard_normal(shp)).astype(np.float32)
        elif n.startswith("classifier"):
            out[n] = (rng.standard_normal(shp) / np.sqrt(shp[1])).astype(np.float32)
        else:
            out[n] = (0.05 * rng.standard_normal(shp)).astype(np.float32)
    return out


def synth_tokens(seed, B, L, vocab, ragged=True):
    """Right-padded batch as ``tokenizer.pad`` produces it (nlp_collactor.py:63-69): ids int64 [B, L], mask int64 [B, L]; the longest
    row fills L, padded positions carry id 0 ([PAD])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = rng.integers(1, vocab, size=(B, L), dtype=np.int64)
    lens = rng.integers(max(2, L // 3), L + 1, size=B) if ragged else np.full(B, L)
    lens[rng.integers(0, B)] = L
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    return ids * mask, mask
