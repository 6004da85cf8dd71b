"""BERT classification backbone engine on libsrhip: forward / backward over a flat parameter block.

Mirrors the reference's ``semilearn/nets/bert/bert.py`` plugin surface (class ``ClassificationBert``, builders ``bert_base_uncased`` /
``bert_base_cased``, ``state_dict`` keys = ``bert.`` + the HF BertModel names + ``classifier.0/2``, dict input
``{'input_ids', 'attention_mask'}``, ``{'logits','feat'}`` result, ``group_matcher`` / ``no_weight_decay``) but not its implementation:
no nn.Module, no autograd, no ``transformers``.  The encoder the reference obtains from ``transformers.BertModel`` (post-LN layers,
erf-GELU, LayerNorm eps 1e-12, dropout 0.1 in train mode) is a short list of HIP launches per layer:

    qkv  = x_bf16 . Wqkv^T + b                       srhip_gemm_nt            (q | k | v packed: one product, N = 3D)
    ctx  = softmax(q k^T / 8 + key mask) v           srhip_attn_masked_fwd    (per-sequence key length, dropout on the probabilities)
    y1   = x + dropout(ctx . Wo^T + b)               srhip_gemm_nt_resid_dropout
    x    = LayerNorm(y1)          (fp32 + bf16)      srhip_postln_fwd
    h    = GELU(x_bf16 . W1^T + b)                   srhip_gemm_nt (GELU epilogue)
    y2   = x + dropout(h . W2^T + b)                 srhip_gemm_nt_resid_dropout
    x    = LayerNorm(y2)                             srhip_postln_fwd

and the backward is hand-written (post-LN: the gradient of a LayerNorm input feeds the residual path in fp32 and, dropout-masked, the
branch's dX / dW products in bf16).  Data layout: B sequences padded to a common L, M = B * L token rows; the residual stream is fp32
[M, D], GEMM operands bf16; the flat parameter block keeps q/k/v weights (and biases) of a layer adjacent, so the packed [3D, D]
operand is a VIEW, and a reference checkpoint still maps by name.  ``from_pretrained`` needs the network: weights are random-init
(HF: normal(0, 0.02), LayerNorm 1/0) or loaded with ``load_state_dict``.

Padding: batches are right-padded (``tokenizer.pad``, nlp_collactor.py:63-69), so the attention mask is a prefix mask and the engine
carries ``key_len = mask.sum(1)``; every position -- padding included -- is computed and averaged, exactly like bert.py:36-37.
"""
import types

import torch

from .. import ops
from .encoder import PostLNEncoderMixin

SITE_EMB, SITE_HEAD = 0x7FFFFFF0, 0x7FFFFFF1


class BertConfig:
    def __init__(self, vocab=30522, hidden=768, layers=12, heads=12, inter=3072, max_pos=512, num_classes=2, p_drop=0.1, eps=1e-12,
                 pad_id=0):
        self.vocab, self.hidden, self.layers, self.heads, self.inter, self.max_pos = vocab, hidden, layers, heads, inter, max_pos
        self.num_classes, self.p_drop, self.eps, self.pad_id = num_classes, p_drop, eps, pad_id
        assert hidden // heads == 64 and hidden in (128, 384, 768), "libsrhip attention is built for head_dim 64"
        assert inter % 32 == 0 and max_pos <= 512
    embed_dim = property(lambda self: self.hidden)
    depth = property(lambda self: self.layers)
    drop_path_rate = 0.0


E = "bert.embeddings."


def _layer(i):
    return "bert.encoder.layer.%d." % i


def param_names_shapes(cfg):
    """Flat-block order: the reference's names; q/k/v weights, then q/k/v biases, adjacent (packed operand views)."""
    D, I = cfg.hidden, cfg.inter
    out = [(E + "word_embeddings.weight", (cfg.vocab, D)), (E + "position_embeddings.weight", (cfg.max_pos, D)),
           (E + "token_type_embeddings.weight", (2, D)), (E + "LayerNorm.weight", (D,)), (E + "LayerNorm.bias", (D,))]
    for i in range(cfg.layers):
        p = _layer(i)
        out += [(p + "attention.self.%s.weight" % n, (D, D)) for n in ("query", "key", "value")]
        out += [(p + "attention.self.%s.bias" % n, (D,)) for n in ("query", "key", "value")]
        out += [(p + "attention.output.dense.weight", (D, D)), (p + "attention.output.dense.bias", (D,)),
                (p + "attention.output.LayerNorm.weight", (D,)), (p + "attention.output.LayerNorm.bias", (D,)),
                (p + "intermediate.dense.weight", (I, D)), (p + "intermediate.dense.bias", (I,)),
                (p + "output.dense.weight", (D, I)), (p + "output.dense.bias", (D,)),
                (p + "output.LayerNorm.weight", (D,)), (p + "output.LayerNorm.bias", (D,))]
    out += [("bert.pooler.dense.weight", (D, D)), ("bert.pooler.dense.bias", (D,)),
            ("classifier.0.weight", (D, D)), ("classifier.0.bias", (D,)), ("classifier.2.weight", (cfg.num_classes, D)),
            ("classifier.2.bias", (cfg.num_classes,))]
    return out


class TokenBatch:
    """A right-padded token batch on the device: ids int64 [S, L] (contiguous), key_len int32 [S] = valid tokens per sequence,
    seq_len int32 [S] or None = the padded length of every sequence's own source batch when batches of different padded lengths were
    concatenated (rows [seq_len, L) are filler: masked as keys, left out of the mean pool, zero gradient -- results equal separate calls)."""
    __slots__ = ("ids", "key_len", "seq_len", "S", "L")

    def __init__(self, ids, key_len, seq_len=None):
        self.ids, self.key_len, self.seq_len = ids, key_len, seq_len
        self.S, self.L = ids.shape

    @classmethod
    def from_dict(cls, x, device):
        ids = x["input_ids"].to(device=device, dtype=torch.int64).contiguous()
        S, L = ids.shape
        kl = torch.empty(S, dtype=torch.int32, device=device)
        am = x.get("attention_mask")
        if am is None:
            kl.fill_(L)
        else:
            ops.mask_lengths(am.to(device=device, dtype=torch.int64).contiguous(), kl, S, L)
        return cls(ids, kl)

    @classmethod
    def cat(cls, batches):
        """One batch out of several (x_lb, x_ulb_w, x_ulb_s of one step; the reference forwards them in separate model calls under
        use_cat=False, each padded to its own longest row).  Shorter batches are filled up to the longest L with [PAD]."""
        L = max(b.L for b in batches)
        kl = torch.cat([b.key_len for b in batches]).contiguous()
        if all(b.L == L and b.seq_len is None for b in batches):
            return cls(torch.cat([b.ids for b in batches]).contiguous(), kl)
        ids = torch.zeros(sum(b.S for b in batches), L, dtype=torch.int64, device=kl.device)
        sl, r = [], 0
        for b in batches:
            ids[r:r + b.S, :b.L] = b.ids
            sl.append(b.seq_len if b.seq_len is not None else torch.full((b.S,), b.L, dtype=torch.int32, device=kl.device))
            r += b.S
        return cls(ids, kl, torch.cat(sl).contiguous())


class ClassificationBert(PostLNEncoderMixin):
    couples_batch_rows = False
    takes_tokens = True
    rows_independent = True       # LayerNorm only, counter-based dropout indexed per row

    def __init__(self, cfg=None, device="cuda", **kw):
        self.cfg = cfg if cfg is not None else BertConfig(**kw)
        cfg = self.cfg
        self.device = torch.device(device)
        self.num_features = cfg.hidden
        self.names_shapes = param_names_shapes(cfg)
        self.offsets, o = {}, 0
        for n, s in self.names_shapes:
            self.offsets[n] = (o, s)
            o += int(torch.Size(s).numel())
            o = (o + 7) // 8 * 8                     # 16-byte alignment of every tensor in the bf16 copy
        self.numel = o
        f32, bf16 = torch.float32, torch.bfloat16
        self.flat = torch.zeros(o, dtype=f32, device=self.device)
        self.grad = torch.zeros(o, dtype=f32, device=self.device)
        self.flat_bf16 = torch.zeros(o, dtype=bf16, device=self.device)
        self.enc_alloc_wT()                           # per layer: transposed bf16 operands of the dX products
        self.enc_p = dict(attn=cfg.p_drop, hidden=cfg.p_drop, act=0.0)
        self.training = True
        self._ws, self._wT_desc = {}, None
        self._rng_calls, self.seed = 0, 0
        self.inject_seed = None                       # tests: the 64-bit dropout seed of the next forward calls

    # ---- parameter plumbing (same surface as nets/vit.py) --------------------------------------------
    def p(self, name, buf=None):
        """Flat view of parameter ``name`` inside ``buf`` (default: the parameter block).  Cached per (name, buffer): building a slice view costs
        ~3 us of host time and a step asks for ~500 of them -- more than half of the step's enqueue time before the cache."""
        b = self.flat if buf is None else buf
        if not (b is self.flat or b is self.grad or b is getattr(self, "flat_bf16", None)):
            o, s = self.offsets[name]                     # some other block (optimizer state, a test's copy): no entry is kept for it
            return b[o:o + int(torch.Size(s).numel())]
        pv = self.__dict__.setdefault("_pviews", {})
        ent = pv.get((name, id(b)))
        if ent is None:
            o, s = self.offsets[name]
            ent = pv[(name, id(b))] = (b, b[o:o + int(torch.Size(s).numel())])     # (holds ``b``: its id stays unique)
        return ent[1]

    def view(self, name, buf=None):
        return self.p(name, buf).view(self.offsets[name][1])

    def enc_names(self, i):
        p = _layer(i)
        return dict(q_w=p + "attention.self.query.weight", q_b=p + "attention.self.query.bias",
                    o_w=p + "attention.output.dense.weight", o_b=p + "attention.output.dense.bias",
                    ln1_w=p + "attention.output.LayerNorm.weight", ln1_b=p + "attention.output.LayerNorm.bias",
                    w1=p + "intermediate.dense.weight", b1=p + "intermediate.dense.bias",
                    w2=p + "output.dense.weight", b2=p + "output.dense.bias",
                    ln2_w=p + "output.LayerNorm.weight", ln2_b=p + "output.LayerNorm.bias")

    def named_parameters(self):
        return [(n, self.view(n)) for n, _ in self.names_shapes]

    def named_grads(self):
        return [(n, self.view(n, self.grad)) for n, _ in self.names_shapes]

    def state_dict(self):
        return {n: self.view(n).detach().clone() for n, _ in self.names_shapes}

    def load_state_dict(self, sd, strict=True):
        for n, s in self.names_shapes:
            if n in sd:
                self.view(n).copy_(torch.as_tensor(sd[n]).to(self.device, torch.float32).reshape(s))
            elif strict:
                raise KeyError(n)
        self.refresh_operands()

    def init_weights(self, seed=0):
        """HF BertPreTrainedModel._init_weights: normal(0, 0.02) for Linear / Embedding weights ([PAD] row zero), LayerNorm 1 / 0, biases 0;
        the classifier Linears keep torch's default U(+-1/sqrt(fan_in))."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        sd = {}
        for n, s in self.names_shapes:
            if "LayerNorm.weight" in n:
                sd[n] = torch.ones(s)
            elif n.startswith("classifier"):
                bound = 1.0 / (self.cfg.hidden ** 0.5)
                sd[n] = (torch.rand(s, generator=g) * 2 - 1) * bound
            elif len(s) == 1:
                sd[n] = torch.zeros(s)
            else:
                sd[n] = torch.randn(s, generator=g) * 0.02
        sd[E + "word_embeddings.weight"][self.cfg.pad_id].zero_()
        self.load_state_dict(sd)

    def refresh_operands(self):
        ops.cast_f32_bf16(self.flat, self.flat_bf16, self.numel)
        self.refresh_transposed()

    def refresh_transposed(self):
        if self._wT_desc is None:
            self._wT_desc = ops.make_transpose_desc(self.enc_transpose_items(), self.device)
        ops.transpose_batched(*self._wT_desc)

    def no_weight_decay(self):
        return []

    frozen_params = ("bert.pooler.dense.weight", "bert.pooler.dense.bias")    # feed nothing on this path: grad None in the reference

    def layer_ids(self):
        """group_with_matcher(named_parameters, group_matcher, reverse=True) (nets/utils.py:208-270): embeddings 0, encoder layer i -> i + 1,
        everything unmatched (pooler, classifier) -> the last id."""
        L = self.cfg.layers
        ids = {}
        for n, _ in self.names_shapes:
            if n.startswith("bert.embeddings"):
                ids[n] = 0
            elif n.startswith("bert.encoder.layer."):
                ids[n] = int(n.split(".")[3]) + 1
            else:
                ids[n] = L + 1
        return ids, L + 1

    def group_matcher(self, coarse=False, prefix=""):
        return dict(stem=r"^{}bert.embeddings".format(prefix), blocks=r"^{}bert.encoder.layer.(\d+)".format(prefix))

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def zero_grad(self):
        self.grad.zero_()

    def _buf(self, key, shape, dtype):
        t = self._ws.get(key)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    def next_seed(self):
        """64-bit dropout seed of one forward call (None in eval mode / p_drop == 0)."""
        if not self.training or self.cfg.p_drop <= 0.0:
            return None
        if self.inject_seed is not None:
            return self.inject_seed
        self._rng_calls += 1
        return ((self.seed & 0xFFFFFFFF) << 32) + self._rng_calls

    # ---- forward ----------------------------------------------------------------------------------------
    def _ctx_buffers(self, B, L, tag):
        key = ("ctx", B, L, tag)
        if key in self._ws:
            return self._ws[key]
        c = types.SimpleNamespace()
        self.enc_alloc_ctx(c, B, L)
        self.head_alloc_ctx(c, B)
        c.st0 = torch.empty(2, B * L, dtype=torch.float32, device=self.device)
        self._ws[key] = c
        return c

    def forward_features(self, tok, seq_index=None, droppath=None, save=False, B=None, seed="auto", tag="", buftag=""):
        """tok: TokenBatch; seq_index int32 [B] (optional gather of sequences: lets the K+1 passes of one SemiReward step share one
        copy of the tokens).  Returns (logits [B, C], feat [B, D], ctx or None).  ``droppath`` is accepted for interface parity
        with the ViT engine and unused (BERT has no stochastic depth)."""
        cfg = self.cfg
        D, I, H, C, L = cfg.hidden, cfg.inter, cfg.heads, cfg.num_classes, tok.L
        B = int(seq_index.numel()) if seq_index is not None else tok.S
        M = B * L
        f32, bf16 = torch.float32, torch.bfloat16
        seed = self.next_seed() if seed == "auto" else seed
        key_len = tok.key_len if seq_index is None else tok.key_len.index_select(0, seq_index.long()).contiguous()
        seq_len = None
        if tok.seq_len is not None:
            seq_len = tok.seq_len if seq_index is None else tok.seq_len.index_select(0, seq_index.long()).contiguous()
        P = self.p
        t = ("s" if save else "i") + buftag
        dr = (lambda site, p=cfg.p_drop: ops.Drop(seed, site, p) if p > 0 else None) if seed is not None else (lambda site, p=0.0: None)
        ctx = None
        x = self._buf(t + "x", (M, D), f32)
        if save:
            ctx = self._ctx_buffers(B, L, tag)
            ctx.B, ctx.L, ctx.tok, ctx.seq_index, ctx.key_len, ctx.seq_len, ctx.seed = B, L, tok, seq_index, key_len, seq_len, seed
            xb = ctx.xb[0]
        else:
            xb = self._buf(t + "xb", (M, D), bf16)
        ops.embed_ln_fwd(tok.ids, seq_index, P(E + "word_embeddings.weight"), P(E + "position_embeddings.weight"),
                         P(E + "token_type_embeddings.weight"), P(E + "LayerNorm.weight"), P(E + "LayerNorm.bias"), cfg.eps, x, xb,
                         ctx.st0[0] if save else None, ctx.st0[1] if save else None, B, L, D, dr(SITE_EMB))
        self.enc_forward(x, xb, ctx, save, B, L, key_len, dr, tag=t)
        logits, feat = self.head_forward(x, B, L, dr(SITE_HEAD), seq_len, ctx)
        return logits, feat, ctx

    def forward(self, x, only_fc=False, only_feat=False, return_embed=False, **kw):
        """Reference-compatible entry (bert.py:22-48): x = {'input_ids', 'attention_mask'} -> {'logits','feat'}."""
        assert not only_fc and not return_embed, "only_fc / return_embed (VAT) are not on the SemiReward hot path"
        tok = x if isinstance(x, TokenBatch) else TokenBatch.from_dict(x, self.device)
        logits, feat, _ = self.forward_features(tok, None, save=False)
        return feat if only_feat else {"logits": logits, "feat": feat}

    __call__ = forward

    def extract(self, x):
        return self.forward(x, only_feat=True)

    # ---- backward -----------------------------------------------------------------------------------------
    def backward(self, ctx, dlogits):
        """Accumulates d(loss)/d(params) into ``self.grad`` given dlogits fp32 [B, C] of a save=True forward."""
        cfg = self.cfg
        D = cfg.hidden
        B, L, seed = ctx.B, ctx.L, ctx.seed
        M = B * L
        dr = (lambda site, p=cfg.p_drop: ops.Drop(seed, site, p) if p > 0 else None) if seed is not None else (lambda site, p=0.0: None)
        P, G = self.p, (lambda n: self.p(n, self.grad))
        dx = self._buf("b_dx", (M, D), torch.float32)
        self.head_backward(ctx, dlogits, dx, B, L, dr(SITE_HEAD), ctx.seq_len)
        self.enc_backward(dx, ctx, B, L, ctx.key_len, dr)
        ops.embed_ln_bwd(dx, ctx.tok.ids, ctx.seq_index, P(E + "word_embeddings.weight"), P(E + "position_embeddings.weight"),
                         P(E + "token_type_embeddings.weight"), ctx.st0[0], ctx.st0[1], P(E + "LayerNorm.weight"),
                         G(E + "word_embeddings.weight"), G(E + "position_embeddings.weight"), G(E + "token_type_embeddings.weight"),
                         G(E + "LayerNorm.weight"), G(E + "LayerNorm.bias"), B, L, D, cfg.pad_id, dr(SITE_EMB))


# ---- builders with the reference's names (bert.py:62-69); pretrained checkpoints need the network -> random init -------------
def _build(num_classes, kw, **cfg):
    kw = {k: v for k, v in kw.items() if k not in ("pretrained", "pretrained_path")}
    device = kw.pop("device", "cuda")
    m = ClassificationBert(BertConfig(num_classes=num_classes, **cfg), device=device)
    m.init_weights(kw.pop("seed", 0))
    return m


def bert_base_uncased(num_classes=2, **kw):
    return _build(num_classes, kw, vocab=30522)


def bert_base_cased(num_classes=2, **kw):
    return _build(num_classes, kw, vocab=28996)


def bert_tiny_test(num_classes=4, **kw):
    return _build(num_classes, kw, vocab=120, hidden=128, layers=2, heads=2, inter=512, max_pos=64)
