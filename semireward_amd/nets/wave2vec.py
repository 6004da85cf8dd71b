"""Wav2Vec2 classification backbone engine on libsrhip: forward / backward over a flat parameter block.

Mirrors the reference's ``semilearn/nets/wave2vecv2/wave2vecv2.py`` plugin surface (class ``ClassificationWave2Vec``, builder
``wave2vecv2_base``, ``state_dict`` keys = ``model.`` + the HF Wav2Vec2Model names + ``classifier.0/2``, raw-waveform input [B, samples],
``{'logits','feat'}`` result, ``group_matcher``) but not its implementation: no nn.Module, no autograd, no ``transformers``.  The model the
reference obtains from ``transformers.Wav2Vec2Model`` (facebook/wav2vec2-base-960h family: GroupNorm feature encoder, post-LN encoder) is

    waveform -> conv0 + GroupNorm + GELU            srhip_w2v_conv0            (direct kernel, 1 input channel)
             -> 6 x Conv1d + GELU                   srhip_gemm_nt, GELU epilogue: the unfolded operand is the previous activation
                                                    itself, read with overlapping rows (lda = stride * C)  -- no im2col
             -> LayerNorm -> Linear -> dropout      srhip_w2v_featln_fwd, srhip_gemm_nt_resid_dropout
             -> SpecAugment (train)                 srhip_w2v_spec_mask_fwd
             -> x + GELU(grouped pos-conv(x))       srhip_w2v_pos_stage + srhip_gemm_nt_grouped_f32 (16 groups, overlapping rows)
             -> LayerNorm -> dropout                srhip_w2v_pos_finish_fwd
             -> 12 post-LN layers (LayerDrop)       nets/encoder.py (the BERT layer kernels + dropout after the GELU)
             -> dropout -> mean over frames -> Linear / GELU / Linear

with a hand-written backward through all of it (the feature encoder is trained: ``_requires_grad = False`` at wave2vecv2.py:14 only stops
HF from marking the INPUT as requiring a gradient).  Frames per clip are fixed by the sample count (64000 -> 199); every activation keeps a
per-layer frame pitch (see csrc/w2v_ops.hip), the encoder runs on pitch-P rows with key length T, so the single filler frame is masked,
not averaged and receives zero gradient.  ``from_pretrained`` needs the network: weights are random-init or ``load_state_dict``.

Train-mode randomness of the reference (torch dropout, numpy SpecAugment spans, torch.rand LayerDrop) is drawn here from the counter-based
dropout generator and a per-call numpy Generator; tests inject all three.  LayerDrop is decided once per engine forward (= one launch train
over all the (pass, clip) rows it batches), like one ``model(x)`` call of the reference.
"""
import types

import numpy as np
import torch

from .. import ops
from .encoder import PostLNEncoderMixin

SITE_EMB, SITE_HEAD, SITE_FEATPROJ = 0x7FFFFFF0, 0x7FFFFFF1, 0x7FFFFFF2
M_ = "model."
FE = "model.feature_extractor.conv_layers."
PC = "model.encoder.pos_conv_embed.conv."


class W2vConfig:
    def __init__(self, hidden=768, layers=12, heads=12, inter=3072, conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2),
                 conv_stride=(5, 2, 2, 2, 2, 2, 2), pos_k=128, pos_groups=16, num_classes=2, p_hidden=0.1, p_act=0.1, p_attn=0.1,
                 p_featproj=0.1, p_head=0.1, layerdrop=0.1, mask_time_prob=0.05, mask_time_length=10, mask_time_min_masks=2, eps=1e-5):
        self.hidden, self.layers, self.heads, self.inter = hidden, layers, heads, inter
        self.conv_dim, self.conv_kernel, self.conv_stride = tuple(conv_dim), tuple(conv_kernel), tuple(conv_stride)
        self.pos_k, self.pos_groups, self.num_classes, self.eps = pos_k, pos_groups, num_classes, eps
        self.p_hidden, self.p_act, self.p_attn, self.p_featproj, self.p_head, self.layerdrop = p_hidden, p_act, p_attn, p_featproj, p_head, layerdrop
        self.mask_time_prob, self.mask_time_length, self.mask_time_min_masks = mask_time_prob, mask_time_length, mask_time_min_masks
        assert hidden // heads == 64 and hidden in (128, 384, 768), "libsrhip attention is built for head_dim 64"
        assert len(set(conv_dim)) == 1 and conv_dim[0] in (128, 256, 512, 768) and conv_kernel[0] <= 16
        assert all(k <= 2 * s for k, s in zip(conv_kernel[1:], conv_stride[1:])) and hidden % pos_groups == 0
        assert (pos_k * (hidden // pos_groups)) % 32 == 0 and (hidden // pos_groups) % 8 == 0 and inter % 32 == 0
    embed_dim = property(lambda self: self.hidden)
    depth = property(lambda self: self.layers)
    drop_path_rate = 0.0

    def frames(self, samples):
        out, t = [], samples
        for k, s in zip(self.conv_kernel, self.conv_stride):
            t = (t - k) // s + 1
            out.append(t)
        return out


def _enc(i):
    return "model.encoder.layers.%d." % i


def param_names_shapes(cfg):
    """Flat-block order: the reference's names; q/k/v weights then q/k/v biases adjacent (packed operand views)."""
    D, I, C = cfg.hidden, cfg.inter, cfg.conv_dim
    out = [(M_ + "masked_spec_embed", (D,))]
    for i, (c, k) in enumerate(zip(C, cfg.conv_kernel)):
        out.append((FE + "%d.conv.weight" % i, (c, C[i - 1] if i else 1, k)))
        if i == 0:
            out += [(FE + "0.layer_norm.weight", (c,)), (FE + "0.layer_norm.bias", (c,))]
    out += [(M_ + "feature_projection.layer_norm.weight", (C[-1],)), (M_ + "feature_projection.layer_norm.bias", (C[-1],)),
            (M_ + "feature_projection.projection.weight", (D, C[-1])), (M_ + "feature_projection.projection.bias", (D,)),
            (PC + "bias", (D,)), (PC + "parametrizations.weight.original0", (1, 1, cfg.pos_k)),
            (PC + "parametrizations.weight.original1", (D, D // cfg.pos_groups, cfg.pos_k)),
            (M_ + "encoder.layer_norm.weight", (D,)), (M_ + "encoder.layer_norm.bias", (D,))]
    for i in range(cfg.layers):
        p = _enc(i)
        out += [(p + "attention.%s.weight" % n, (D, D)) for n in ("q_proj", "k_proj", "v_proj")]
        out += [(p + "attention.%s.bias" % n, (D,)) for n in ("q_proj", "k_proj", "v_proj")]
        out += [(p + "attention.out_proj.weight", (D, D)), (p + "attention.out_proj.bias", (D,)),
                (p + "layer_norm.weight", (D,)), (p + "layer_norm.bias", (D,)),
                (p + "feed_forward.intermediate_dense.weight", (I, D)), (p + "feed_forward.intermediate_dense.bias", (I,)),
                (p + "feed_forward.output_dense.weight", (D, I)), (p + "feed_forward.output_dense.bias", (D,)),
                (p + "final_layer_norm.weight", (D,)), (p + "final_layer_norm.bias", (D,))]
    out += [("classifier.0.weight", (D, D)), ("classifier.0.bias", (D,)), ("classifier.2.weight", (cfg.num_classes, D)),
            ("classifier.2.bias", (cfg.num_classes,))]
    return out


def spec_augment_mask(rng, B, T, mask_prob, mask_length, min_masks):
    """transformers' _compute_mask_indices (no attention mask): per clip max(int(p T / len + eps), min_masks) spans of ``mask_length``
    frames at distinct random starts.  bool [B, T]."""
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


class ClassificationWave2Vec(PostLNEncoderMixin):
    couples_batch_rows = False

    def __init__(self, cfg=None, device="cuda", **kw):
        self.cfg = cfg if cfg is not None else W2vConfig(**kw)
        cfg = self.cfg
        self.device = torch.device(device)
        self.num_features = cfg.hidden
        self.names_shapes = param_names_shapes(cfg)
        self.offsets, o = {}, 0
        for n, s in self.names_shapes:
            self.offsets[n] = (o, s)
            o = (o + int(torch.Size(s).numel()) + 7) // 8 * 8
        self.numel = o
        f32, bf16, dev = torch.float32, torch.bfloat16, self.device
        self.flat = torch.zeros(o, dtype=f32, device=dev)
        self.grad = torch.zeros(o, dtype=f32, device=dev)
        self.flat_bf16 = torch.zeros(o, dtype=bf16, device=dev)
        self.enc_alloc_wT()
        self.enc_p = dict(attn=cfg.p_attn, hidden=cfg.p_hidden, act=cfg.p_act)
        C, D, k, G = cfg.conv_dim[0], cfg.hidden, cfg.pos_k, cfg.pos_groups
        self.cg = D // G
        self.conv_w = [None] + [(torch.zeros(C, kk * C, dtype=bf16, device=dev), torch.zeros(kk * C, C, dtype=bf16, device=dev))
                                for kk in cfg.conv_kernel[1:]]
        self.projT = torch.zeros(C, D, dtype=bf16, device=dev)
        self.pos_norms = torch.zeros(k, dtype=f32, device=dev)
        self.pos_Wf = torch.zeros(G, self.cg, k * self.cg, dtype=bf16, device=dev)
        self.pos_Wb = torch.zeros(G, self.cg, k * self.cg, dtype=bf16, device=dev)
        self.training = True
        self._ws, self._wT_desc = {}, None
        self._rng_calls, self.seed = 0, 0
        self.inject = None            # tests: dict(seed=..., spec_mask=bool [B, T], skip=[...]) used by the next forward calls

    # ---- parameter plumbing ---------------------------------------------------------------------------------
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
        p = _enc(i)
        return dict(q_w=p + "attention.q_proj.weight", q_b=p + "attention.q_proj.bias", o_w=p + "attention.out_proj.weight",
                    o_b=p + "attention.out_proj.bias", ln1_w=p + "layer_norm.weight", ln1_b=p + "layer_norm.bias",
                    w1=p + "feed_forward.intermediate_dense.weight", b1=p + "feed_forward.intermediate_dense.bias",
                    w2=p + "feed_forward.output_dense.weight", b2=p + "feed_forward.output_dense.bias",
                    ln2_w=p + "final_layer_norm.weight", ln2_b=p + "final_layer_norm.bias")

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
        """HF Wav2Vec2PreTrainedModel._init_weights magnitudes: kaiming-normal conv filters, normal(0, 0.02) dense weights, LayerNorm /
        GroupNorm 1 / 0, uniform masked_spec_embed; the classifier Linears keep torch's default."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        sd = {}
        for n, s in self.names_shapes:
            if n.endswith("layer_norm.weight"):
                sd[n] = torch.ones(s)
            elif n.endswith("original0"):
                sd[n] = torch.ones(s)
            elif n == M_ + "masked_spec_embed":
                sd[n] = torch.rand(s, generator=g)
            elif n.startswith("classifier"):
                sd[n] = (torch.rand(s, generator=g) * 2 - 1) / (self.cfg.hidden ** 0.5)
            elif len(s) == 1:
                sd[n] = torch.zeros(s)
            elif len(s) == 3:
                sd[n] = torch.randn(s, generator=g) * (2.0 / (s[1] * s[2])) ** 0.5
            else:
                sd[n] = torch.randn(s, generator=g) * 0.02
        self.load_state_dict(sd)

    def refresh_operands(self):
        ops.cast_f32_bf16(self.flat, self.flat_bf16, self.numel)
        self.refresh_transposed()

    def refresh_transposed(self):
        cfg = self.cfg
        C, D = cfg.conv_dim[0], cfg.hidden
        if self._wT_desc is None:
            items = self.enc_transpose_items()
            items.append((self.p(M_ + "feature_projection.projection.weight"), True, C, self.projT, D, D, D, C, False))
            self._wT_desc = ops.make_transpose_desc(items, self.device)
        ops.transpose_batched(*self._wT_desc)
        for l in range(1, len(cfg.conv_kernel)):
            ops.w2v_conv_weight_prep(self.p(FE + "%d.conv.weight" % l), self.conv_w[l][0], self.conv_w[l][1], C, C, cfg.conv_kernel[l])
        ops.w2v_weightnorm_prep(self.p(PC + "parametrizations.weight.original1"), self.p(PC + "parametrizations.weight.original0"),
                                self.pos_norms, self.pos_Wf, self.pos_Wb, D, cfg.pos_groups, cfg.pos_k)

    def no_weight_decay(self):
        return []

    def group_matcher(self, coarse=False, prefix=""):
        return dict(stem=r"^{}model.feature_projection|^{}model.feature_extractor".format(prefix, prefix),
                    blocks=r"^{}model.encoder.layers.(\d+)".format(prefix))

    def layer_ids(self):
        """group_with_matcher(reverse=True) on group_matcher (wave2vecv2.py:51-53): feature extractor / projection 0, encoder layer i -> i + 1,
        everything unmatched (masked_spec_embed, pos_conv_embed, encoder.layer_norm, classifier) -> the last id."""
        L = self.cfg.layers
        ids = {}
        for n, _ in self.names_shapes:
            if n.startswith(M_ + "feature_projection") or n.startswith(M_ + "feature_extractor"):
                ids[n] = 0
            elif n.startswith(M_ + "encoder.layers."):
                ids[n] = int(n.split(".")[3]) + 1
            else:
                ids[n] = L + 1
        return ids, L + 1

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def zero_grad(self):
        self.grad.zero_()

    def _buf(self, key, shape, dtype, zero=False):
        t = self._ws.get(key)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    # ---- geometry ---------------------------------------------------------------------------------------------
    def geometry(self, samples):
        """(T_l frames, P_l pitches) per conv layer + the positional-conv pitch.  P_last >= T_last + 1, P_{l-1} = stride_l * P_l."""
        cfg = self.cfg
        T = cfg.frames(samples)
        P = [0] * len(T)
        P[-1] = (T[-1] + 1 + 7) // 8 * 8
        for l in range(len(T) - 1, 0, -1):
            P[l - 1] = cfg.conv_stride[l] * P[l]
        assert all(p >= t + 1 for p, t in zip(P, T))
        Pp = (T[-1] + cfg.pos_k + 7) // 8 * 8
        return T, P, Pp

    def _random_inputs(self, B, T):
        """(seed, spec_mask uint8 [B, T] or None, skip list or None) of one forward call."""
        cfg = self.cfg
        if not self.training:
            return None, None, None
        if self.inject is not None:
            sm = self.inject.get("spec_mask")
            return self.inject["seed"], (None if sm is None else np.asarray(sm, dtype=bool)), self.inject.get("skip")
        self._rng_calls += 1
        seed = ((self.seed & 0xFFFFFFFF) << 32) + self._rng_calls
        rng = np.random.Generator(np.random.PCG64(seed))
        skip = [bool(rng.random() < cfg.layerdrop) for _ in range(cfg.layers)]
        spec = spec_augment_mask(rng, B, T, cfg.mask_time_prob, cfg.mask_time_length, cfg.mask_time_min_masks) if cfg.mask_time_prob > 0 else None
        return seed, spec, skip

    def _front_buffers(self, B, samples, tag, save):
        key = ("front", B, samples, tag)
        if key in self._ws:
            return self._ws[key]
        cfg = self.cfg
        T, P, Pp = self.geometry(samples)
        C, D, G, cg, k = cfg.conv_dim[0], cfg.hidden, cfg.pos_groups, self.cg, cfg.pos_k
        f32, bf16, dev = torch.float32, torch.bfloat16, self.device
        n = len(T)
        c = types.SimpleNamespace(T=T, P=P, Pp=Pp)
        SL = 16                                         # slack rows: the overlapping-row reads of the last rows run past the last clip
        c.act = [torch.zeros(B * P[l] + SL, C, dtype=bf16, device=dev) for l in range(n)]
        c.pre = [None] + [torch.zeros(B * P[l] + SL, C, dtype=bf16, device=dev) if save else None for l in range(1, n)]
        c.ws = torch.zeros(B, C, 2, dtype=torch.float64, device=dev)
        M = B * P[-1]
        c.lnb = torch.zeros(M, C, dtype=bf16, device=dev)
        c.st_f = torch.empty(2, M, dtype=f32, device=dev)
        c.hidden = torch.empty(M, D, dtype=f32, device=dev)
        c.rows_total = B * Pp + k + 8
        c.Xg = torch.zeros(G, c.rows_total, cg, dtype=bf16, device=dev)
        c.conv = torch.zeros(B * Pp + 8, D, dtype=f32, device=dev)
        c.ysave = torch.empty(M, D, dtype=f32, device=dev) if save else None
        c.st_e = torch.empty(2, M, dtype=f32, device=dev)
        c.mask = torch.zeros(M, dtype=torch.uint8, device=dev)
        c.key_len = torch.full((B,), T[-1], dtype=torch.int32, device=dev)
        # grouped positional conv: conv[clip*Pp + t, g*cg : (g+1)*cg] = sum_{j, ci} Xg[g][clip*Pp + t + j][ci] * Wf[g][co][j][ci]
        Kc = k * cg
        c.pos_desc = ops.make_group_desc_ld(
            [(ops._pa(c.Xg, g * c.rows_total * cg), cg, ops._pa(self.pos_Wf, g * cg * Kc), Kc, ops._pa(c.conv, g * cg), D, B * Pp, cg, Kc)
             for g in range(G)], dev, bn=64 if cg <= 64 else 128)
        self._ws[key] = c
        return c

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward_features(self, wave, clip_index=None, droppath=None, save=False, B=None, tag="", buftag=""):
        """wave fp32 [n, samples]; clip_index int32 [B] (optional gather: the K+1 passes of one SemiReward step share one copy of the
        clips).  Returns (logits [B, C], feat [B, D], ctx or None)."""
        cfg = self.cfg
        D, C, nl = cfg.hidden, cfg.conv_dim[0], len(cfg.conv_kernel)
        if clip_index is not None:
            wave = wave.index_select(0, clip_index.long())
        wave = wave.contiguous()
        B, S = wave.shape
        t = ("s" + tag) if save else ("i" + buftag)
        f = self._front_buffers(B, S, t, save)
        T, P, Pp = f.T, f.P, f.Pp
        Tn, Pn = T[-1], P[-1]
        M = B * Pn
        seed, spec, skip = self._random_inputs(B, Tn)
        dr = (lambda site, p: ops.Drop(seed, site, p) if p > 0 else None) if seed is not None else (lambda site, p: None)
        Pm, wb = self.p, self.flat_bf16
        # ---- feature encoder
        f.ws.zero_()
        k0, s0 = cfg.conv_kernel[0], cfg.conv_stride[0]
        a0 = (wave, Pm(FE + "0.conv.weight"), Pm(FE + "0.layer_norm.weight"), Pm(FE + "0.layer_norm.bias"), f.ws)
        ops.w2v_conv0(0, *a0, None, None, None, None, None, None, B, S, T[0], P[0], C, k0, s0)
        ops.w2v_conv0(1, *a0, None, f.act[0], None, None, None, None, B, S, T[0], P[0], C, k0, s0)
        for l in range(1, nl):
            kk, ss = cfg.conv_kernel[l], cfg.conv_stride[l]
            ops.gemm_nt(ops.EPI_GELU_BF16, f.act[l - 1], self.conv_w[l][0], f.act[l], B * P[l], C, kk * C, lda=ss * C,
                        aux_out=f.pre[l] if save else None, ldaux=C)
        # ---- projection, SpecAugment
        ops.w2v_featln_fwd(f.act[-1], Pm(M_ + "feature_projection.layer_norm.weight"), Pm(M_ + "feature_projection.layer_norm.bias"), cfg.eps,
                           f.lnb, f.st_f[0], f.st_f[1], B, Tn, Pn, C)
        f.hidden.zero_()
        ops.gemm_nt_resid_dropout(f.lnb, Pm(M_ + "feature_projection.projection.weight", wb), f.hidden, M, D, C,
                                  Pm(M_ + "feature_projection.projection.bias"), None, dr(SITE_FEATPROJ, cfg.p_featproj))
        if spec is not None:
            mk = np.zeros((B, Pn), dtype=np.uint8)
            mk[:, :Tn] = spec
            f.mask.copy_(torch.from_numpy(mk.reshape(-1)), non_blocking=True)
            ops.w2v_spec_mask_fwd(f.hidden, f.mask, Pm(M_ + "masked_spec_embed"), M, D)
        # ---- positional conv, encoder input
        ops.w2v_pos_stage(f.hidden, f.Xg, B, Tn, Pn, Pp, D, cfg.pos_groups, cfg.pos_k // 2, f.rows_total)
        desc, npb, ntiles, flops, nbytes = f.pos_desc
        ops.gemm_nt_grouped_f32(desc, npb, ntiles, alpha=1.0, beta=0.0, flops=flops, nbytes=nbytes, n64=D // cfg.pos_groups <= 64)
        ctx = None
        x = self._buf(t + "x", (M, D), torch.float32)
        if save:
            ctx = self._ctx_buffers(B, Pn, t)
            ctx.f, ctx.B, ctx.S, ctx.wave, ctx.seed, ctx.spec, ctx.skip = f, B, S, wave, seed, spec is not None, skip
            xb = ctx.xb[0]
        else:
            xb = self._buf(t + "xb", (M, D), torch.bfloat16)
        ops.w2v_pos_finish_fwd(f.hidden, f.conv, Pm(PC + "bias"), Pm(M_ + "encoder.layer_norm.weight"), Pm(M_ + "encoder.layer_norm.bias"),
                               cfg.eps, x, xb, f.ysave, f.st_e[0], f.st_e[1], B, Tn, Pn, Pp, D, dr(SITE_EMB, cfg.p_hidden))
        self.enc_forward(x, xb, ctx, save, B, Pn, f.key_len, dr, skip=skip, tag=t)
        logits, feat = self.head_forward(x, B, Pn, dr(SITE_HEAD, cfg.p_head), f.key_len, ctx)
        return logits, feat, ctx

    def _ctx_buffers(self, B, L, tag):
        key = ("ctx", B, L, tag)
        if key not in self._ws:
            c = types.SimpleNamespace()
            self.enc_alloc_ctx(c, B, L)
            self.head_alloc_ctx(c, B)
            self._ws[key] = c
        return self._ws[key]

    def forward(self, x, only_fc=False, only_feat=False, **kw):
        """Reference-compatible entry (wave2vecv2.py:23-40): x fp32 [B, samples] -> {'logits','feat'}."""
        assert not only_fc, "only_fc is not on the SemiReward hot path"
        logits, feat, _ = self.forward_features(x.to(self.device, torch.float32), None, save=False)
        return feat if only_feat else {"logits": logits, "feat": feat}

    __call__ = forward

    def extract(self, x):
        return self.forward(x, only_feat=True)

    # ---- backward ---------------------------------------------------------------------------------------------------
    def _bwd_front(self, B, f):
        key = ("bwdfront", B, id(f))
        if key in self._ws:
            return self._ws[key]
        cfg = self.cfg
        C, D, G, cg, k, nl = cfg.conv_dim[0], cfg.hidden, cfg.pos_groups, self.cg, cfg.pos_k, len(cfg.conv_kernel)
        P, Pp = f.P, f.Pp
        f32, bf16, dev = torch.float32, torch.bfloat16, self.device
        M = B * P[-1]
        Kc = k * cg
        t = types.SimpleNamespace()
        t.dconv = torch.empty(M, D, dtype=f32, device=dev)
        t.dYg = torch.zeros(G, f.rows_total, cg, dtype=bf16, device=dev)
        t.dWf = torch.empty(G, cg, Kc, dtype=f32, device=dev)
        t.dxpos = torch.zeros(B * Pp + 8, D, dtype=f32, device=dev)
        padl = k - 1 - k // 2
        gb = self.view(PC + "bias", self.grad)
        t.pos_dw = ops.make_group_tn_desc_ld(
            [(ops._pa(t.dYg, (g * f.rows_total + padl) * cg), cg, ops._pa(f.Xg, g * f.rows_total * cg), cg, ops._pa(t.dWf, g * cg * Kc), Kc,
              ops._pa(gb, g * cg), cg, Kc, B * Pp) for g in range(G)], dev)
        t.pos_dx = ops.make_group_desc_ld(
            [(ops._pa(t.dYg, g * f.rows_total * cg), cg, ops._pa(self.pos_Wb, g * cg * Kc), Kc, ops._pa(t.dxpos, g * cg), D, B * Pp, cg, Kc)
             for g in range(G)], dev, bn=64 if cg <= 64 else 128)
        t.gproj = torch.empty(M, D, dtype=bf16, device=dev)
        t.dln = torch.empty(M, C, dtype=bf16, device=dev)
        t.proj_dw = ops.make_group_tn_desc([(t.gproj, f.lnb, self.view(M_ + "feature_projection.projection.weight", self.grad),
                                             self.view(M_ + "feature_projection.projection.bias", self.grad), D, C, M)], dev,
                                           split_k=256, slabs=True)      # 24 tiles over M tokens: token slices through slabs fill the chip (141 us as one slice)
        t.dpre = [None] + [torch.zeros(B * P[l] + 16, C, dtype=bf16, device=dev) for l in range(1, nl)]
        t.dY0 = torch.zeros(B * P[0] + 16, C, dtype=bf16, device=dev)
        t.dcol = [None] + [torch.empty(B * P[l], cfg.conv_kernel[l] * C, dtype=bf16, device=dev) for l in range(1, nl)]
        # weight gradients of the conv layers 1..: ONE grouped launch, the reduction over the frames split into chunks of CH rows (a
        # 512 x 1536 product is 48 tiles; at 100 000 frames it would run on 48 CUs for a millisecond) -> partial products, summed by wgrad_add
        # C a multiple of 256: the launch goes to the 256 x 256 persistent kernel (srhip_gemm_tn_grouped_pp_f32), whose walk is static -- workgroup w
        # takes tiles w, w + 256, ...: the chunk is the shortest (>= 1024 frames, a multiple of 64) that gives at most `rounds` x 256 tiles, so
        # that every workgroup multiplies about the same number of K-tiles (8 clips: 5120 frames per chunk, 256 tiles, one round; the 128 x 128
        # kernel took 785 us for this launch with chunks of 12800: 1168 tiles = 1.5 rounds of its 768 slots at 0.9 us per k-step)
        t.conv_dw_pp = C % 256 == 0
        if t.conv_dw_pp:
            tiles_l = [0] + [(C // 256) * (-(-(cfg.conv_kernel[l] * C) // 256)) for l in range(1, nl)]
            total = sum(tiles_l[l] * (B * P[l]) for l in range(1, nl))                      # tile-frames
            rounds = max(1, -(-total // (256 * 8192)))                                     # at most ~8192 frames per tile and round
            CH = 1024
            while sum(tiles_l[l] * -(-(B * P[l]) // CH) for l in range(1, nl)) > 256 * rounds:
                CH += 64
        else:
            CH = 12800
        t.dW_parts = [0] + [-(-(B * P[l]) // CH) for l in range(1, nl)]
        t.dWr = [None] + [torch.empty(t.dW_parts[l], C, cfg.conv_kernel[l] * C, dtype=f32, device=dev) for l in range(1, nl)]
        probs = []
        for l in range(1, nl):
            kk, ss, R = cfg.conv_kernel[l], cfg.conv_stride[l], B * P[l]
            for c_ in range(t.dW_parts[l]):
                r0 = c_ * CH
                probs.append((ops._pa(t.dpre[l], r0 * C), C, ops._pa(f.act[l - 1], r0 * ss * C), ss * C, ops._pa(t.dWr[l], c_ * C * kk * C), kk * C, 0,
                              C, kk * C, min(CH, R - r0)))
        t.conv_dw = ops.make_group_tn_desc_ld(probs, dev, tile=256 if t.conv_dw_pp else 128)
        t.ws2 = torch.zeros(B, C, 2, dtype=torch.float64, device=dev)
        self._ws[key] = t
        return t

    def backward(self, ctx, dlogits):
        """Accumulates d(loss)/d(params) into ``self.grad`` given dlogits fp32 [B, C] of a save=True forward."""
        cfg, f = self.cfg, ctx.f
        D, C, nl = cfg.hidden, cfg.conv_dim[0], len(cfg.conv_kernel)
        B, S, seed = ctx.B, ctx.S, ctx.seed
        T, P, Pp = f.T, f.P, f.Pp
        Tn, Pn = T[-1], P[-1]
        M = B * Pn
        dr = (lambda site, p: ops.Drop(seed, site, p) if p > 0 else None) if seed is not None else (lambda site, p: None)
        Pm, G = self.p, (lambda n: self.p(n, self.grad))
        t = self._bwd_front(B, f)
        dx = self._buf("b_dx", (M, D), torch.float32)
        self.head_backward(ctx, dlogits, dx, B, Pn, dr(SITE_HEAD, cfg.p_head), f.key_len)
        self.enc_backward(dx, ctx, B, Pn, f.key_len, dr, skip=ctx.skip)
        # ---- encoder input: LayerNorm / dropout, positional conv
        ops.w2v_pos_finish_bwd(dx, f.ysave, f.conv, Pm(PC + "bias"), f.st_e[0], f.st_e[1], Pm(M_ + "encoder.layer_norm.weight"), t.dconv,
                               G(M_ + "encoder.layer_norm.weight"), G(M_ + "encoder.layer_norm.bias"), B, Tn, Pn, Pp, D, dr(SITE_EMB, cfg.p_hidden))
        ops.w2v_pos_stage(t.dconv, t.dYg, B, Tn, Pn, Pp, D, cfg.pos_groups, cfg.pos_k - 1 - cfg.pos_k // 2, f.rows_total)
        desc, npb, ntiles, flops, nbytes = t.pos_dw
        ops.gemm_tn_grouped_f32(desc, npb, ntiles, alpha=1.0, beta=0.0, flops=flops, nbytes=nbytes)
        ops.w2v_weightnorm_bwd(t.dWf, Pm(PC + "parametrizations.weight.original1"), Pm(PC + "parametrizations.weight.original0"), self.pos_norms,
                               G(PC + "parametrizations.weight.original1"), G(PC + "parametrizations.weight.original0"), D, cfg.pos_groups, cfg.pos_k)
        desc, npb, ntiles, flops, nbytes = t.pos_dx
        ops.gemm_nt_grouped_f32(desc, npb, ntiles, alpha=1.0, beta=0.0, flops=flops, nbytes=nbytes, n64=D // cfg.pos_groups <= 64)
        # ---- SpecAugment, projection (dropout' folded into the bf16 cast of the branch gradient)
        ops.w2v_spec_mask_bwd(dx, t.dxpos, f.mask if ctx.spec else None, G(M_ + "masked_spec_embed") if ctx.spec else None, B, Tn, Pn, Pp, D)
        ops.dropout_cast(dx, t.gproj, M * D, dr(SITE_FEATPROJ, cfg.p_featproj))
        desc, npb, ntiles, flops, nbytes = t.proj_dw
        ops.gemm_tn_grouped_f32(desc, npb, ntiles, alpha=1.0, beta=1.0, flops=flops, nbytes=nbytes)
        ops.gemm_nt(ops.EPI_BF16, t.gproj, self.projT, t.dln, M, C, D)
        ops.w2v_featln_bwd(t.dln, f.act[-1], f.pre[-1], f.st_f[0], f.st_f[1], Pm(M_ + "feature_projection.layer_norm.weight"), t.dpre[-1],
                           G(M_ + "feature_projection.layer_norm.weight"), G(M_ + "feature_projection.layer_norm.bias"), B, Tn, Pn, C)
        # ---- conv layers n-1 .. 1: dW from the overlapping-row operand, dX through the transposed filter + fold
        for l in range(nl - 1, 0, -1):
            kk, ss = cfg.conv_kernel[l], cfg.conv_stride[l]
            ops.gemm_nt(ops.EPI_BF16, t.dpre[l], self.conv_w[l][1], t.dcol[l], B * P[l], kk * C, C)
            ops.w2v_col2im_dgelu(t.dcol[l], f.pre[l - 1] if l > 1 else None, t.dpre[l - 1] if l > 1 else t.dY0, B, P[l], P[l - 1], C, kk, ss)
        desc, npb, ntiles, flops, nbytes = t.conv_dw
        ops.gemm_tn_grouped_f32(desc, npb, ntiles, alpha=1.0, beta=0.0, flops=flops, nbytes=nbytes, pp=t.conv_dw_pp)
        for l in range(1, nl):
            ops.w2v_conv_wgrad_add(t.dWr[l], G(FE + "%d.conv.weight" % l), C, C, cfg.conv_kernel[l], t.dW_parts[l])
        # ---- conv layer 0 + GroupNorm
        t.ws2.zero_()
        k0, s0 = cfg.conv_kernel[0], cfg.conv_stride[0]
        a0 = (ctx.wave, Pm(FE + "0.conv.weight"), Pm(FE + "0.layer_norm.weight"), Pm(FE + "0.layer_norm.bias"), f.ws, t.ws2)
        ops.w2v_conv0(2, *a0, None, t.dY0, None, G(FE + "0.layer_norm.weight"), G(FE + "0.layer_norm.bias"), B, S, T[0], P[0], C, k0, s0)
        ops.w2v_conv0(3, *a0, None, t.dY0, G(FE + "0.conv.weight"), None, None, B, S, T[0], P[0], C, k0, s0)


# ---- builders with the reference's names (wave2vecv2.py:58-60); the pretrained checkpoint needs the network -> random init -------------
def _build(num_classes, kw, **cfg):
    kw = {k: v for k, v in kw.items() if k not in ("pretrained", "pretrained_path")}
    device = kw.pop("device", "cuda")
    m = ClassificationWave2Vec(W2vConfig(num_classes=num_classes, **cfg), device=device)
    m.init_weights(kw.pop("seed", 0))
    return m


def wave2vecv2_base(num_classes=2, **kw):
    return _build(num_classes, kw)


def wave2vecv2_tiny_test(num_classes=4, **kw):
    return _build(num_classes, kw, hidden=128, layers=2, heads=2, inter=256, conv_dim=(128, 128, 128), conv_kernel=(10, 3, 2),
                  conv_stride=(5, 2, 2), pos_k=16, pos_groups=4)
