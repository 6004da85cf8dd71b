"""Post-LN transformer encoder stack shared by the BERT and Wav2Vec2 engines (the encoders the reference obtains from ``transformers``:
BertLayer behind semilearn/nets/bert/bert.py:34, Wav2Vec2EncoderLayer behind semilearn/nets/wave2vecv2/wave2vecv2.py:44 -- the same
arithmetic up to parameter names, LayerNorm eps and one extra dropout after the GELU in Wav2Vec2).  Per layer:

    qkv  = x_bf16 . Wqkv^T + b                       srhip_gemm_nt            (q | k | v packed: one product, N = 3D)
    ctx  = softmax(q k^T / 8 + key mask) v           srhip_attn_masked_fwd    (per-sequence key length, dropout on the probabilities)
    y1   = x + dropout(ctx . Wo^T + b)               srhip_gemm_nt_resid_dropout
    x    = LayerNorm(y1)          (fp32 + bf16)      srhip_postln_fwd
    h    = [dropout] GELU(x_bf16 . W1^T + b)         srhip_gemm_nt / srhip_gemm_nt_dropout (GELU epilogue)
    y2   = x + dropout(h . W2^T + b)                 srhip_gemm_nt_resid_dropout
    x    = LayerNorm(y2)                             srhip_postln_fwd

and a hand-written backward (post-LN: the gradient of a LayerNorm input feeds the residual path in fp32 and, dropout-masked, the branch's
dX / dW products in bf16).  A host class mixes this in and provides: ``cfg`` (hidden, inter, heads, layers, eps), ``p`` / ``view`` /
``flat_bf16`` / ``grad`` / ``offsets``, ``enc_names(i)`` (parameter names of layer i) and ``enc_p`` (dropout probabilities).
LayerDrop (Wav2Vec2, train mode): ``skip[i]`` leaves layer i out of forward and backward.
"""
import torch

from .. import ops
from .surface import ModuleSurface

_LAZY_LN = True            # inference rows: a LayerNorm between two sub-layers is applied by the residual GEMM that follows it (enc_forward)
SITE_PROBS, SITE_ATTN_OUT, SITE_FFN_OUT, SITE_ACT = 0, 1, 2, 3
LN_REP = 16                # partial copies of a LayerNorm's dgamma / dbeta in the backward (ops.postln_bwd_part)


class PostLNEncoderMixin(ModuleSurface):
    # ---- packed q|k|v views (the three tensors are adjacent in the flat block) ---------------------------------
    def packed_qkv(self, i, buf=None):
        D = self.cfg.hidden
        nm = self.enc_names(i)
        ow, ob = self.offsets[nm["q_w"]][0], self.offsets[nm["q_b"]][0]
        b = self.flat if buf is None else buf
        return b[ow:ow + 3 * D * D].view(3 * D, D), b[ob:ob + 3 * D]

    def enc_alloc_wT(self):
        D, I, bf16 = self.cfg.hidden, self.cfg.inter, torch.bfloat16
        self.wT = [dict(qkv=torch.zeros(D, 3 * D, dtype=bf16, device=self.device), o=torch.zeros(D, D, dtype=bf16, device=self.device),
                        w1=torch.zeros(D, I, dtype=bf16, device=self.device), w2=torch.zeros(I, D, dtype=bf16, device=self.device))
                   for _ in range(self.cfg.layers)]

    def enc_transpose_items(self):
        items, D, I = [], self.cfg.hidden, self.cfg.inter
        for i in range(self.cfg.layers):
            nm, t = self.enc_names(i), self.wT[i]
            items += [(self.packed_qkv(i)[0], True, D, t["qkv"], 3 * D, 3 * D, 3 * D, D, False),
                      (self.p(nm["o_w"]), True, D, t["o"], D, D, D, D, False),
                      (self.p(nm["w1"]), True, D, t["w1"], I, I, I, D, False),
                      (self.p(nm["w2"]), True, I, t["w2"], D, D, D, I, False)]
        return items

    def enc_alloc_ctx(self, c, B, L):
        cfg = self.cfg
        D, I, H, M = cfg.hidden, cfg.inter, cfg.heads, B * L
        f32, bf16 = torch.float32, torch.bfloat16
        mk = lambda shape, dt: [torch.empty(shape, dtype=dt, device=self.device) for _ in range(cfg.layers)]   # noqa: E731
        c.xb = mk((M, D), bf16) + [torch.empty(M, D, dtype=bf16, device=self.device)]     # layer inputs (bf16): X operand of dWqkv
        c.qkv, c.ao, c.lse = mk((M, 3 * D), bf16), mk((M, D), bf16), mk((B, H, L), f32)
        c.y1, c.st1, c.xbm = mk((M, D), f32), mk((2, M), f32), mk((M, D), bf16)
        c.pre, c.h, c.y2, c.st2 = mk((M, I), bf16), mk((M, I), bf16), mk((M, D), f32), mk((2, M), f32)

    def enc_forward(self, x, xb, ctx, save, B, L, key_len, dr, skip=None, tag="i"):
        """x fp32 [M, D] (updated in place), xb bf16 [M, D] = the same values (first layer's GEMM operand; ctx.xb[0] when save).
        dr(site, p) -> ops.Drop or None.  Returns the bf16 copy of the final x."""
        cfg, P, wb, pr = self.cfg, self.p, self.flat_bf16, self.enc_p
        D, I, H = cfg.hidden, cfg.inter, cfg.heads
        M = B * L
        bf16 = torch.bfloat16
        if not save:
            qkv, ao = self._buf(tag + "qkv", (M, 3 * D), bf16), self._buf(tag + "ao", (M, D), bf16)
            hbuf = self._buf(tag + "h", (M, I), bf16)
            st = [self._buf(tag + "lnst%d" % j, (2, M), torch.float32) for j in range(2)]
        scale = 64 ** -0.5
        # Rows without a backward never materialise the fp32 output of a LayerNorm between two sub-layers: postln_fwd writes the bf16 GEMM operand
        # and the row statistics (6 instead of 10 bytes per element of a launch that is 11 % of the BERT leg's kernel time), `x` keeps the
        # PRE-LayerNorm sums, and the next residual GEMM applies the LayerNorm to the residual it reads (gemm_nt_resid_ln_dropout).
        # `pend` = (mean, rstd, gamma, beta) of the LayerNorm still owed to `x`; the last executed layer writes the real thing.
        pend = None
        live = [i for i in range(cfg.layers) if not (skip is not None and skip[i])]
        lazy = (not save) and _LAZY_LN

        def resid(a_op, w, out_y, bias_, k_dim, resid_src, site, p_):
            if pend is not None and resid_src is None:
                ops.gemm_nt_resid_ln_dropout(a_op, w, out_y, M, D, k_dim, bias_, pend[0], pend[1], pend[2], pend[3], dr(site, p_))
            else:
                ops.gemm_nt_resid_dropout(a_op, w, out_y, M, D, k_dim, bias_, resid_src, dr(site, p_))

        for i in range(cfg.layers):
            if skip is not None and skip[i]:
                if save:                             # the next layer's X operand is this layer's input
                    ctx.xb[i + 1].copy_(xb)
                    xb = ctx.xb[i + 1]
                continue
            nm = self.enc_names(i)
            Wqkv, bqkv = self.packed_qkv(i, wb)[0], self.packed_qkv(i)[1]
            if save:
                qkv, ao = ctx.qkv[i], ctx.ao[i]
            ops.gemm_nt(ops.EPI_BF16, xb, Wqkv, qkv, M, 3 * D, D, bias=bqkv)
            ops.attn_masked_fwd(qkv, ao, ctx.lse[i] if save else None, key_len, B, L, H, scale, dr(4 * i + SITE_PROBS, pr["attn"]))
            y1 = ctx.y1[i] if save else x
            resid(ao, P(nm["o_w"], wb), y1, P(nm["o_b"]), D, x if save else None, 4 * i + SITE_ATTN_OUT, pr["hidden"])
            xbm = ctx.xbm[i] if save else xb
            if lazy:
                ops.postln_fwd(y1, P(nm["ln1_w"]), P(nm["ln1_b"]), cfg.eps, None, xbm, st[0][0], st[0][1], M, D)
                pend = (st[0][0], st[0][1], P(nm["ln1_w"]), P(nm["ln1_b"]))
            else:
                ops.postln_fwd(y1, P(nm["ln1_w"]), P(nm["ln1_b"]), cfg.eps, x, xbm, ctx.st1[i][0] if save else None, ctx.st1[i][1] if save else None, M, D)
            h = ctx.h[i] if save else hbuf
            da = dr(4 * i + SITE_ACT, pr["act"])
            if da is None:
                ops.gemm_nt(ops.EPI_GELU_BF16, xbm, P(nm["w1"], wb), h, M, I, D, bias=P(nm["b1"]), aux_out=ctx.pre[i] if save else None, ldaux=I)
            else:
                ops.gemm_nt_dropout(ops.EPI_GELU_BF16, xbm, P(nm["w1"], wb), h, M, I, D, da, bias=P(nm["b1"]),
                                    aux_out=ctx.pre[i] if save else None, ldaux=I)
            y2 = ctx.y2[i] if save else x
            resid(h, P(nm["w2"], wb), y2, P(nm["b2"]), I, x if save else None, 4 * i + SITE_FFN_OUT, pr["hidden"])
            xb = ctx.xb[i + 1] if save else xb
            if lazy and i != live[-1]:
                ops.postln_fwd(y2, P(nm["ln2_w"]), P(nm["ln2_b"]), cfg.eps, None, xb, st[1][0], st[1][1], M, D)
                pend = (st[1][0], st[1][1], P(nm["ln2_w"]), P(nm["ln2_b"]))
            else:
                ops.postln_fwd(y2, P(nm["ln2_w"]), P(nm["ln2_b"]), cfg.eps, x, xb, ctx.st2[i][0] if save else None, ctx.st2[i][1] if save else None, M, D)
                pend = None
        return xb

    def enc_bwd_plan(self, M, ctx):
        """Output-gradient buffers (bf16 A operands of dW = dY^T X), ONE SET PER LAYER, + the problem list of every layer for the weight-gradient
        launch (row-major operands, bias gradients summed on the way).  The four dY of a layer used to be shared by all layers, which forced one
        weight-gradient launch per layer inside the chain -- 108 tiles of 256 x 256 on 256 CUs (BERT: 336 us per layer, 13 % of the step).  Kept
        per layer (BERT: 113 MB x 12 of 288 GB) they feed ONE launch behind the layer loop: 1 296 tiles in a persistent walk
        (srhip_gemm_tn_grouped_pp_f32: 113 us per layer, tools/gemm_tn_pp_bench.py)."""
        key = ("encbwd", M, id(ctx))
        if key in self._ws:
            return self._ws[key]
        cfg = self.cfg
        D, I = cfg.hidden, cfg.inter
        mk = lambda c: [torch.empty(M, c, dtype=torch.bfloat16, device=self.device) for _ in range(cfg.layers)]   # noqa: E731
        T = dict(g2=mk(D), dpre=mk(I), g1=mk(D), dqkv=mk(3 * D), dao=torch.empty(M, D, dtype=torch.bfloat16, device=self.device),
                 probs=[], tables={})
        G = lambda n: self.view(n, self.grad)   # noqa: E731
        for i in range(cfg.layers):
            nm = self.enc_names(i)
            gw, gb = self.packed_qkv(i, self.grad)
            T["probs"].append([(T["g2"][i], ctx.h[i], G(nm["w2"]), G(nm["b2"]), D, I, M), (T["dpre"][i], ctx.xbm[i], G(nm["w1"]), G(nm["b1"]), I, D, M),
                               (T["g1"][i], ctx.ao[i], G(nm["o_w"]), G(nm["o_b"]), D, D, M), (T["dqkv"][i], ctx.xb[i], gw, gb, 3 * D, D, M)])
        # LayerNorm affine gradients go through LN_REP partial copies per LayerNorm (ops.postln_bwd_part), folded after the layer loop
        T["ln_part"] = torch.zeros(2 * cfg.layers, LN_REP, 2, D, dtype=torch.float32, device=self.device)
        T["ln_desc"] = ops.make_ln_reduce_desc([(G(self.enc_names(i)[k + "_w"]), G(self.enc_names(i)[k + "_b"]))
                                                for i in range(cfg.layers) for k in ("ln1", "ln2")], self.device)
        self._ws[key] = T
        return T

    def enc_dw_table(self, T, skip):
        """Descriptor table of the weight-gradient launch over the layers LayerDrop left in (one table per pattern, built when it first occurs)."""
        pat = tuple(bool(s) for s in skip) if skip is not None else ()
        if pat not in T["tables"]:
            probs = [pr for i, lp in enumerate(T["probs"]) if not (pat and pat[i]) for pr in lp]
            pp = bool(probs) and ops.tn_pp_efficiency(probs) >= 0.85
            # A table is built INSIDE a step whenever LayerDrop leaves out a set of layers not seen before (most of the first steps of a run): its
            # upload must not stall the host (ops.TableStager), and it allocates nothing -- the token slices that balance the persistent
            # kernel's last round meet through atomics here (slabs would be a 50-MB allocation per pattern; measured equal, 29.57 vs 29.64 ms)
            if T.get("stager") is None:
                T["stager"] = ops.TableStager(64 * (4 * len(T["probs"]) + 64))
            T["tables"][pat] = (ops.make_group_tn_desc(probs, self.device, tile=256 if pp else 128, stager=T["stager"]), pp) if probs else None
        return T["tables"][pat]

    def enc_backward(self, dx, ctx, B, L, key_len, dr, skip=None):
        """dx fp32 [M, D]: gradient w.r.t. the encoder output on entry, w.r.t. its input on return (in place)."""
        cfg, P, pr = self.cfg, self.p, self.enc_p
        D, I, H = cfg.hidden, cfg.inter, cfg.heads
        M = B * L
        G = lambda n: self.p(n, self.grad)   # noqa: E731
        delta = self._buf("b_delta", (B, H, L), torch.float32)
        T = self.enc_bwd_plan(M, ctx)
        scale = 64 ** -0.5
        for i in reversed(range(cfg.layers)):
            if skip is not None and skip[i]:
                continue
            nm, wT = self.enc_names(i), self.wT[i]
            # ---- FFN: x_out = LN(y2), y2 = x_mid + dropout(W2 [dropout] gelu(W1 x_mid))
            ops.postln_bwd_part(dx, ctx.y2[i], ctx.st2[i][0], ctx.st2[i][1], P(nm["ln2_w"]), dx, T["g2"][i], T["ln_part"][2 * i + 1], LN_REP, M, D,
                                dr(4 * i + SITE_FFN_OUT, pr["hidden"]))
            da = dr(4 * i + SITE_ACT, pr["act"])
            if da is None:
                ops.gemm_nt(ops.EPI_DGELU_BF16, T["g2"][i], wT["w2"], T["dpre"][i], M, I, D, aux_in=ctx.pre[i], ldaux=I)
            else:
                ops.gemm_nt_dropout(ops.EPI_DGELU_BF16, T["g2"][i], wT["w2"], T["dpre"][i], M, I, D, da, aux_in=ctx.pre[i], ldaux=I)
            ops.gemm_nt(ops.EPI_RESID_F32, T["dpre"][i], wT["w1"], dx, M, D, I)
            # ---- attention: x_mid = LN(y1), y1 = x_in + dropout(Wo attn(qkv(x_in)))
            ops.postln_bwd_part(dx, ctx.y1[i], ctx.st1[i][0], ctx.st1[i][1], P(nm["ln1_w"]), dx, T["g1"][i], T["ln_part"][2 * i], LN_REP, M, D,
                                dr(4 * i + SITE_ATTN_OUT, pr["hidden"]))
            ops.gemm_nt(ops.EPI_BF16, T["g1"][i], wT["o"], T["dao"], M, D, D)
            ops.attn_masked_bwd(ctx.qkv[i], ctx.ao[i], T["dao"], ctx.lse[i], T["dqkv"][i], delta, key_len, B, L, H, scale,
                                dr(4 * i + SITE_PROBS, pr["attn"]))
            ops.gemm_nt(ops.EPI_RESID_F32, T["dqkv"][i], wT["qkv"], dx, M, D, 3 * D)
        tab = self.enc_dw_table(T, skip)
        if tab is not None:
            (desc, npb, ntiles, flops, nbytes), pp = tab
            ops.gemm_tn_grouped_f32(desc, npb, ntiles, alpha=1.0, beta=1.0, flops=flops, nbytes=nbytes, pp=pp)
        ops.ln_grad_reduce(T["ln_desc"], T["ln_part"], 2 * cfg.layers, LN_REP, D)

    # ---- mean-pool + 2-layer classifier head (bert.py:16-20,36-37 / wave2vecv2.py:17-21,46-48) ---------------------------
    def head_forward(self, x, B, L, drop, seq_len, c=None):
        D, C, f32, P = self.cfg.hidden, self.cfg.num_classes, torch.float32, self.p
        mkb = lambda: torch.empty(B, D, dtype=f32, device=self.device)   # noqa: E731
        feat, hpre, hact = (c.feat, c.hpre, c.hact) if c is not None else (mkb(), mkb(), mkb())
        logits = torch.empty(B, C, dtype=f32, device=self.device)
        ops.meanpool_fwd(x, feat, B, L, D, drop, seq_len)
        ops.fc_fwd(feat, P("classifier.0.weight"), P("classifier.0.bias"), hpre, B, D, D)
        ops.gelu_f32(hpre, hact, B * D)
        ops.fc_fwd(hact, P("classifier.2.weight"), P("classifier.2.bias"), logits, B, D, C)
        return logits, (feat.clone() if c is not None else feat)

    def head_backward(self, c, dlogits, dx, B, L, drop, seq_len):
        D, C, f32, P = self.cfg.hidden, self.cfg.num_classes, torch.float32, self.p
        G = lambda n: self.p(n, self.grad)   # noqa: E731
        dhact, dhpre, dfeat = self._buf("b_dhact", (B, D), f32), self._buf("b_dhpre", (B, D), f32), self._buf("b_dfeat", (B, D), f32)
        ops.fc_bwd(dlogits, c.hact, P("classifier.2.weight"), dhact, G("classifier.2.weight"), G("classifier.2.bias"), B, D, C)
        ops.gelu_bwd_f32(dhact, c.hpre, dhpre, B * D)
        ops.fc_bwd(dhpre, c.feat, P("classifier.0.weight"), dfeat, G("classifier.0.weight"), G("classifier.0.bias"), B, D, D)
        ops.meanpool_bwd(dfeat, dx, B, L, D, drop, seq_len)

    def head_alloc_ctx(self, c, B):
        D, f32 = self.cfg.hidden, torch.float32
        c.feat, c.hpre, c.hact = (torch.empty(B, D, dtype=f32, device=self.device) for _ in range(3))
