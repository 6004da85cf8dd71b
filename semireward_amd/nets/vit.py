"""ViT backbone engine on libsrhip: forward / backward over a flat parameter block.

Mirrors the reference's ``semilearn/nets/vit/vit.py`` plugin surface (class name, builder names,
``state_dict`` keys, ``{'logits','feat'}`` result dict, ``no_weight_decay`` / ``group_matcher``) but not
its implementation: there is no nn.Module, no autograd.  Parameters, gradients, bf16 operand copies
and Adam moments are each ONE contiguous device buffer (the layout is the reference's
``named_parameters()`` order, so a reference checkpoint maps 1:1), every layer is a short list of
HIP kernel launches, and the backward is hand-written.

Data layout in HBM (B images, N tokens, D channels, M = B*N rows):
  residual stream x         fp32 [M, D]      (in place when nothing is saved for a backward)
  LN output / attn out      bf16 [M, D]      GEMM A-operands
  qkv                       bf16 [M, 3D]     exactly as the qkv Linear writes it; attention indexes heads in place
  MLP hidden                bf16 [M, 4D]
  weights                   bf16 [out, in]   (+ transposed bf16 copies for the dX products)
"""
import math
import types

import torch

from .. import ops
from .surface import ModuleSurface


class VitConfig:
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, drop_path_rate=0.0, eps=1e-6):
        self.img_size, self.patch_size, self.in_chans, self.num_classes = img_size, patch_size, in_chans, num_classes
        self.embed_dim, self.depth, self.num_heads, self.mlp_ratio = embed_dim, depth, num_heads, mlp_ratio
        self.drop_path_rate, self.eps = drop_path_rate, eps
        assert embed_dim % num_heads == 0 and embed_dim // num_heads == 64, "libsrhip attention is built for head_dim 64"
        assert embed_dim in (128, 384, 768)

    @property
    def num_tokens(self):
        return (self.img_size // self.patch_size) ** 2 + 1

    @property
    def hidden(self):
        return int(self.embed_dim * self.mlp_ratio)


def param_names_shapes(cfg):
    """named_parameters() order of the reference VisionTransformer (vit.py:228-275)."""
    D, Hd, p, C = cfg.embed_dim, cfg.hidden, cfg.patch_size, cfg.num_classes
    out = [("cls_token", (1, 1, D)), ("pos_embed", (1, cfg.num_tokens, D)),
           ("patch_embed.proj.weight", (D, cfg.in_chans, p, p)), ("patch_embed.proj.bias", (D,))]
    for i in range(cfg.depth):
        b = "blocks.%d." % i
        out += [(b + "norm1.weight", (D,)), (b + "norm1.bias", (D,)),
                (b + "attn.qkv.weight", (3 * D, D)), (b + "attn.qkv.bias", (3 * D,)),
                (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,)),
                (b + "norm2.weight", (D,)), (b + "norm2.bias", (D,)),
                (b + "mlp.fc1.weight", (Hd, D)), (b + "mlp.fc1.bias", (Hd,)),
                (b + "mlp.fc2.weight", (D, Hd)), (b + "mlp.fc2.bias", (D,))]
    out += [("norm.weight", (D,)), ("norm.bias", (D,)), ("head.weight", (C, D)), ("head.bias", (C,))]
    return out


DW_GROUPS = 3              # layer groups of the weight-gradient launch when a grad_ready_cb is installed (data parallel)
LN_REP = 16                # partial copies of a LayerNorm's dgamma / dbeta in the backward (ops.layernorm_bwd_part)
# Which launches the rows without a backward take (module constants: the tests flip them to compare the fused chain with the launches it replaces)
_FUSED_MLP = True          # LN2 + fc1 + GELU + fc2 + residual as one launch
_FUSED_ATTN = True         # qkv Linear + attention as one launch
_FUSED_PROJ = True         # attention projection + residual inside the fused MLP launch
_FUSED_NEXT_LN = True      # ... which then also writes the next block's norm1 output
# the fused kernel owns a CU per 128-row tile for ~90 us whatever the launch size: below ~half a chip of tiles (the 8 inference images of the
# pre-start_timing regime = 17 tiles) LayerNorm + two 64x64-tiled GEMMs spread over all CUs are faster
_FUSED_MLP_MIN_ROWS = 16384


def _round_up(a, b):
    return (a + b - 1) // b * b


class FwdContext:
    """Activations kept by a ``save=True`` forward for the hand-written backward."""
    __slots__ = ("B", "img", "img_index", "dp", "xs", "xmid", "ln1", "ln2", "qkv", "ao", "lse", "pre", "h", "st1", "st2",
                 "feat", "xhat", "rstd")


class VisionTransformer(ModuleSurface):
    GEMM_WEIGHTS = ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")
    rows_independent = True       # LayerNorm only: a row's outputs do not depend on which other rows share the launch
    scatter_outputs = True        # forward_features(out=...) writes logits / features at the caller's row numbers (no index_copy_)
    droppath_by_cols = True       # make_droppath(cols=...) lays the DropPath table out in the caller's column order (no index_select)

    def __init__(self, cfg=None, device="cuda", **kw):
        self.cfg = cfg if cfg is not None else VitConfig(**kw)
        cfg = self.cfg
        self.device = torch.device(device)
        self.names_shapes = param_names_shapes(cfg)
        self.grad_ready_cb = None          # callable(lo, hi) or None: see backward() / distributed.DataParallel.install_overlap
        self.offsets, o = {}, 0
        for n, s in self.names_shapes:
            self.offsets[n] = (o, s)
            o += int(torch.Size(s).numel())
        # 2-D GEMM weights must start on a 16-byte boundary in the bf16 copy (8 elements)
        assert all(v[0] % 8 == 0 for k, v in self.offsets.items() if len(v[1]) == 2)
        self.numel = o
        self.flat = torch.zeros(o, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(o, dtype=torch.float32, device=self.device)
        self.flat_bf16 = torch.zeros(o, dtype=torch.bfloat16, device=self.device)
        self.wT = {}
        self._wT_stale = True
        for i in range(cfg.depth):
            for w in self.GEMM_WEIGHTS:
                n = "blocks.%d.%s" % (i, w)
                r, c = self.offsets[n][1]
                self.wT[n] = torch.zeros(c, r, dtype=torch.bfloat16, device=self.device)
        self.dp_probs = torch.linspace(0, cfg.drop_path_rate, cfg.depth).to(self.device)    # vit.py:247-249
        self.training = True
        self._ws = {}
        self._wT_desc = None
        self._rng_calls = 0
        self.seed = 0

    # ---- parameter plumbing ---------------------------------------------------------------------
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
        """The reference VisionTransformer has no init function: torch defaults -- nn.Linear / Conv2d kaiming_uniform(a = sqrt 5) =
        U(+-1/sqrt(fan_in)) for weight and bias, LayerNorm 1 / 0, cls_token and pos_embed zeros (vit.py:241-244)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        sd, fan = {}, {}
        for n, s_ in self.names_shapes:
            if n in ("cls_token", "pos_embed"):
                sd[n] = torch.zeros(s_)
            elif "norm" in n.split(".")[-2]:
                sd[n] = torch.ones(s_) if n.endswith("weight") else torch.zeros(s_)
            else:
                if n.endswith("weight"):
                    fan[n[:-7]] = int(torch.Size(s_[1:]).numel())
                bound = 1.0 / (fan[n.rsplit(".", 1)[0]] ** 0.5)
                sd[n] = (torch.rand(s_, generator=g) * 2 - 1) * bound
        self.load_state_dict(sd)

    def refresh_operands(self):
        """bf16 operand copy of the whole block + transposed GEMM weights (after any parameter change)."""
        ops.cast_f32_bf16(self.flat, self.flat_bf16, self.numel)
        self.refresh_transposed()

    lazy_transposed = True   # the optimizer only marks the transposed weight copies stale (ensure_transposed)

    def ensure_transposed(self):
        """The transposed bf16 weight copies are operands of the BACKWARD only (dX products).  After an optimizer step they are refreshed
        where it costs nothing -- the step's second stream, before the gradient rows' forward (srflexmatch._forward_plan) -- instead of at
        the end of the optimizer step, on the critical path (40 us per step); backward() calls this again as the safety net."""
        if self._wT_stale:
            self.refresh_transposed()

    def refresh_transposed(self):
        """W [out,in] fp32 -> W^T [in,out] bf16 for all 4*depth GEMM weights: one batched launch."""
        self._wT_stale = False
        if self._wT_desc is None:
            items = []
            for n, t in self.wT.items():
                r, c = self.offsets[n][1]
                items.append((self.p(n), True, c, t, r, r, r, c, False))
            self._wT_desc = ops.make_transpose_desc(items, self.device)
        ops.transpose_batched(*self._wT_desc)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def zero_grad(self):
        self.grad.zero_()

    # ---- workspaces -------------------------------------------------------------------------------
    def _buf(self, key, shape, dtype):
        t = self._ws.get(key)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    def make_droppath(self, B, cols=None):
        """timm DropPath scales [depth, 2, B] for one forward (vit.py:148,161); cols (int64 device tensor): only those columns of the draw, in
        that order ([depth, 2, len(cols)])."""
        dp = torch.empty(self.cfg.depth, 2, B if cols is None else cols.numel(), dtype=torch.float32, device=self.device)
        self._rng_calls += 1
        sc = getattr(self, "step_scalars", None)
        if sc is not None:
            # core/stepgraph.py: the step's seed base ((seed << 32) + the draw counter at the start of the step) sits in device memory; this call
            # adds its own number inside the step -- the same 64-bit seed as below, from a launch that can be replayed
            self._step_draws = getattr(self, "_step_draws", 0) + 1
            ops.droppath_fill(dp, self.dp_probs, self.cfg.depth, B, self._step_draws, cols=cols, seed_dev=sc.seed_ptr)
        else:
            ops.droppath_fill(dp, self.dp_probs, self.cfg.depth, B, (self.seed << 32) + self._rng_calls, cols=cols)
        return dp

    def _ctx_buffers(self, B):
        """Activation buffers of a save=True forward.  Persistent per batch size (one live context per size), so the
        batched-transpose / grouped-GEMM descriptor tables that point into them are built once."""
        key = ("ctx", B)
        if key in self._ws:
            return self._ws[key]
        cfg = self.cfg
        D, N, H, Hd = cfg.embed_dim, cfg.num_tokens, cfg.num_heads, cfg.hidden
        M = B * N
        f32, bf16 = torch.float32, torch.bfloat16
        mk = lambda shape, dt: [torch.empty(shape, dtype=dt, device=self.device) for _ in range(cfg.depth)]   # noqa: E731
        ctx = FwdContext()
        ctx.xs = mk((M, D), f32) + [torch.empty(M, D, dtype=f32, device=self.device)]
        ctx.xmid, ctx.ln1, ctx.ln2 = mk((M, D), f32), mk((M, D), bf16), mk((M, D), bf16)
        ctx.qkv, ctx.ao, ctx.pre = mk((M, 3 * D), bf16), mk((M, D), bf16), mk((M, Hd), bf16)
        ctx.h = mk((M, Hd), bf16)                # GELU output: the X operand of dW_fc2
        ctx.lse = mk((B, H, N), f32)
        ctx.st1 = [(t[0], t[1]) for t in mk((2, M), f32)]            # (mean, rstd) rows as ready views: indexing costs host time per launch
        ctx.st2 = [(t[0], t[1]) for t in mk((2, M), f32)]
        ctx.xhat = torch.empty(B, D, dtype=f32, device=self.device)
        ctx.rstd = torch.empty(B, dtype=f32, device=self.device)
        self._ws[key] = ctx
        return ctx

    # ---- forward ----------------------------------------------------------------------------------
    def forward_features(self, img, img_index=None, droppath=None, save=False, B=None, buftag="", out=None):
        """img fp32 [n_img, C, H, W]; img_index int32 [B] (optional gather); droppath fp32 [depth,2,B] or None.
        Returns (logits [B,C], feat [B,D], ctx or None)."""
        cfg = self.cfg
        D, N, H, Hd, C = cfg.embed_dim, cfg.num_tokens, cfg.num_heads, cfg.hidden, cfg.num_classes
        B = int(img_index.numel()) if img_index is not None else (B or img.shape[0])
        M = B * N
        f32, bf16 = torch.float32, torch.bfloat16
        tag = ("s" if save else "i") + buftag        # buftag: a second inference launch train on another stream needs its own workspaces
        ctx = None
        if save:
            ctx = self._ctx_buffers(B)
            ctx.B, ctx.img, ctx.img_index, ctx.dp = B, img, img_index, droppath
            x = ctx.xs[0]
        else:
            x = self._buf(tag + "x", (M, D), f32)
            ln = self._buf(tag + "ln", (M, D), bf16)
            qkv = self._buf(tag + "qkv", (M, 3 * D), bf16)
            ao = self._buf(tag + "ao", (M, D), bf16)
        # rows without a backward run LN2 + fc1 + GELU + fc2 + residual as ONE kernel (ViT-S width; SRHIP_FUSED_MLP=0: off)
        fused_mlp = (not save) and D == 384 and Hd % 128 == 0 and Hd <= 4096 and M >= _FUSED_MLP_MIN_ROWS and _FUSED_MLP
        hbuf = None if (fused_mlp or save) else self._buf(tag + "h", (M, Hd), bf16)
        fused_attn = (not save) and _FUSED_ATTN and ops.attn_block_supported(N, D, H)       # SRHIP_FUSED_ATTN=0: separate qkv GEMM + attention
        qkvx = self._buf(tag + "qkvx", (B, 3 * D), bf16) if (fused_attn and N == 257) else None
        wb = self.flat_bf16
        P = self.p
        Kp = cfg.in_chans * cfg.patch_size ** 2
        if Kp <= 64:                                # CIFAR-style 2x2 / 4x4 patches: direct fp32 kernel
            ops.patch_embed_fwd(img, img_index, P("patch_embed.proj.weight"), P("patch_embed.proj.bias"), P("cls_token"),
                                P("pos_embed"), x, B, cfg.in_chans, cfg.img_size, cfg.patch_size, D)
        else:                                       # ViT-S/16 at 224: unfold (bf16) -> GEMM with the [D, 768] filter -> + bias / pos / cls
            Np = N - 1
            col = self._buf(tag + "col", (B * Np, Kp), bf16)
            tok = self._buf(tag + "tok", (B * Np, D), f32)
            ops.patch_im2col(img, img_index, col, B, cfg.in_chans, cfg.img_size, cfg.patch_size)
            ops.gemm_nt(ops.EPI_F32, col, P("patch_embed.proj.weight", wb), tok, B * Np, D, Kp)
            ops.patch_assemble(tok, P("patch_embed.proj.bias"), P("cls_token"), P("pos_embed"), x, B, Np, D)
        scale = 64 ** -0.5
        dst0, dst1 = (droppath.stride(0), droppath.stride(1)) if droppath is not None else (0, 0)      # (a column range of the step's table: strided rows)
        ln_ready = False           # the previous block's fused launch already wrote this block's norm1 output
        for i in range(cfg.depth):
            b = "blocks.%d." % i
            s1 = ops.RawRows(droppath, i * dst0) if droppath is not None else None            # droppath[i, 0], droppath[i, 1] without building views
            s2 = ops.RawRows(droppath, i * dst0 + dst1) if droppath is not None else None
            if save:
                ln, qkv, ao = ctx.ln1[i], ctx.qkv[i], ctx.ao[i]
                ops.layernorm_fwd(x, P(b + "norm1.weight"), P(b + "norm1.bias"), cfg.eps, ln, ctx.st1[i][0], ctx.st1[i][1], M, D)
            elif not ln_ready:
                ops.layernorm_fwd(x, P(b + "norm1.weight"), P(b + "norm1.bias"), cfg.eps, ln, None, None, M, D)
            ao_scaled = False
            if fused_attn:
                # rows without a backward: qkv Linear + attention as ONE launch (one workgroup per image; qkv never reaches HBM)
                # (the DropPath factor of the branch rides on its bf16 output when the fused proj + MLP launch consumes it)
                ao_scaled = fused_mlp and _FUSED_PROJ and s1 is not None
                ops.attn_block_fused(ln, P(b + "attn.qkv.weight", wb), P(b + "attn.qkv.bias"), ao, B, N, D, H, scale, qkv_extra=qkvx,
                                     out_scale=s1 if ao_scaled else None)
            else:
                ops.gemm_nt(ops.EPI_BF16, ln, P(b + "attn.qkv.weight", wb), qkv, M, 3 * D, D, bias=P(b + "attn.qkv.bias"))
                ops.attn_fwd(qkv, ao, ctx.lse[i] if save else None, B, N, H, scale)
            if save:
                xm = ctx.xmid[i]
                ops.gemm_nt(ops.EPI_RESID_F32, ao, P(b + "attn.proj.weight", wb), xm, M, D, D, bias=P(b + "attn.proj.bias"),
                            row_scale=s1, rows_per_sample=N, aux_in=x, ldaux=D)
                ln2 = ctx.ln2[i]
                ops.layernorm_fwd(xm, P(b + "norm2.weight"), P(b + "norm2.bias"), cfg.eps, ln2, ctx.st2[i][0], ctx.st2[i][1], M, D)
                ops.gemm_nt(ops.EPI_GELU_BF16, ln2, P(b + "mlp.fc1.weight", wb), ctx.h[i], M, Hd, D, bias=P(b + "mlp.fc1.bias"),
                            aux_out=ctx.pre[i], ldaux=Hd)
                xn = ctx.xs[i + 1]
                ops.gemm_nt(ops.EPI_RESID_F32, ctx.h[i], P(b + "mlp.fc2.weight", wb), xn, M, D, Hd, bias=P(b + "mlp.fc2.bias"),
                            row_scale=s2, rows_per_sample=N, aux_in=xm, ldaux=D)
                x = xn
            elif fused_mlp and _FUSED_PROJ:
                # rows without a backward: proj + residual + LN2 + fc1 + GELU + fc2 + residual as ONE launch -- which also writes the NEXT
                # block's norm1 output (the operand of its fused qkv + attention launch) when there is a next block
                nb = "blocks.%d." % (i + 1)
                ln_ready = _FUSED_NEXT_LN and i + 1 < cfg.depth
                ops.mlp_fused_proj(x, ao, P(b + "attn.proj.weight", wb), P(b + "attn.proj.bias"), s1, P(b + "norm2.weight"),
                                   P(b + "norm2.bias"), cfg.eps, P(b + "mlp.fc1.weight", wb), P(b + "mlp.fc1.bias"),
                                   P(b + "mlp.fc2.weight", wb), P(b + "mlp.fc2.bias"), s2, N, M, D, Hd,
                                   ln_next=ln if ln_ready else None, next_gamma=P(nb + "norm1.weight") if ln_ready else None,
                                   next_beta=P(nb + "norm1.bias") if ln_ready else None, ao_scaled=ao_scaled)
            else:
                ops.gemm_nt(ops.EPI_RESID_F32, ao, P(b + "attn.proj.weight", wb), x, M, D, D, bias=P(b + "attn.proj.bias"),
                            row_scale=s1, rows_per_sample=N)
                if fused_mlp:
                    ops.mlp_fused(x, P(b + "norm2.weight"), P(b + "norm2.bias"), cfg.eps, P(b + "mlp.fc1.weight", wb),
                                  P(b + "mlp.fc1.bias"), P(b + "mlp.fc2.weight", wb), P(b + "mlp.fc2.bias"), s2, N, M, D, Hd)
                else:
                    ops.layernorm_fwd(x, P(b + "norm2.weight"), P(b + "norm2.bias"), cfg.eps, ln, None, None, M, D)
                    ops.gemm_nt(ops.EPI_GELU_BF16, ln, P(b + "mlp.fc1.weight", wb), hbuf, M, Hd, D, bias=P(b + "mlp.fc1.bias"))
                    ops.gemm_nt(ops.EPI_RESID_F32, hbuf, P(b + "mlp.fc2.weight", wb), x, M, D, Hd, bias=P(b + "mlp.fc2.bias"),
                                row_scale=s2, rows_per_sample=N)
        if out is not None:
            # out = (logits_all, feats_all, rows): the head writes image b's outputs to row rows[b] of the caller's buffers (the step's
            # [(pass, image), .] tables) -- no index_copy_ launches behind this forward; the dense feat is only kept for a backward
            logits_all, feats_all, rows = out
            feat = torch.empty(B, D, dtype=f32, device=self.device) if save else None
            logits = torch.empty(B, C, dtype=f32, device=self.device) if save else None
            if save:
                ctx.feat = feat
            ops.cls_head_fwd_scatter(x, P("norm.weight"), P("norm.bias"), cfg.eps, P("head.weight"), P("head.bias"), feat, logits,
                                     ctx.xhat if save else None, ctx.rstd if save else None, feats_all, logits_all, rows, B, N, D, C)
            return logits, feat, ctx
        feat = torch.empty(B, D, dtype=f32, device=self.device)
        logits = torch.empty(B, C, dtype=f32, device=self.device)
        if save:
            ctx.feat = feat
        ops.cls_head_fwd(x, P("norm.weight"), P("norm.bias"), cfg.eps, P("head.weight"), P("head.bias"), feat, logits,
                         ctx.xhat if save else None, ctx.rstd if save else None, B, N, D, C)
        return logits, feat, ctx

    def forward(self, x, only_fc=False, only_feat=False, **kw):
        """Reference-compatible entry (vit.py:285-306): returns {'logits','feat'}.  Inference-style call
        (DropPath active only in train mode, no activations kept)."""
        assert not only_fc, "only_fc is not on the SemiReward hot path"
        dp = self.make_droppath(x.shape[0]) if (self.training and self.cfg.drop_path_rate > 0) else None
        logits, feat, _ = self.forward_features(x.contiguous(), None, dp, save=False)
        return feat if only_feat else {"logits": logits, "feat": feat}

    __call__ = forward

    # ---- backward ---------------------------------------------------------------------------------
    def _bwd_plan(self, M, ctx):
        """Per-layer output-gradient buffers (the A operands of dW = dY^T X, kept until the ONE grouped launch after the layer
        loop) + the descriptor table of that launch (built once per batch size; ``ctx`` buffers are persistent too).
        Operands stay row-major [tokens, features]: srhip_gemm_tn_grouped_f32 gathers the MFMA fragments with LDS transpose
        reads, and sums the bias gradients on the way (no transposes, no column-sum kernels)."""
        key = ("bwdplan", M, id(ctx))
        if key in self._ws:
            return self._ws[key]
        cfg = self.cfg
        D, Hd = cfg.embed_dim, cfg.hidden
        mk = lambda c: torch.empty(M, c, dtype=torch.bfloat16, device=self.device)   # noqa: E731
        layers, problems = [], []
        G = lambda n: self.view(n, self.grad)   # noqa: E731
        for i in range(cfg.depth):
            b = "blocks.%d." % i
            t = dict(g2=mk(D), dpre=mk(Hd), g1=mk(D), dqkv=mk(3 * D))
            layers.append(t)
            problems += [(t["g2"], ctx.h[i], G(b + "mlp.fc2.weight"), G(b + "mlp.fc2.bias"), D, Hd, M),
                         (t["dpre"], ctx.ln2[i], G(b + "mlp.fc1.weight"), G(b + "mlp.fc1.bias"), Hd, D, M),
                         (t["g1"], ctx.ao[i], G(b + "attn.proj.weight"), G(b + "attn.proj.bias"), D, D, M),
                         (t["dqkv"], ctx.ln1[i], G(b + "attn.qkv.weight"), G(b + "attn.qkv.bias"), 3 * D, D, M)]
        out = dict(layers=layers, desc=ops.make_group_tn_desc(problems, self.device))
        # LayerNorm affine gradients: LN_REP partial copies per LayerNorm (same-address atomics of ~500 workgroups serialise), folded into
        # the gradient block by ONE launch after the layer loop.  Order: norm1, norm2 of block 0, 1, ...
        out["ln_part"] = torch.zeros(2 * cfg.depth, LN_REP, 2, D, dtype=torch.float32, device=self.device)
        out["ln_desc"] = ops.make_ln_reduce_desc([(G("blocks.%d.norm%d.weight" % (i, j)), G("blocks.%d.norm%d.bias" % (i, j)))
                                                  for i in range(cfg.depth) for j in (1, 2)], self.device)
        # Data parallel: the same launches cut into DW_GROUPS layer groups (last layers first), so that the all-reduce of a group's slice of the
        # flat gradient block can travel under the backward of the earlier layers (grad_ready_cb, see distributed.py).
        ng = max(1, min(DW_GROUPS, cfg.depth))
        per = -(-cfg.depth // ng)
        groups = []
        for hi_l in range(cfg.depth, 0, -per):
            lo_l = max(0, hi_l - per)
            names = [n for n, _ in self.names_shapes if n.startswith("blocks.") and lo_l <= int(n.split(".")[1]) < hi_l]
            lo = min(self.offsets[n][0] for n in names)
            hi = max(self.offsets[n][0] + int(torch.Size(self.offsets[n][1]).numel()) for n in names)
            assert sum(int(torch.Size(self.offsets[n][1]).numel()) for n in names) == hi - lo, "block parameters are contiguous in the flat block"
            groups.append(dict(lo_layer=lo_l, hi_layer=hi_l, flat=(lo, hi), desc=ops.make_group_tn_desc(problems[4 * lo_l:4 * hi_l], self.device),
                               ln_desc=ops.make_ln_reduce_desc([(G("blocks.%d.norm%d.weight" % (i, j)), G("blocks.%d.norm%d.bias" % (i, j)))
                                                                for i in range(lo_l, hi_l) for j in (1, 2)], self.device)))
        out["groups"] = groups
        self._ws[key] = out
        return out

    def backward(self, ctx, dlogits):
        """Accumulates d(loss)/d(params) into ``self.grad`` given dlogits fp32 [B, C] for a save=True forward."""
        self.backward_rows(ctx, dlogits, 0, ctx.B)
        self.backward_finish(ctx, dlogits)

    def _bwd_views(self, ctx, T, b0, b1):
        """Per-layer operands of the dX chain restricted to the images [b0, b1) (built once per range: a view costs ~3 us of host time)."""
        key = ("bwdviews", id(ctx), b0, b1)
        v = self._ws.get(key)
        if v is not None:
            return v
        cfg = self.cfg
        D, N, H = cfg.embed_dim, cfg.num_tokens, cfg.num_heads
        B = ctx.B
        M = B * N
        f32, bf16 = torch.float32, torch.bfloat16
        whole = b0 == 0 and b1 == B
        r = (lambda t: t) if whole else (lambda t: t[b0 * N:b1 * N])            # rows of a [M, .] buffer
        im = (lambda t: t) if whole else (lambda t: t[b0:b1])                   # images of a [B, ..] buffer
        v = types.SimpleNamespace()
        v.dx, v.dln, v.dao = r(self._buf("b_dx", (M, D), f32)), r(self._buf("b_dln", (M, D), bf16)), r(self._buf("b_dao", (M, D), bf16))
        v.delta = im(self._buf("b_delta", (B, H, N), f32))
        v.xhat, v.rstd = im(ctx.xhat), im(ctx.rstd)
        v.layers = []
        for i in range(cfg.depth):
            Ti = T["layers"][i]
            v.layers.append(types.SimpleNamespace(
                g2=r(Ti["g2"]), dpre=r(Ti["dpre"]), g1=r(Ti["g1"]), dqkv=r(Ti["dqkv"]), pre=r(ctx.pre[i]), xmid=r(ctx.xmid[i]), xs=r(ctx.xs[i]),
                st1=(r(ctx.st1[i][0]), r(ctx.st1[i][1])), st2=(r(ctx.st2[i][0]), r(ctx.st2[i][1])), qkv=r(ctx.qkv[i]), ao=r(ctx.ao[i]),
                lse=im(ctx.lse[i])))
        self._ws[key] = v
        return v

    def backward_rows(self, ctx, dlogits, b0, b1):
        """The input-gradient chain (head -> blocks 11 .. 0) of the images [b0, b1) of a save=True forward: every operand is a row range, rows of
        different images never meet before the weight-gradient products, so disjoint ranges may run on different streams at different times
        (measured in round 5 and not used by the step: profiles/r05_early_sup_backward_ab.txt; backward() runs ONE whole-batch chain).  ``dlogits``
        is the whole [B, C] buffer; only its rows [b0, b1) are read.  LayerNorm / final-norm affine gradients are added with atomics into the
        partial copies; everything that sums over ALL rows -- weight, bias, head and patch-embedding gradients -- is backward_finish."""
        cfg = self.cfg
        D, N, H, Hd, C = cfg.embed_dim, cfg.num_tokens, cfg.num_heads, cfg.hidden, cfg.num_classes
        nb = b1 - b0
        M = nb * N
        P = self.p
        G = lambda n: self.p(n, self.grad)   # noqa: E731
        self.ensure_transposed()
        T = self._bwd_plan(ctx.B * N, ctx)
        v = self._bwd_views(ctx, T, b0, b1)
        dl = dlogits if (b0 == 0 and b1 == ctx.B) else dlogits[b0:b1]
        v.dx.zero_()
        ops.cls_head_bwd(dl, P("head.weight"), P("norm.weight"), None, v.xhat, v.rstd, v.dx, None, None, G("norm.weight"), G("norm.bias"),
                         nb, N, D, C)
        scale = 64 ** -0.5
        dp = ctx.dp
        lnp = T["ln_part"]
        cb = self.grad_ready_cb if (b0 == 0 and b1 == ctx.B) else None          # data parallel overlap: whole-batch chains only
        self._groups_launched = cb is not None         # backward_finish: the layer groups' weight / LayerNorm launches already ran inside this chain
        gdone = {g["lo_layer"]: g for g in T["groups"]} if cb is not None else {}
        # dp[i, j, b0:] as a raw pointer (the DropPath factors of this range's images)
        dpr = (lambda i_, j_: ops.RawRows(dp, i_ * dp.stride(0) + j_ * dp.stride(1) + b0)) if dp is not None else (lambda i_, j_: None)
        L = v.layers
        ops.cast_scale_rows(v.dx, dpr(cfg.depth - 1, 1), N, L[cfg.depth - 1].g2, M, D)
        for i in reversed(range(cfg.depth)):
            b = "blocks.%d." % i
            Li = L[i]
            s1 = dpr(i, 0)
            # ---- MLP branch: x_out = x_mid + s2 * fc2(gelu(fc1(ln2(x_mid))));  g2 = bf16(s2 * dx) came from the previous LayerNorm backward
            ops.gemm_nt(ops.EPI_DGELU_BF16, Li.g2, self.wT[b + "mlp.fc2.weight"], Li.dpre, M, Hd, D, aux_in=Li.pre, ldaux=Hd)
            ops.gemm_nt(ops.EPI_BF16, Li.dpre, self.wT[b + "mlp.fc1.weight"], v.dln, M, D, Hd)
            ops.layernorm_bwd_part(v.dln, Li.xmid, Li.st2[0], Li.st2[1], P(b + "norm2.weight"), v.dx, lnp[2 * i + 1], LN_REP,
                                   Li.g1, s1, N, M, D)
            # ---- attention branch: x_mid = x_in + s1 * proj(attn(qkv(ln1(x_in))))
            ops.gemm_nt(ops.EPI_BF16, Li.g1, self.wT[b + "attn.proj.weight"], v.dao, M, D, D)
            ops.attn_bwd(Li.qkv, Li.ao, v.dao, Li.lse, Li.dqkv, v.delta, nb, N, H, scale)
            ops.gemm_nt(ops.EPI_BF16, Li.dqkv, self.wT[b + "attn.qkv.weight"], v.dln, M, D, 3 * D)
            ops.layernorm_bwd_part(v.dln, Li.xs, Li.st1[0], Li.st1[1], P(b + "norm1.weight"), v.dx, lnp[2 * i], LN_REP,
                                   L[i - 1].g2 if i > 0 else None, dpr(i - 1, 1) if i > 0 else None, N, M, D)
            g = gdone.get(i)
            if g is not None:               # layers [i, g.hi_layer) are finished: their weight / bias / LayerNorm gradients, then the hand-over
                desc, npb, ntiles, flops, nbytes = g["desc"]
                ops.gemm_tn_grouped_f32(desc, npb, ntiles, alpha=1.0, beta=1.0, flops=flops, nbytes=nbytes)
                ops.ln_grad_reduce(g["ln_desc"], lnp[2 * g["lo_layer"]:2 * g["hi_layer"]], 2 * (g["hi_layer"] - g["lo_layer"]), LN_REP, D)
                cb(*g["flat"])

    def backward_finish(self, ctx, dlogits):
        """Everything of the backward that sums over ALL rows, after the chain(s) of backward_rows have covered every image: head weight / bias
        gradients, the LayerNorm partial copies folded into the gradient block, all 4 * depth weight (and bias) gradients in ONE grouped
        launch (dW += dY^T X, db += colsum dY), the patch embedding."""
        cfg = self.cfg
        D, N, C = cfg.embed_dim, cfg.num_tokens, cfg.num_classes
        B = ctx.B
        M = B * N
        f32, bf16 = torch.float32, torch.bfloat16
        G = lambda n: self.p(n, self.grad)   # noqa: E731
        T = self._bwd_plan(M, ctx)
        dx = self._buf("b_dx", (M, D), f32)
        ops.cls_head_bwd(dlogits, None, None, ctx.feat, None, None, None, G("head.weight"), G("head.bias"), None, None, B, N, D, C)
        if not getattr(self, "_groups_launched", False):      # (a partial-range chain never launches them, whether or not a callback is installed)
            ops.ln_grad_reduce(T["ln_desc"], T["ln_part"], 2 * cfg.depth, LN_REP, D)
            desc, npb, ntiles, flops, nbytes = T["desc"]
            ops.gemm_tn_grouped_f32(desc, npb, ntiles, alpha=1.0, beta=1.0, flops=flops, nbytes=nbytes)
            if self.grad_ready_cb is not None:
                lo = min(g["flat"][0] for g in T["groups"]); hi = max(g["flat"][1] for g in T["groups"])
                self.grad_ready_cb(lo, hi)              # the blocks' range in one piece (the chain did not hand it over in groups)
        self._groups_launched = False
        Kp = cfg.in_chans * cfg.patch_size ** 2
        if Kp <= 64:
            ws = self._buf("b_pe_ws", (ops.patch_embed_bwd_ws_floats(B, cfg.in_chans, cfg.img_size, cfg.patch_size, D),), f32)
            ops.patch_embed_bwd_ws(dx, ctx.img, ctx.img_index, G("patch_embed.proj.weight"), G("patch_embed.proj.bias"), G("cls_token"),
                                   G("pos_embed"), ws, B, cfg.in_chans, cfg.img_size, cfg.patch_size, D)
        else:                                       # dWp += dx_tok^T col, dbp += colsum dx_tok (TN grouped GEMM, one problem); dpos, dcls
            Np = N - 1
            key = ("pebwd", B)
            if key not in self._ws:
                col = torch.empty(B * Np, Kp, dtype=bf16, device=self.device)
                dxt = torch.empty(B * Np, D, dtype=bf16, device=self.device)
                gw = self.view("patch_embed.proj.weight", self.grad).view(D, Kp)
                desc = ops.make_group_tn_desc([(dxt, col, gw, self.view("patch_embed.proj.bias", self.grad), D, Kp, B * Np)], self.device)
                self._ws[key] = (col, dxt, desc)
            col, dxt, desc = self._ws[key]
            ops.patch_im2col(ctx.img, ctx.img_index, col, B, cfg.in_chans, cfg.img_size, cfg.patch_size)
            ops.patch_grad_operands(dx, dxt, G("pos_embed"), G("cls_token"), B, Np, D)
            ops.gemm_tn_grouped_f32(desc[0], desc[1], desc[2], alpha=1.0, beta=1.0, flops=desc[3], nbytes=desc[4])


# ---- builders with the reference's names (vit.py:323-408); pretrained checkpoints need network -> ignored ------------
def _build(num_classes, kw, **cfg):
    kw = {k: v for k, v in kw.items() if k not in ("pretrained", "pretrained_path")}
    device = kw.pop("device", "cuda")
    m = VisionTransformer(VitConfig(num_classes=num_classes, **cfg), device=device)
    m.init_weights(kw.pop("seed", 0))
    return m


def vit_tiny_test(num_classes=10, **kw):
    return _build(num_classes, kw, img_size=8, patch_size=2, embed_dim=128, depth=2, num_heads=2, drop_path_rate=0.2)


def vit_small_patch16_224(num_classes=1000, **kw):
    """vit.py:358-371: ViT-S/16, 197 tokens."""
    return _build(num_classes, kw, img_size=224, patch_size=16, embed_dim=384, depth=12, num_heads=6, drop_path_rate=0.2)


def vit_base_patch16_96(num_classes=1000, **kw):
    """vit.py:374-390: ViT-B/16 on 96x96 images, 37 tokens (the stl10 / eurosat-style usb_cv configs)."""
    return _build(num_classes, kw, img_size=96, patch_size=16, embed_dim=768, depth=12, num_heads=12, drop_path_rate=0.2)


def vit_small_patch2_32(num_classes=1000, **kw):
    return _build(num_classes, kw, img_size=32, patch_size=2, embed_dim=384, depth=12, num_heads=6, drop_path_rate=0.2)


