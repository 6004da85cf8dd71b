"""srhip_attn_block_fused (qkv Linear + attention of vit.py:93-104 in one launch, inference rows) on the norm1 output against
  * the launches it replaces (srhip_gemm_nt + srhip_attn_fwd) on the same inputs, and
  * an fp32 torch restatement of the reference ops (LayerNorm -> Linear -> softmax(q k^T / 8) v),
for both sequence lengths the kernel is built for (257 = ViT-S/2 at 32x32 incl. the lone 17th-tile token, 197 = ViT-S/16 at 224x224), batch
sizes 1 / 3 / 40, and as part of the whole backbone against the reference's golden logits (SRHIP_FUSED_ATTN on is the default)."""
import numpy as np
import pytest
import torch

from semireward_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("N", [257, 197])
@pytest.mark.parametrize("B", [1, 3, 40])
def test_fused_attention_half_matches_unfused_and_fp32(N, B):
    D, H = 384, 6
    assert ops.attn_block_supported(N, D, H) and not ops.attn_block_supported(N, 768, 12) and not ops.attn_block_supported(37, D, H)
    g = torch.Generator().manual_seed(100 * N + B)
    M = B * N
    x = (torch.randn(M, D, generator=g) * 1.5 + 0.3).to(DEV)
    x[5 % M] *= 20.0                                              # a row with a large norm
    gam, bet = (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.randn(D, generator=g) * 0.2).to(DEV)
    W = (torch.randn(3 * D, D, generator=g) * 0.08).to(DEV)
    Wb = W.to(torch.bfloat16)
    bq = (torch.randn(3 * D, generator=g) * 0.3).to(DEV)
    out = torch.full((M, D), 7.0, dtype=torch.bfloat16, device=DEV)
    ln = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    ops.layernorm_fwd(x, gam, bet, 1e-6, ln, None, None, M, D)
    qx = torch.empty(B, 3 * D, dtype=torch.bfloat16, device=DEV) if N == 257 else None
    ops.attn_block_fused(ln, Wb, bq, out, B, N, D, H, 0.125, qkv_extra=qx)
    # the path it replaces
    qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=DEV)
    ref = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(ops.EPI_BF16, ln, Wb, qkv, M, 3 * D, D, bias=bq)
    ops.attn_fwd(qkv, ref, None, B, N, H, 0.125)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    # same rounding points (xn, q/k/v, p in bf16): the two paths differ by accumulation order and the chunked softmax
    assert rel(out, ref) < 6e-3, rel(out, ref)
    assert float((out.float() - ref.float()).abs().max()) < 0.06 * float(ref.float().abs().max())
    # fp32 restatement of the reference ops
    xn = torch.nn.functional.layer_norm(x, (D,), gam, bet, 1e-6)
    q, k, v = (xn @ Wb.float().t() + bq).view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    want = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(M, D)
    assert rel(out, want) < 1.2e-2, rel(out, want)
    assert rel(ref, want) < 1.2e-2
    # every row of every image written, images independent: image 0 alone gives the same bytes
    out1 = torch.empty(N, D, dtype=torch.bfloat16, device=DEV)
    ops.attn_block_fused(ln[:N].contiguous(), Wb, bq, out1, 1, N, D, H, 0.125, qkv_extra=qx[:1] if qx is not None else None)
    assert torch.equal(out1, out[:N])
    # out_scale: the per-image DropPath factor of the branch (vit.py:163) applied before the bf16 rounding -- 1 is the identity, 0 gives
    # zeros, anything else the scaled fp32 result rounded once
    sc = torch.tensor([1.0, 0.0, 1.0 / 0.85] * B, device=DEV)[:B].contiguous()
    outs = torch.full((M, D), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.attn_block_fused(ln, Wb, bq, outs, B, N, D, H, 0.125, qkv_extra=qx, out_scale=sc)
    torch.cuda.synchronize()
    for bi in range(B):
        o_s, o_1 = outs[bi * N:(bi + 1) * N].float(), out[bi * N:(bi + 1) * N].float()
        if bi % 3 == 0:
            assert torch.equal(o_s, o_1)
        elif bi % 3 == 1:
            assert float(o_s.abs().max()) == 0.0
        else:                                                     # within one bf16 ulp of the scaled, once-rounded value
            assert float((o_s - o_1 / 0.85).abs().max()) <= 2.0 ** -7 * float(o_1.abs().max()) / 0.85


def test_unsupported_shapes_are_an_argument_error():
    x = torch.zeros(37 * 2, 384, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(37 * 2, 384, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(1152, 384, dtype=torch.bfloat16, device=DEV)
    b = torch.zeros(1152, device=DEV)
    with pytest.raises(RuntimeError):
        ops.attn_block_fused(x, w, b, o, 2, 37, 384, 6, 0.125)
    x257 = torch.zeros(257, 384, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):                              # N = 257 without the extra-token row
        ops._call("srhip_attn_block_fused", x257.data_ptr(), w.data_ptr(), b.data_ptr(), None, o.data_ptr(), None, 1, 257, 384, 6, 0.125, None)


@pytest.mark.parametrize("tag", ["small_p2_32"])
def test_backbone_inference_logits_with_and_without_the_fused_kernel(golden, tag, monkeypatch):
    """Whole ViT-S/2 inference forward with the fused attention half (default) against the reference's golden logits, and against the
    three-launch path (SRHIP_FUSED_ATTN=0 semantics) on the same weights."""
    from oracle import vit_ref as V
    from semireward_amd.nets import vit
    from semireward_amd.utils import synth
    g = golden("vit")
    cfg = V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32)
    B, seed = 24, 52
    m = vit.vit_small_patch2_32(num_classes=100, device=DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_params(V.param_shapes(cfg), seed).items()})
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    x = torch.from_numpy(rng.standard_normal((B, 3, cfg.img_size, cfg.img_size)).astype(np.float32)).to(DEV)
    m.eval()
    assert vit._FUSED_ATTN
    lg1, ft1, _ = m.forward_features(x, None, None, save=False)
    monkeypatch.setattr(vit, "_FUSED_ATTN", False)
    lg0, ft0, _ = m.forward_features(x, None, None, save=False)
    want = torch.from_numpy(g[f"{tag}/eval_logits"])              # KeyError if the fixture ever loses the key: the oracle assertion is not optional
    assert rel(lg1, lg0) < 6e-3 and rel(ft1, ft0) < 6e-3
    assert rel(lg1.cpu(), want) < 2e-2 and rel(lg0.cpu(), want) < 2e-2
    # the same 24 images through the production chain (fused proj + MLP + next-LN launches, as in every launch of >= _FUSED_MLP_MIN_ROWS rows)
    monkeypatch.setattr(vit, "_FUSED_ATTN", True)
    monkeypatch.setattr(vit, "_FUSED_MLP_MIN_ROWS", 1024)
    lg2, ft2, _ = m.forward_features(x, None, None, save=False)
    assert rel(lg2.cpu(), want) < 2e-2 and rel(lg2, lg1) < 6e-3 and rel(ft2, ft1) < 6e-3
