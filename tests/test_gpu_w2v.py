"""Wav2Vec2 engine (semireward_amd/nets/wave2vec.py) on the HIP kernels against vectors produced by the reference ClassificationWave2Vec on a
random-init HF Wav2Vec2Model (tests/golden/w2v.npz): eval forward, train forward with injected dropout / SpecAugment / LayerDrop, gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import w2v2_ref as WR                  # noqa: E402
from semireward_amd import ops                     # noqa: E402
from semireward_amd.nets import wave2vec           # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def test_spec_augment_mask_matches_oracle():
    for seed, B, T in [(5, 3, 19), (6, 4, 199), (7, 1, 49)]:
        a = wave2vec.spec_augment_mask(np.random.Generator(np.random.PCG64(seed)), B, T, 0.05, 10, 2)
        assert np.array_equal(a, WR.spec_augment_mask(seed, B, T, 0.05, 10, 2)) and a.sum(1).min() >= 10


@pytest.mark.parametrize("tag", ["tiny", "tiny_skip", "base", "hubert_tiny"])
def test_w2v_matches_reference_golden(golden, tag):
    g = golden("w2v")
    C, B, S, seed, dseed = [int(v) for v in g[f"{tag}/meta"]]
    cfgd = WR.W2V_BASE if tag == "base" else WR.W2V_TINY_TEST
    cfg = WR.W2vCfg(num_classes=C, **cfgd)
    from semireward_amd.nets import hubert
    cls = hubert.ClassificationHubert if tag.startswith("hubert") else wave2vec.ClassificationWave2Vec      # same engine, same parameter names
    model = cls(wave2vec.W2vConfig(num_classes=C, **cfgd), device=DEV)
    assert sorted(n for n, _ in model.names_shapes) == sorted(n for n, _ in WR.param_shapes(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in WR.synth_params(cfg, seed).items()})
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    wave = torch.from_numpy(rng.standard_normal((B, S)).astype(np.float32)).to(DEV)
    y = torch.from_numpy(rng.integers(0, C, size=(B,), dtype=np.int64)).to(DEV)
    w = torch.from_numpy(rng.random(B).astype(np.float32)).to(DEV)
    TOL, TOLF = 4e-2, 2.5e-2                         # bf16 conv / GEMM / attention operands vs the fp32 reference
    model.eval()
    o = model(wave)
    assert rel(o["logits"], g[f"{tag}/eval_logits"]) < TOL and rel(o["feat"], g[f"{tag}/eval_feat"]) < TOLF
    model.train()
    model.inject = dict(seed=dseed, spec_mask=g[f"{tag}/spec_mask"], skip=[bool(v) for v in g[f"{tag}/skip"]])
    lg, ft, ctx = model.forward_features(wave, None, save=True)
    assert rel(lg, g[f"{tag}/train_logits"]) < TOL and rel(ft, g[f"{tag}/train_feat"]) < TOLF
    lg_i, ft_i, _ = model.forward_features(wave, None, save=False)
    assert rel(lg_i, lg) < 2e-3 and rel(ft_i, ft) < 2e-3
    loss, dl = torch.empty(1, device=DEV), torch.empty(B, C, device=DEV)
    ops.masked_ce(lg, y, w, None, 1.0, loss, dl, B, C)
    assert float(loss) == pytest.approx(float(g[f"{tag}/loss"]), rel=3e-2)
    model.zero_grad()
    model.backward(ctx, dl)
    worst = {}
    for n, gr in model.named_grads():
        gs = g.samp(f"{tag}/grad/{n}")
        a = gr.reshape(-1).cpu().numpy()[::gs["stride"]]
        if np.abs(gs["sample"]).max() == 0.0:
            assert np.abs(a).max() == 0.0, n                                               # parameters of a LayerDrop-skipped layer
            continue
        if n.endswith("k_proj.bias"):                                                      # analytically zero (softmax shift invariance)
            qs = np.abs(g.samp(f"{tag}/grad/{n.replace('k_proj', 'q_proj')}")["sample"]).max()
            assert np.abs(a).max() < 3e-2 * qs, n
            continue
        worst[n] = rel(a, gs["sample"])
    bad = {k: v for k, v in worst.items() if v > 8e-2}
    assert not bad, bad
    # gathered clips (clip_index) = the same rows
    idx = torch.tensor([B - 1, 0], dtype=torch.int32, device=DEV)
    model.eval()
    lg2, _, _ = model.forward_features(wave, idx, save=False)
    assert rel(lg2, torch.from_numpy(g[f"{tag}/eval_logits"])[[B - 1, 0]]) < TOL
