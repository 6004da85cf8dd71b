"""HuBERT classification backbone (the ``net: hubert_base`` of every config/SemiReward/usb_audio yaml) on the Wav2Vec2 engine.

Mirrors ``semilearn/nets/hubert/hubert.py`` (ClassificationHubert, builder ``hubert_base`` = facebook/hubert-base-ls960).  HF's HubertModel
is the Wav2Vec2 architecture -- GroupNorm conv feature encoder, LayerNorm + Linear projection, SpecAugment, weight-normed grouped positional
conv, post-LN encoder with LayerDrop -- under the same parameter names, so the engine is nets/wave2vec.py unchanged (parity pinned by the
``hubert_tiny`` vectors of tests/golden/w2v.npz, produced by the reference ClassificationHubert on a random-init HubertModel).  The one
difference on this path is the optimizer grouping: ``group_matcher`` (hubert.py:52-54) puts the positional conv into the stem group.
"""
from .wave2vec import ClassificationWave2Vec, W2vConfig, M_


class ClassificationHubert(ClassificationWave2Vec):
    def group_matcher(self, coarse=False, prefix=""):
        return dict(stem=r"^{}model.feature_projection|^{}model.feature_extractor|^{}model.encoder.pos_conv_embed".format(prefix, prefix, prefix),
                    blocks=r"^{}model.encoder.layers.(\d+)".format(prefix))

    def layer_ids(self):
        ids, lmax = ClassificationWave2Vec.layer_ids(self)
        for n in ids:
            if n.startswith(M_ + "encoder.pos_conv_embed"):
                ids[n] = 0
        return ids, lmax


def _build(num_classes, kw, **cfg):
    kw = {k: v for k, v in kw.items() if k not in ("pretrained", "pretrained_path")}
    device = kw.pop("device", "cuda")
    m = ClassificationHubert(W2vConfig(num_classes=num_classes, **cfg), device=device)
    m.init_weights(kw.pop("seed", 0))
    return m


def hubert_base(num_classes=2, **kw):
    return _build(num_classes, kw)


def hubert_tiny_test(num_classes=4, **kw):
    return _build(num_classes, kw, hidden=128, layers=2, heads=2, inter=256, conv_dim=(128, 128, 128), conv_kernel=(10, 3, 2),
                  conv_stride=(5, 2, 2), pos_k=16, pos_groups=4)
