"""Backbone builders under the reference's names (semilearn/nets/__init__.py): the yaml key ``net:`` resolves here exactly as
``get_net_builder(net_name, from_name=False)`` does in the reference (semilearn/core/utils/build.py:14-39: ``getattr(semilearn.nets, name)``)."""
import importlib

_BUILDERS = {
    "vit_small_patch2_32": "vit", "vit_small_patch16_224": "vit", "vit_base_patch16_96": "vit",
    "wrn_28_2": "wrn", "bert_base_uncased": "bert", "bert_base_cased": "bert", "hubert_base": "hubert", "wave2vecv2_base": "wave2vec",
}
# reference nets the SemiReward hot path does not cover (SURVEY.md 8 scope): named so that the error says why
_NOT_BUILT = {
    "vit_tiny_patch2_32": "embed_dim 192 is outside the widths the row kernels are built for (128 / 384 / 768); the two tissuemnist SR yamls that "
                          "name it pair it with feature_dim 384 and fail in the reference's own Rewarder as well (SURVEY.md A.8)",
    "vit_base_patch16_224": "no config/SemiReward yaml uses it", "dinov2_vitl14": "no config/SemiReward yaml uses it",
    "dinov2_vitb14": "no config/SemiReward yaml uses it", "resnet50": "no config/SemiReward yaml uses it",
    "wrn_28_8": "no config/SemiReward yaml uses it", "wrn_var_37_2": "no config/SemiReward yaml uses it",
}


def get_net_builder(net_name, from_name=False):
    if from_name:
        raise NotImplementedError("net_from_name: True (torchvision.models) is outside the SemiReward hot path; every config/SemiReward yaml sets False")
    if net_name in _BUILDERS:
        return getattr(importlib.import_module("." + _BUILDERS[net_name], __name__), net_name)
    if net_name in _NOT_BUILT:
        raise NotImplementedError("net %r is not built on the HIP engine: %s" % (net_name, _NOT_BUILT[net_name]))
    raise AttributeError("unknown net %r" % (net_name,))


def __getattr__(name):                      # ``getattr(semireward_amd.nets, 'vit_small_patch2_32')`` as the reference does it
    if name in _BUILDERS or name in _NOT_BUILT:
        return get_net_builder(name)
    raise AttributeError(name)
