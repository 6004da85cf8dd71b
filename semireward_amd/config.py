"""yaml -> argument namespace, the contract of the reference's ``get_config()`` (train.py:29-269) + ``over_write_args_from_file``
(semilearn/core/utils/misc.py:18-27): parser defaults, then the algorithm's ``get_argument()`` defaults (train.py:248-254), then every key
of the yaml on top -- so a config/SemiReward/**.yaml file runs unchanged:

    args = get_config("config/SemiReward/usb_cv/flexmatch/flexmatch_cifar100_200_0.yaml")
    alg = get_algorithm(args, get_net_builder(args.net, args.net_from_name))

The reference parses with ruamel.yaml (YAML 1.2: ``lr: 5e-05`` is a float); this image only has PyYAML (YAML 1.1: the same scalar is a
str), so the exponent-float form is added to the loader.  tests/test_cpu_sr_configs.py compares the result with the namespace the
reference itself produces for all 59 SR yamls (tests/golden/sr_configs.json).
"""
import argparse
import re

# Defaults of the reference's parser (train.py:36-225), as data.  Keys with dashes are stored under argparse's dest names.
PARSER_DEFAULTS = {
    "save_dir": "./saved_models", "save_name": "fixmatch", "resume": False, "load_path": None, "overwrite": True, "use_tensorboard": False,
    "use_wandb": False, "use_aim": False,
    "epoch": 1, "num_train_iter": 20, "num_warmup_iter": 0, "num_eval_iter": 10, "num_log_iter": 5, "num_labels": 400, "batch_size": 8,
    "uratio": 1, "eval_batch_size": 16, "ema_m": 0.999, "ulb_loss_ratio": 1.0,
    "optim": "SGD", "lr": 0.03, "momentum": 0.9, "weight_decay": 0.0005, "layer_decay": 1.0,
    "net": "wrn_28_2", "net_from_name": False, "use_pretrain": False, "pretrain_path": "",
    "use_cat": True, "amp": False, "clip_grad": 0, "imb_algorithm": None,
    "data_dir": "./data", "dataset": "cifar10", "num_classes": 10, "train_sampler": "RandomSampler", "num_workers": 1,
    "include_lb_to_ulb": True, "lb_imb_ratio": 1, "ulb_imb_ratio": 1, "ulb_num_labels": None, "img_size": 32, "crop_ratio": 0.875,
    "max_length": 512, "max_length_seconds": 4.0, "sample_rate": 16000,
    "world_size": 1, "rank": 0, "dist_url": "tcp://127.0.0.1:11111", "dist_backend": "nccl", "seed": 1, "gpu": None,
    "multiprocessing_distributed": False,
}


def _loader():
    import yaml

    class Loader12(yaml.SafeLoader):
        pass
    Loader12.add_implicit_resolver("tag:yaml.org,2002:float", re.compile(r"^[-+]?[0-9][0-9_]*[eE][-+]?[0-9]+$"), list("-+0123456789"))
    return yaml, Loader12


def load_yaml(path):
    yaml, Loader12 = _loader()
    with open(path, "r", encoding="utf-8") as f:
        return yaml.load(f.read(), Loader=Loader12)


def get_config(yml="", overrides=None):
    """Namespace for ``--c yml`` (+ ``overrides``, a dict applied last, the way command-line flags never win over the yaml in the reference:
    over_write_args_from_file runs after parse_args -- overrides are this package's addition for tests / benches)."""
    from .algorithms import ALGORITHMS
    a = argparse.Namespace(**PARSER_DEFAULTS)
    a.c = yml
    dic = load_yaml(yml) if yml else {}
    if "algorithm" not in dic and not (overrides and "algorithm" in overrides):
        raise KeyError("the yaml does not name an algorithm (train.py:248 looks it up in name2alg)")
    alg = (overrides or {}).get("algorithm", dic.get("algorithm"))
    for arg in ALGORITHMS[alg].get_argument():
        setattr(a, arg.name.lstrip("-"), arg.default)
    for k, v in dic.items():
        setattr(a, k, v)
    for k, v in (overrides or {}).items():
        setattr(a, k, v)
    return a
