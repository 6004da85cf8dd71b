"""Importing this package registers the SemiReward algorithms under the reference's keys."""
from ..core.registry import ALGORITHMS  # noqa: F401
from .srflexmatch import SRFixMatch, SRFlexMatch  # noqa: F401
from .srfreematch import SRFreeMatch  # noqa: F401
from .srpseudolabel import SRPseudoLabel  # noqa: F401
from .srsoftmatch import SRSoftMatch  # noqa: F401


def get_algorithm(args, net_builder, tb_log=None, logger=None):
    """semilearn/algorithms/__init__.py:8-18."""
    return ALGORITHMS[args.algorithm](args=args, net_builder=net_builder, tb_log=tb_log, logger=logger)
