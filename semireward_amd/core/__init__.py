from .algorithmbase import AlgorithmBase, DeferredScalar  # noqa: F401
from .hooks import Hook, ParamUpdateHook, EMAHook, get_priority  # noqa: F401
from .registry import ALGORITHMS  # noqa: F401
