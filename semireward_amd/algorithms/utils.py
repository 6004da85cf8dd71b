"""SSL_Argument / str2bool (semilearn/algorithms/utils/misc.py) -- consumed by the reference's train.py:248-254."""


class SSL_Argument:
    def __init__(self, name, type, default, help=""):
        self.name, self.type, self.default, self.help = name, type, default, help


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise ValueError("Boolean value expected.")
