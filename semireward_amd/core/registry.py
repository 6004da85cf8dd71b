"""Algorithm registry with the reference's keys (semilearn/core/utils/registry.py:11-46): the reference's yaml
`algorithm: srflexmatch` resolves to this package's class when `semireward_amd.algorithms` is imported."""


class Register:
    def __init__(self, name):
        self._dict, self._name = {}, name

    def register(self, key):
        def deco(cls):
            self._dict[key] = cls
            return cls
        return deco

    def __getitem__(self, key):
        return self._dict[key]

    def __contains__(self, key):
        return key in self._dict

    def keys(self):
        return self._dict.keys()


ALGORITHMS = Register("algorithms")
