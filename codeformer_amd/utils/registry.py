"""Name -> class tables: the plugin boundary of the reference (basicsr/utils/registry.py:4-82).

`ARCH_REGISTRY.get('CodeFormer')` is how the reference's entrypoints obtain the network class (inference_codeformer.py:135).
Behaviour kept from the reference: registering a name twice is an AssertionError, looking up an unknown name is a KeyError
with the same message, `register` works both as a decorator factory and as a plain call.
"""


class Registry:
    """A named dict of callables keyed by their `__name__`."""

    def __init__(self, name):
        self.name = name
        self.table = {}

    def _add(self, thing):
        key = thing.__name__
        assert key not in self.table, f"An object named '{key}' was already registered in '{self.name}' registry!"
        self.table[key] = thing
        return thing

    def register(self, obj=None):
        """`@REG.register()` above a class / function, or `REG.register(obj)` directly (returns None in that form)."""
        if obj is None:
            return self._add
        self._add(obj)

    def get(self, name):
        if name not in self.table:
            raise KeyError(f"No object named '{name}' found in '{self.name}' registry!")
        return self.table[name]

    def __contains__(self, name):
        return name in self.table

    def __iter__(self):
        return iter(self.table.items())

    def keys(self):
        return self.table.keys()


ARCH_REGISTRY = Registry('arch')
# the reference also instantiates these (training-side plugins; nothing registers into them here)
DATASET_REGISTRY, MODEL_REGISTRY, LOSS_REGISTRY, METRIC_REGISTRY = (Registry(n) for n in ('dataset', 'model', 'loss', 'metric'))
