"""Name -> class registries (API of the reference's basicsr/utils/registry.py:4-82).

`ARCH_REGISTRY.get('CodeFormer')` is the plugin boundary the reference's entrypoints use
(inference_codeformer.py:135); behaviour kept: duplicate registration asserts, unknown name raises KeyError.
"""


class Registry:

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, (f"An object named '{name}' was already registered "
                                           f"in '{self._name}' registry!")
        self._obj_map[name] = obj

    def register(self, obj=None):
        """Use as `@REG.register()` or `REG.register(obj)`; the key is `obj.__name__`."""
        if obj is not None:
            self._do_register(obj.__name__, obj)
            return None

        def decorator(target):
            self._do_register(target.__name__, target)
            return target

        return decorator

    def get(self, name):
        try:
            return self._obj_map[name]
        except KeyError:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!") from None

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self):
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


DATASET_REGISTRY = Registry('dataset')
ARCH_REGISTRY = Registry('arch')
MODEL_REGISTRY = Registry('model')
LOSS_REGISTRY = Registry('loss')
METRIC_REGISTRY = Registry('metric')
