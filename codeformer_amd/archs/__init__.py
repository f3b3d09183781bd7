"""Arch package.  Importing it imports every `*_arch.py` next to this file, which fills ARCH_REGISTRY through the classes'
`@ARCH_REGISTRY.register()` decorators -- the auto-registration contract of basicsr/archs/__init__.py:13-25."""
import importlib
import pkgutil
from copy import deepcopy

from ..utils.logger import get_root_logger
from ..utils.registry import ARCH_REGISTRY

__all__ = ['build_network', 'ARCH_REGISTRY']

for _mod in sorted(m.name for m in pkgutil.iter_modules(__path__) if m.name.endswith('_arch')):
    importlib.import_module(f'{__name__}.{_mod}')


def build_network(opt):
    """opt: {'type': registered class name, **constructor kwargs} -> instance (basicsr/archs/__init__.py:19-25)."""
    kwargs = deepcopy(opt)
    cls = ARCH_REGISTRY.get(kwargs.pop('type'))
    net = cls(**kwargs)
    get_root_logger().info(f'Network [{cls.__name__}] is created.')
    return net
