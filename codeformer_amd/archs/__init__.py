"""Arch package: importing it registers every `*_arch.py` class in ARCH_REGISTRY (reference:
basicsr/archs/__init__.py:13-25)."""
import importlib
from copy import deepcopy
from os import path as osp

from ..utils.logger import get_root_logger
from ..utils.misc import scandir
from ..utils.registry import ARCH_REGISTRY

__all__ = ['build_network', 'ARCH_REGISTRY']

_here = osp.dirname(osp.abspath(__file__))
_arch_modules = [importlib.import_module(f'{__name__}.{osp.splitext(osp.basename(f))[0]}')
                 for f in sorted(scandir(_here)) if f.endswith('_arch.py')]


def build_network(opt):
    opt = deepcopy(opt)
    net = ARCH_REGISTRY.get(opt.pop('type'))(**opt)
    get_root_logger().info(f'Network [{net.__class__.__name__}] is created.')
    return net
