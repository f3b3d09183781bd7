"""Device pick / seeds / directory scan (API of the reference's basicsr/utils/misc.py:15-112).

The reference parses torch.__version__ with a regex that raises IndexError on '2.10.0+rocm7.0'
(misc.py:12-13); here the version is parsed tolerantly.
"""
import os
import random
import re
import time
from os import path as osp

import numpy as np
import torch


def _version_tuple(v):
    m = re.match(r'^(\d+)\.(\d+)(?:\.(\d+))?', v)
    return tuple(int(g or 0) for g in m.groups()) if m else (0, 0, 0)


IS_HIGH_VERSION = _version_tuple(torch.__version__) >= (1, 12, 0)


def _mps_available():
    return IS_HIGH_VERSION and hasattr(torch.backends, 'mps') and torch.backends.mps.is_available()


def gpu_is_available():
    """True on Apple MPS or when torch sees a GPU with a working conv backend (MIOpen answers for cudnn on ROCm)."""
    return _mps_available() or bool(torch.cuda.is_available() and torch.backends.cudnn.is_available())


def get_device(gpu_id=None):
    """torch.device for `gpu_id` (int or None): mps > cuda (== HIP on ROCm) > cpu, as misc.py:27-47 orders them."""
    if gpu_id is not None and not isinstance(gpu_id, int):
        raise TypeError('Input should be int value.')
    index = '' if gpu_id is None else f':{gpu_id}'
    if _mps_available():
        return torch.device(f'mps{index}')
    return torch.device(f'cuda{index}' if gpu_is_available() else 'cpu')


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_time_str():
    return time.strftime('%Y%m%d_%H%M%S', time.localtime())


def scandir(dir_path, suffix=None, recursive=False, full_path=False):
    """Yield files under dir_path (relative paths unless full_path), optionally filtered by suffix."""
    if suffix is not None and not isinstance(suffix, (str, tuple)):
        raise TypeError('"suffix" must be a string or tuple of strings')
    root = dir_path

    def walk(d):
        for entry in sorted(os.scandir(d), key=lambda e: e.name):
            if entry.name.startswith('.'):
                continue
            if entry.is_file():
                p = entry.path if full_path else osp.relpath(entry.path, root)
                if suffix is None or p.endswith(suffix):
                    yield p
            elif recursive and entry.is_dir():
                yield from walk(entry.path)

    return walk(dir_path)


def sizeof_fmt(size, suffix='B'):
    """Human-readable byte count with binary prefixes ('1.5 KB')."""
    units = ('', 'K', 'M', 'G', 'T', 'P', 'E', 'Z', 'Y')
    i = 0
    while abs(size) >= 1024.0 and i < len(units) - 1:
        size /= 1024.0
        i += 1
    return f'{size:3.1f} {units[i]}{suffix}'
