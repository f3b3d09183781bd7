import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLD


@pytest.fixture(scope='session')
def seed0_net():
    """CodeFormer restoration config with torch.manual_seed(0) weights (bit-identical to the reference's init)."""
    import torch
    import codeformer_amd.archs  # noqa: F401
    from codeformer_amd.utils.registry import ARCH_REGISTRY
    torch.manual_seed(0)
    return ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9,
                                           connect_list=['32', '64', '128', '256']).eval()
