"""The C-ABI library builds, loads without a GPU, and exports every symbol include/codeformer_hip.h declares."""
import ctypes
import os
import re

import pytest

from codeformer_amd import build as cf_build
from codeformer_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'codeformer_hip.h')


@pytest.fixture(scope='module')
def native():
    cf_build.build()
    return lib.load()


def _declared():
    src = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r'\b(cf_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_are_exported(native):
    names = _declared()
    assert len(names) >= 18
    raw = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f'{n} declared in the header but not exported'
    assert set(names) == set(lib.SIGNATURES), set(names) ^ set(lib.SIGNATURES)


def test_version_and_error_string(native):
    assert native.cf_version() == lib.ABI_VERSION == 17
    assert isinstance(lib.last_error(), str)


def test_conv_desc_layout_matches_header():
    """ctypes mirror vs the C struct: same field order and count as the header text."""
    src = open(HEADER).read()
    body = src[src.index('typedef struct cf_conv_desc {'):src.index('} cf_conv_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split('{')[-1]
        for nm in names.split(','):
            fields.append(nm.strip().split()[-1].lstrip('*'))
    assert fields == [f[0] for f in lib.ConvDesc._fields_]


def test_argument_errors_are_reported_without_a_gpu(native):
    """Validation happens before any launch, so it is testable on CPU: bad descriptors return CF_ERR_ARG."""
    d = lib.ConvDesc()
    assert native.cf_conv2d(ctypes.byref(d), None) == -1
    assert 'null' in lib.last_error()
    d = lib.ConvDesc(in0=1, weight=1, out=1, taps=5)
    assert native.cf_conv2d(ctypes.byref(d), None) == -1
    assert 'taps' in lib.last_error()
    assert native.cf_attention(1, 512, 1, 512, 1, 512, 1, 512, 1, 8, 64, 128, 1.0, None) == -1
    assert '256 keys' in lib.last_error()
    assert native.cf_packed_weight_elems(64, 9, 128) == 9 * 64 * 128
    # split-half operands (CF_OPERAND_F16X2): shape rules and the pack-time scale are checked before any launch
    d = lib.ConvDesc(in0=1, weight=1, out=1, taps=9, stride=1, batch=1, hin=16, win=16, hout=16, wout=16, c0=48, cout=64,
                     cout_pad=64, bf16_mfma=3, acc_scale=1.0)
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'multiples of 32' in lib.last_error()
    d.c0, d.acc_scale = 64, 0.0
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'acc_scale' in lib.last_error()
    assert native.cf_pack_conv_weight_f16x2(1, 64, 64, 0, 64, 64, 3.0, 1, None) == -1 and 'power of two' in lib.last_error()


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(lib.NativeLibraryError):
        lib.load()


def test_gpu_ops_refuse_cpu_tensors(native):
    import torch
    from codeformer_amd import ops
    with pytest.raises(ValueError):
        ops.to_nhwc(torch.zeros(1, 4, 2, 2))
