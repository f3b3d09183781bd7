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
    assert native.cf_version() == lib.ABI_VERSION == 22
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
    # ABI v17: weight forms 2 (stride 2) / 3 (1x1 on images) and the descriptors they belong to
    assert native.cf_pack_conv_weight_f16x2(1, 64, 64, 4, 64, 64, 2.0, 1, None) == -1 and 'form' in lib.last_error()
    assert native.cf_pack_conv_weight_f16x2(1, 64, 20, 2, 64, 20, 2.0, 1, None) == -1 and 'padding' in lib.last_error()       # 4 * 20 % 32
    assert native.cf_pack_conv_weight_f16x2(1, 64, 24, 2, 64, 24, 2.0, 1, None) == -1 and 'stride-2 form' in lib.last_error()  # cin % 16
    assert native.cf_pack_conv_weight_f16x2(1, 64, 24, 2, 64, 32, 2.0, 1, None) == -1 and 'stride-2 form' in lib.last_error()  # padded cin
    d = lib.ConvDesc(in0=1, weight=1, out=1, taps=9, stride=2, batch=1, hin=64, win=64, hout=32, wout=32, c0=64, cout=64,
                     cout_pad=64, bf16_mfma=3, acc_scale=1.0, pad_lo=1)
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'stride 2' in lib.last_error()         # symmetric padding: the fp32 kernel's
    d.pad_lo, d.hin, d.hout = 0, 40, 20
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'multiple of the 8x16 tile' in lib.last_error()
    d = lib.ConvDesc(in0=1, weight=1, out=1, taps=1, stride=1, batch=1, hin=64, win=64, hout=64, wout=64, c0=48, cout=64,
                     cout_pad=64, bf16_mfma=3, acc_scale=1.0)
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'multiples of 32' in lib.last_error()  # 1x1 on an image: the convolution kernel's rules
    d.c0, d.stats_out, d.stats_cpg = 64, 1, 2
    assert native.cf_conv2d(ctypes.byref(d), None) == -1
    # cf_act_scale_fused: statistics (one or two tensors) or a tensor, never both; cells and the table are required
    assert native.cf_act_scale_fused(None, 0, None, 0, None, 0, 1, 4.0, 1, 1, None) == -1 and 'not both' in lib.last_error()
    assert native.cf_act_scale_fused(1, 8, None, 0, 1, 1024, 1, 4.0, 1, 1, None) == -1 and 'not both' in lib.last_error()
    assert native.cf_act_scale_fused(None, 0, 1, 8, 1, 1024, 1, 4.0, 1, 1, None) == -1
    assert native.cf_act_scale_fused(None, 0, None, 0, 1, 1022, 1, 4.0, 1, 1, None) == -1 and 'multiple of 4' in lib.last_error()
    assert native.cf_act_scale_fused(1, 8, None, 0, None, 0, 1, 4.0, None, 1, None) == -1
    # ABI v19: winograd == 2 (F(4x4,3x3), 16x16 output patches, 64-wide channel tiles, cin <= 256) and its weight form; v20: split-half
    # or fp32 operands (the latter without a range scale)
    d = lib.ConvDesc(in0=1, weight=1, out=1, taps=9, stride=1, batch=1, hin=32, win=32, hout=32, wout=32, c0=64, cout=64,
                     cout_pad=64, bf16_mfma=1, acc_scale=1.0, winograd=2)
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'split-half or fp32' in lib.last_error()
    d.bf16_mfma, d.act_scale = 0, 1
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'range scale' in lib.last_error()
    d.bf16_mfma, d.act_scale, d.acc_scale = 3, None, 0.0
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'acc_scale' in lib.last_error()
    d.acc_scale, d.hout, d.hin = 1.0, 24, 24
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and '16x16' in lib.last_error()
    d.hout, d.hin, d.split_k = 32, 32, 1
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'split_k' in lib.last_error()
    d.split_k, d.winograd = 0, 3
    assert native.cf_conv2d(ctypes.byref(d), None) == -1 and 'winograd must be' in lib.last_error()
    d.winograd, d.stats_cpg = 2, 2
    assert native.cf_conv2d_stats_parts(ctypes.byref(d)) == 4        # one partial per 16x16 patch, four patches per 32x32 image
    assert native.cf_pack_conv_weight_winograd43_f16x2(1, 64, 64, 64, 64, 3.0, 1, None) == -1 and 'power of two' in lib.last_error()
    assert native.cf_pack_conv_weight_winograd43_f16x2(1, 64, 64, 96, 64, 2.0, 1, None) == -1 and 'padding' in lib.last_error()
    assert native.cf_pack_conv_weight_winograd43(1, 64, 64, 96, 64, 1, None) == -1 and 'padding' in lib.last_error()


def test_kernel_attribute_table_holds_every_instantiation(native):
    """cf_device_init checks its table of (kernel, dynamic LDS bytes) entries before it touches the device: the count of kernel
    instantiations that register themselves must fit (a GPU-less host fails later, at the first hipFuncSetAttribute)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('the table check is covered by every GPU test')
    assert native.cf_device_init() != 0
    assert 'overflow' not in lib.last_error(), lib.last_error()


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
