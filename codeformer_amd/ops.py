"""Host-side operator layer: torch tensors in, HIP kernels through the C ABI, torch tensors out.

Activations are fp32 channels-last tensors of shape (B, H, W, C) ("NHWC"); token matrices are (rows, cols).
PyTorch only provides device memory (torch.empty) and the current stream; every FLOP runs in
libcodeformer_hip.so.  Nothing here has a CPU/eager fallback.
"""
import ctypes
import math
import os
import weakref

import torch

from . import lib as L
from .lib import (EPI_AXPY, EPI_AXPY2, EPI_GELU, EPI_LEAKY, EPI_NONE, EPI_RESIDUAL, EPI_SFT, PRO_AFFINE, PRO_AFFINE_SWISH,
                  PAD_EDGE, PAD_REFLECT, PAD_ZERO, PRO_LEAKY, PRO_NONE)

GN_GROUPS = 32
GN_EPS = 1e-6

# Optional per-launch timing for bench.py's roofline leg: when a list is installed here, conv2d brackets every
# kernel launch with events on the launch stream and appends (kind, algorithmic_flops, algorithmic_bytes, start, end).
PROFILE = None


def _f32(t):
    if t.dtype != torch.float32:
        raise TypeError(f'expected float32, got {t.dtype}')
    return t


def _act_dtype(t, what='activation'):
    """dtype of a conv activation: float32, or bfloat16 (bf16 STORAGE, cf_conv_desc.io_bf16 -- precision 'bf16' from 64x64 pixels up)."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f'{what}: expected float32 or bfloat16, got {t.dtype}')
    return t.dtype


def to_bf16(x):
    """fp32 NHWC activation -> its bf16 copy (round to nearest even, cf_f32_to_bf16): the tensors that ENTER the bf16-storage part of the
    generator (the decoder feature in front of the first bf16 Upsample, the encoder taps of the fusion blocks).  The GroupNorm partials /
    range-scale table the producer attached travel with the copy: they were taken from the fp32 values, as the bf16 epilogues take theirs."""
    if x.dtype == torch.bfloat16:
        return x
    _f32(x)
    if not x.is_contiguous() or x.numel() % 8:
        raise ValueError('to_bf16: expected a dense tensor with a multiple of 8 elements')
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    L.check(L.load().cf_f32_to_bf16(L.ptr(x), x.numel(), L.ptr(y, dtype=torch.bfloat16), L.stream_ptr()), 'cf_f32_to_bf16')
    for attr in ('_cf_stats', '_cf_act'):
        v = getattr(x, attr, None)
        if v is not None:
            setattr(y, attr, v)
    return y


class GNStats:
    """fp64 (sum, sumsq) partials of one NHWC tensor: [batch][channels/cpg][parts][2] (see cf_groupnorm_finalize).
    Produced by a conv epilogue (attached to the conv's output tensor as `._cf_stats`) or by the stand-alone pass."""

    __slots__ = ('part', 'parts', 'cpg')

    def __init__(self, part, parts, cpg):
        self.part, self.parts, self.cpg = part, parts, cpg


class PackedWeight:
    """A conv / linear weight in the kernel layout [tap][cin_pad/16][cout_pad][16] (+ bias).
    `bf16` is the operand code of cf_conv_desc.bf16_mfma: 0 / False fp32, 1 / True bf16, 2 IEEE half."""

    __slots__ = ('w', 'bias', 'cout', 'cin', 'taps', 'cout_pad', 'cin_pad', 'bf16', 'up2x', 'wino', 'scale', 's2', 'conv1')

    def __init__(self, w, bias, cout, cin, taps, cout_pad, cin_pad, bf16=False, up2x=False, wino=False, scale=1.0, s2=False, conv1=False):
        self.w, self.bias, self.cout, self.cin, self.taps = w, bias, cout, cin, taps
        self.cout_pad, self.cin_pad, self.bf16, self.up2x, self.wino = cout_pad, cin_pad, bf16, up2x, wino
        self.s2 = s2         # f16x2 packing in the stride-2 (space-to-depth) form: only conv2d(stride=2) takes it
        self.conv1 = conv1   # f16x2 1x1 weight in the convolution kernel's layout (images of more than TOKEN_IMAGE_MAX pixels), not the token GEMM's
        self.scale = scale   # f16x2 packing: the power of two the weights were multiplied by (cf_conv_desc.acc_scale = 1 / scale)


def _cout_pad(cout):
    if cout <= 32:
        return 32
    if cout <= 64:
        return 64
    return (cout + 127) // 128 * 128


WINOGRAD = 3   # value of the operand-code argument that selects the Winograd F(2x2,3x3) fp32 evaluation
SPLIT = 4      # ... the split-half evaluation: fp32 operands as hi + lo IEEE halves, 3 f16 MFMAs per product (cf_split.hip)
SPLIT_DIRECT = 5   # ... the same, but layers the split kernel does not take run on the direct fp32 kernel instead of Winograd
WSPLIT = 6     # ... Winograd F(2x2,3x3) with split-half operands in the 16 transform-domain GEMMs (cf_winograd.hip, H2)
GSPLIT = 7     # ... a Linear / 1x1 weight for the split-half token GEMM (cf_gemm_split.hip)
WF16 = 8       # ... Winograd F(2x2,3x3) with SINGLE IEEE-half operands (eight-wave kernel of cf_wsplit.hip; precision 'fp16')
WBF16 = 9      # ... the same with single bf16 operands (precision 'bf16')
SPLIT_F43 = 10  # ... REQUEST: SPLIT, with Winograd F(4x4,3x3) where its kernel applies (~5x the error of F(2,3): generator / CFT layers, and the encoder behind its margin gate)
WF43 = 11      # ... Winograd F(4x4,3x3) with split-half operands (cf_wf43.hip; cf_conv_desc.winograd = 2)
WINOGRAD_F43 = 12   # ... REQUEST: WINOGRAD (exact fp32), with Winograd F(4x4,3x3) on fp32 operands where its kernel applies (generator / CFT only)
WF43F = 13     # ... Winograd F(4x4,3x3) with IEEE-fp32 operands (cf_wf43.hip on v_mfma_f32_16x16x4_f32; winograd = 2, operand fp32)
OPERAND_F16X2 = 3   # enum cf_operand value behind SPLIT / WSPLIT / GSPLIT / WF43
SPLIT_CODES = (SPLIT, SPLIT_DIRECT, SPLIT_F43)   # requested codes that put un-normalised inputs / stride-2 / 1x1 layers on the split-half kernels
# SPLIT layers that the Winograd kernel covers take its split-half form (4/9 of the MFMA work); CODEFORMER_HIP_SPLIT_WINOGRAD=0
# keeps them on the direct split-half kernel.
SPLIT_WINOGRAD = os.environ.get('CODEFORMER_HIP_SPLIT_WINOGRAD', '1') != '0'


def split_ok(cin, cout, hin, win, c_split=None):
    """Shapes the split-half kernel covers (3x3 stride-1 dense NHWC, plain or folded upsample): 32-channel K slabs (also at a
    concat boundary), 64-wide channel tiles, whole 16x16 tiles of the conv's INPUT grid."""
    return cin % 32 == 0 and cout % 64 == 0 and hin % 16 == 0 and win % 16 == 0 and (c_split is None or c_split % 32 == 0)


def split_s2_ok(cin, cout, hin, win):
    """Shapes the split-half kernel covers at stride 2 (Downsample: one dense input, zero row / column bottom / right): 32-channel
    slabs of the space-to-depth view (2 * cin % 32 == 0), 64-wide channel tiles, whole 8x16 tiles of the OUTPUT grid."""
    return cin % 16 == 0 and cout % 64 == 0 and hin % 16 == 0 and win % 32 == 0


def split_1x1_ok(cin, cout, h, w, c_split=None):
    """1x1 convolutions the split-half kernel streams (ResBlock skips on images): 32-channel slabs, 64-wide channel tiles, 8x16 tiles."""
    return cin % 32 == 0 and cout % 64 == 0 and h % 8 == 0 and w % 16 == 0 and h * w > TOKEN_IMAGE_MAX and (c_split is None or c_split % 32 == 0)


def winograd_ok(cin, cout, hout, wout):
    """Shapes the Winograd kernel covers (3x3 stride-1 dense NHWC): whole 8x16 output patches, 64-wide channel tiles."""
    return cin % 16 == 0 and cout % 64 == 0 and hout % 8 == 0 and wout % 16 == 0


# Which layers a SPLIT_F43 / WINOGRAD_F43 request puts on the F(4x4,3x3) kernel: 'auto' (default) = the shapes where it pays
# (profiles/r04_f43_per_shape.txt, r04_f43_fp32_check_time.txt, r04_latency_minpix.txt): every covered layer with 64 output channels (the
# 8-wave form, two workgroups per CU: x1.0-1.17 at sixteen faces) and the layers with a multiple of 128 output channels (the 16-wave form)
# from F43_WIDE_MIN_PIXELS up.  Split-half operands: 128x128 -- at 64x64 the form wins x1.11 at sixteen faces (0.15 ms of a 36 ms step) but a
# 64x64 image has 16 patches, and at one face per call its 9 launches cost 0.5 ms of 7.1 (140 -> 150 faces/s with the limit at 128x128);
# fp32 operands: 64x64 (x1.75 there: the fp32 MFMA work itself is the bound).  At 32x32 it loses at any batch (x0.6: 4 patches per image).
# 'c64' = the 64-channel group only; 'all' = every covered shape; '0' = none (A/B).  The rule is a function of the per-image shape only:
# results do not depend on the batch.
F43_LAYERS = os.environ.get('CODEFORMER_HIP_F43', 'auto')
F43_WIDE_MIN_PIXELS = int(os.environ.get('CODEFORMER_HIP_F43_MINPIX', 128 * 128))        # smallest image of the 16-wave form under 'auto', split-half operands
F43_WIDE_MIN_PIXELS_FP32 = int(os.environ.get('CODEFORMER_HIP_F43_MINPIX_FP32', 64 * 64))  # ... fp32 operands


def f43_ok(cin, cout, hout, wout, fp32=False):
    """Shapes the F(4x4,3x3) kernel covers (3x3 stride-1 dense NHWC): whole 16x16 output patches, 64-wide channel tiles, at most 256
    input channels (512 in the 16-wave form on 32-channel slabs: the GroupNorm rows of an image sit in LDS) -- narrowed by F43_LAYERS to
    where it pays (fp32: the operand type)."""
    if F43_LAYERS == '0' or (F43_LAYERS == 'c64' and cout != 64):
        return False
    if F43_LAYERS == 'auto' and cout != 64 and (cout % 128 or hout * wout < (F43_WIDE_MIN_PIXELS_FP32 if fp32 else F43_WIDE_MIN_PIXELS)):
        return False
    cin_max = 512 if (cout % 128 == 0 and cin % 32 == 0) else 256     # GroupNorm rows in LDS: 512 in the 16-wave form on 32-channel slabs (round 6), else 256
    return cin % 16 == 0 and cin <= cin_max and cout % 64 == 0 and hout % 16 == 0 and wout % 16 == 0


# precision 'fp32': the Upsample blocks (nearest x2 + 3x3) on the fp32 F(4x4,3x3) kernel with an UPSAMPLING gather (round 6) instead of the folded
# sub-pixel form on the direct fp32 kernel, which executes every one of its 4 products per output (0.82 of the fp32 MFMA peak: nothing left
# to schedule) -- F(4,3) needs 2.25.  CODEFORMER_HIP_F43_UPSAMPLE=0: the folded form everywhere (A/B).
F43_UPSAMPLE = os.environ.get('CODEFORMER_HIP_F43_UPSAMPLE', '1') != '0'


def f43_up_ok(cin, cout, hout, wout):
    """Shapes the upsampling form of the fp32 F(4x4,3x3) kernel covers ((hout, wout) = the OUTPUT size): the 16-wave workgroup on
    32-channel slabs, GroupNorm-table limit of 256 input channels, whole 16x16 output patches, from the size where the 16-wave form pays."""
    return F43_UPSAMPLE and F43_LAYERS in ('auto', 'all') and cin % 32 == 0 and cin <= 256 and cout % 128 == 0 and hout % 16 == 0 and wout % 16 == 0 and \
        hout * wout >= F43_WIDE_MIN_PIXELS_FP32


# Smallest per-image input of the DIRECT split-half kernel and of the eight-wave Winograd kernel.  The 16x16 latents are below it: with
# SPLIT they run the four-wave Winograd kernel on split halves with split-K (conv_code -> WSPLIT: measured on the reference's crops
# before it became the default, profiles/r02_encoder_split_check.txt), with SPLIT_DIRECT they stay on the exact fp32 kernel.
SPLIT_MIN_PIXELS = 32 * 32
TOKEN_IMAGE_MAX = 1024   # CF_TOKEN_IMAGE_MAX of cf_common.h: f16x2 1x1 layers on larger images run the streaming convolution kernel


def wsingle_ok(cin, cout, h, w):
    """Shapes the eight-wave Winograd kernel covers (the rule of cf_wsplit_covers): 128-wide channel tiles from 32x32 pixels up."""
    return winograd_ok(cin, cout, h, w) and cout % 128 == 0 and h * w >= SPLIT_MIN_PIXELS


# Single 16-bit operand modes ('bf16' / 'fp16'): layers the eight-wave Winograd kernel covers run there (one MFMA per transform-domain
# product); CODEFORMER_HIP_WINOGRAD_16BIT=0 keeps every layer on the direct 16-bit instantiations of cf_igemm.hip.
WINOGRAD_16BIT = os.environ.get('CODEFORMER_HIP_WINOGRAD_16BIT', '1') != '0'


def c_split_ok(c_split, slab):
    return c_split is None or c_split % slab == 0


def conv_code(code, cin, cout, h, w, up2x=False, c_split=None, plain=True):
    """Operand code a 3x3 stride-1 convolution really runs with, given the requested one and its shape ((h, w) = INPUT size).
    SPLIT falls back to WINOGRAD and WINOGRAD to the direct fp32 kernel where their kernels do not apply; the decision depends
    on the per-image shape only (never on the batch), so results stay batch-invariant."""
    code = int(code)
    f43_slab = 32 if (cout % 128 == 0 and cin % 32 == 0) else 16   # slab of the form cf_conv2d runs: a concat boundary must not cut one
    if code == WINOGRAD_F43:
        if plain and not up2x and c_split_ok(c_split, f43_slab) and f43_ok(cin, cout, h, w, fp32=True):
            return WF43F
        code = WINOGRAD
    if code == SPLIT_F43:
        if plain and not up2x and c_split_ok(c_split, f43_slab) and f43_ok(cin, cout, h, w):
            return WF43
        code = SPLIT
    if code in (SPLIT, SPLIT_DIRECT):
        if code == SPLIT and SPLIT_WINOGRAD and plain and not up2x and winograd_ok(cin, cout, h, w):
            return WSPLIT
        if plain and split_ok(cin, cout, h, w, c_split) and (up2x or h * w >= SPLIT_MIN_PIXELS):   # (the folded upsample has no Winograd form: the direct split kernel beats the fp32 one from 16x16 up, tools/up16_probe.py)
            return SPLIT
        code = WINOGRAD if code == SPLIT else 0
    if code == WINOGRAD:
        return WINOGRAD if (plain and not up2x and winograd_ok(cin, cout, h, w)) else 0
    if code and not (plain and cin % 32 == 0 and cout % 4 == 0):
        return 0
    if code in (1, 2) and WINOGRAD_16BIT and plain and not up2x and wsingle_ok(cin, cout, h, w):
        return WBF16 if code == 1 else WF16
    return code


# ---- range of the 16-bit MFMA operands --------------------------------------------------------------------------------------------
# An IEEE-half operand overflows above 65504 (16376 in the Winograd domain: the input transform sums four activations) and its lo half goes
# subnormal below 2^-3.  Weights are scaled into [2^14, 2^15) when they are packed.  Activations:
#   * inputs that pass a GroupNorm prologue are bounded by |gamma| * sqrt(n - 1) + |beta| (n = elements of a group); the arch modules check
#     that bound against HALF_LIMIT when they choose a layer's kernel (gn_range_ok) and fall back to exact fp32 where it fails;
#   * UN-NORMALISED inputs (the residual stream into Upsample.conv, the quantised feature, the CFT branch) get a per-image power-of-two
#     scale (act_scale -> cf_conv_desc.act_scale): exact, any fp32 magnitude.  CODEFORMER_HIP_RANGE_SCALE=0 switches it off (A/B only).
RANGE_SCALE = os.environ.get('CODEFORMER_HIP_RANGE_SCALE', '1') != '0'
HALF_LIMIT = 65504.0 / 4.0    # largest |activation| a Winograd-domain IEEE-half operand can carry


def switches():
    """The module-level A/B switches a captured forward depends on (part of the graph-replay key of the arch modules)."""
    return (SPLIT_WINOGRAD, F43_LAYERS, F43_UPSAMPLE, F43_WIDE_MIN_PIXELS, F43_WIDE_MIN_PIXELS_FP32, WINOGRAD_16BIT, RANGE_SCALE, ACT_FUSED, SPLITK_MAX, GEMM_IN_WG_MAX_OUTPUTS, FINALIZE_FUSED)


def needs_act_scale(pw):
    """True when `pw` runs on a kernel with 16-bit MFMA operands that applies cf_conv_desc.act_scale."""
    if pw.conv1:
        return RANGE_SCALE
    return RANGE_SCALE and pw.taps == 9 and int(pw.bf16) in (1, 2, OPERAND_F16X2) and (bool(pw.wino) or int(pw.bf16) == OPERAND_F16X2)


_ACT_CELLS = {}
ACT_FUSED = os.environ.get('CODEFORMER_HIP_ACT_FUSED', '1') != '0'   # 0: two launches per table (A/B only)


def _act_cells(device, batch):
    """2 * batch zero-initialised words per (device, stream) for cf_act_scale_fused; every launch leaves them at zero."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _ACT_CELLS.get(key)
    if t is None or t.numel() < 2 * batch:
        t = torch.zeros(max(2 * batch, 256), dtype=torch.int32, device=device)
        _ACT_CELLS[key] = t
    return t


def tensor_version(t):
    """Version counter of a tensor, or None for inference tensors (torch.inference_mode), which do not track one -- reading
    `._version` there raises.  Callers skip their version-keyed caches for None."""
    return None if t.is_inference() else t._version


def act_scale(x, x2=None, growth=4.0):
    """(B, 2) float32 table (s_b, 1 / s_b): power-of-two scale that puts growth * max|x_b| into [2^13, 2^14) -- from the statistics
    partials the producing conv wrote (no pass over x; the bound is loose by a few bits, which is harmless) or, for tensors without
    them, from x itself; one launch (cf_act_scale_fused).  x2: the second half of a concatenated input -- the table then covers both.
    Cached on the tensor under (growth, tensor version): one table serves every conv that reads the tensor, and an in-place write
    (out= reuse, a user-held buffer) or another growth factor gets a fresh table; pairs are not cached."""
    ver = tensor_version(x)   # None under torch.inference_mode(): inference tensors keep no version counter
    if ver is None and getattr(x, '_cf_stats', None) is not None:
        # ... but a tensor that carries its producer's statistics is a conv2d output of THIS forward (a fresh torch.empty per launch, never
        # written again): keyed on the object itself, one table serves all its consumers -- without this every consumer relaunched the
        # kernel under inference_mode (and baked the extra launches into captured graphs).  User-held inference tensors (no statistics
        # attached) still compute their table on every call.
        ver = 'inference'
    if x2 is None and ver is not None:
        cached = getattr(x, '_cf_act', None)
        if cached is not None and cached[0] == (float(growth), ver):
            return cached[1]
    if x2 is not None:      # a table groupnorm_tables([x, x2], act_growth=...) wrote in its own launch
        cached = getattr(x, '_cf_act_pair', None)
        if cached is not None and cached[2]() is x2 and cached[0] == _act_key(x, growth) and cached[1] == _act_key(x2, growth):
            return cached[3]
    lib = L.load()
    B = x.shape[0]
    act = torch.empty(B, 2, dtype=torch.float32, device=x.device)
    cells = L.ptr(_act_cells(x.device, B), dtype=torch.int32)
    st = getattr(x, '_cf_stats', None)
    st2 = None if x2 is None else getattr(x2, '_cf_stats', None)
    if x2 is not None and (st is None or st2 is None or not ACT_FUSED):
        # a half without statistics partials: two single tables combined on the host side of the stream (rare: every conv that feeds a
        # concatenation writes partials)
        a1, a2 = act_scale(x, growth=growth), act_scale(x2, growth=growth)
        return torch.stack((torch.minimum(a1[:, 0], a2[:, 0]), torch.maximum(a1[:, 1], a2[:, 1])), dim=1)
    if not ACT_FUSED and x2 is None:   # A/B only: the two-launch entry points of ABI v16
        scratch = torch.empty(B * 32, dtype=torch.float32, device=x.device)
        if st is not None:
            L.check(lib.cf_act_scale_from_stats(L.ptr(st.part, dtype=torch.float64), B, st.part.numel() // (2 * B), float(growth), L.ptr(scratch), L.ptr(act), L.stream_ptr()), 'cf_act_scale_from_stats')
        else:
            L.check(lib.cf_act_scale_from_tensor(L.ptr(_f32(x)), B, x.numel() // B, float(growth), L.ptr(scratch), L.ptr(act), L.stream_ptr()), 'cf_act_scale_from_tensor')
        if ver is not None:
            x._cf_act = ((float(growth), ver), act)
        return act
    if st is not None:
        nper = st.part.numel() // (2 * B)
        p2, n2 = (L.ptr(st2.part, dtype=torch.float64), st2.part.numel() // (2 * B)) if st2 is not None else (None, 0)
        L.check(lib.cf_act_scale_fused(L.ptr(st.part, dtype=torch.float64), nper, p2, n2, None, 0, B, float(growth), cells, L.ptr(act), L.stream_ptr()),
                'cf_act_scale_fused')
    else:
        if not x.is_contiguous() or (x.numel() // B) % 4:
            raise ValueError('act_scale: expected a dense tensor with a multiple of 4 elements per image')
        L.check(lib.cf_act_scale_fused(None, 0, None, 0, L.ptr(_f32(x)), x.numel() // B, B, float(growth), cells, L.ptr(act), L.stream_ptr()),
                'cf_act_scale_fused')
    if x2 is None and ver is not None:
        x._cf_act = ((float(growth), ver), act)
    return act


class StatsPair:
    """Shared statistics buffer of TWO sibling convolutions of equal shape (the scale.0 / shift.0 convolutions of a fusion block,
    codeformer_arch.py:153-154): conv2d(..., stats_into=pair) places each launch's partials in one half, and act_scale_pair(pair, batch)
    turns both into their range-scale tables with ONE cf_act_scale_fused launch over 2 x batch "images"."""

    def __init__(self):
        self.buf, self.n, self.used = None, 0, 0

    def take(self, n, device):
        if self.buf is None:
            self.buf, self.n = torch.empty(2 * n, dtype=torch.float64, device=device), n
        if n != self.n or self.used >= 2:
            raise ValueError('StatsPair: two launches of equal statistics size')
        self.used += 1
        return self.buf[(self.used - 1) * n:self.used * n]


def act_scale_pair(pair, batch, growth=4.0):
    """((B, 2), (B, 2)) range-scale tables of the two tensors whose statistics share `pair` -- bitwise act_scale() of each, one launch."""
    if pair.used != 2:
        raise ValueError('act_scale_pair: the pair has not been filled by two launches')
    lib = L.load()
    act = torch.empty(2 * batch, 2, dtype=torch.float32, device=pair.buf.device)
    cells = L.ptr(_act_cells(pair.buf.device, 2 * batch), dtype=torch.int32)
    L.check(lib.cf_act_scale_fused(L.ptr(pair.buf, dtype=torch.float64), pair.n // (2 * batch), None, 0, None, 0, 2 * batch, float(growth), cells, L.ptr(act),
                                   L.stream_ptr()), 'cf_act_scale_fused')
    return act[:batch], act[batch:]


def gn_range_ok(gmax, bmax, n):
    """GroupNorm output bound |gamma| * sqrt(n - 1) + |beta| (n elements per group) against the IEEE-half operand range."""
    return gmax * math.sqrt(max(n - 1, 1)) + bmax < HALF_LIMIT


def exact_code(code):
    """Operand code of the exact-fp32 evaluation that replaces a 16-bit-operand code (range fallback); bf16 has fp32's exponent."""
    code = int(code)
    return {SPLIT: WINOGRAD, SPLIT_F43: WINOGRAD, SPLIT_DIRECT: 0, 2: WINOGRAD}.get(code, code)


def pack_weight(weight, bias=None, bf16=False, up2x=False, f16=False, stride2=False):
    """weight: (cout, cin, 3, 3) | (cout, cin, 1, 1) | (cout, cin) CUDA fp32 -> PackedWeight.
    bf16=True (3x3 only, cin % 32 == 0): bf16 operands for the v_mfma_f32_32x32x16_bf16 path of cf_conv2d.
    f16=True (3x3 only, cin % 32 == 0): IEEE-half operands (general instantiations; RRDBNet's half mode).
    up2x=True (3x3 only): taps folded for conv2d(upsample=True) -- nearest x2 + 3x3 as four 2x2 sub-pixel convolutions.
    stride2=True (code SPLIT only): the stride-2 form of the split-half kernel -- a 2x2 convolution of the space-to-depth input."""
    lib = L.load()
    code = 2 if f16 else int(bf16)   # callers may pass the operand code (0 fp32 / 1 bf16 / 2 f16 / 3 winograd) through `bf16`
    w = _f32(weight.detach()).contiguous()
    b = None if bias is None else _f32(bias.detach()).contiguous().clone()
    cout, cin = w.shape[0], w.shape[1]
    if code == WINOGRAD:
        if up2x or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or cin % 16 or cout % 64:
            raise ValueError('winograd packing needs a 3x3 weight with cin % 16 == 0 and cout % 64 == 0 (no up2x)')
        packed = torch.empty(16 * cin * cout, dtype=torch.float32, device=w.device)
        L.check(lib.cf_pack_conv_weight_winograd(L.ptr(w), cout, cin, cout, cin, L.ptr(packed, dtype=None), L.stream_ptr()),
                'cf_pack_conv_weight_winograd')
        return PackedWeight(packed, b, cout, cin, 9, cout, cin, wino=True)
    if code == GSPLIT:
        w2 = w.reshape(cout, cin)
        if w.dim() not in (2, 4) or w.numel() != cout * cin or cout % 64 or cin % 128:
            raise ValueError('f16x2 GEMM packing needs a Linear / 1x1 weight with cout % 64 == 0 and cin % 128 == 0')
        wmax = float(w2.abs().max())
        scale = 1.0 if wmax == 0.0 or not math.isfinite(wmax) else 2.0 ** (14 - math.frexp(wmax)[1] + 1)
        packed = torch.empty(cout * cin, dtype=torch.float32, device=w.device)
        L.check(lib.cf_pack_linear_weight_f16x2(L.ptr(w2.contiguous()), cout, cin, scale, L.ptr(packed, dtype=None), L.stream_ptr()),
                'cf_pack_linear_weight_f16x2')
        return PackedWeight(packed, b, cout, cin, 1, cout, cin, bf16=OPERAND_F16X2, scale=scale)
    if code == WF43F:
        if up2x or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or cin % 16 or cout % 64:
            raise ValueError('winograd F(4,3) packing needs a 3x3 weight with cin % 16 == 0 and cout % 64 == 0 (no up2x)')
        packed = torch.empty(36 * cin * cout, dtype=torch.float32, device=w.device)
        L.check(lib.cf_pack_conv_weight_winograd43(L.ptr(w), cout, cin, cout, cin, L.ptr(packed, dtype=None), L.stream_ptr()),
                'cf_pack_conv_weight_winograd43')
        return PackedWeight(packed, b, cout, cin, 9, cout, cin, wino=2)
    if code == WF43:
        if up2x or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or cin % 16 or cout % 64:
            raise ValueError('winograd F(4,3) f16x2 packing needs a 3x3 weight with cin % 16 == 0 and cout % 64 == 0 (no up2x)')
        # max |G' g G'^T| for the power-of-two scale; G' = D^-1 G of the points (0, +-1/2, +-2, inf), the matrix cf_wf43.hip documents
        Gm = torch.tensor([[4.0, 0.0, 0.0], [-32 / 15, -16 / 15, -8 / 15], [-32 / 15, 16 / 15, -8 / 15], [1 / 15, 2 / 15, 4 / 15],
                           [1 / 15, -2 / 15, 4 / 15], [0.0, 0.0, 4.0]], dtype=torch.float64, device=w.device)
        umax = float(torch.einsum('xa,kcab,yb->kcxy', Gm, w.double(), Gm).abs().max())
        scale = 1.0 if umax == 0.0 or not math.isfinite(umax) else 2.0 ** (14 - math.frexp(umax)[1] + 1)
        packed = torch.empty(36 * cin * cout, dtype=torch.float32, device=w.device)
        L.check(lib.cf_pack_conv_weight_winograd43_f16x2(L.ptr(w), cout, cin, cout, cin, scale, L.ptr(packed, dtype=None), L.stream_ptr()),
                'cf_pack_conv_weight_winograd43_f16x2')
        return PackedWeight(packed, b, cout, cin, 9, cout, cin, bf16=OPERAND_F16X2, wino=2, scale=scale)
    if code in (WSPLIT, WF16, WBF16):
        if up2x or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or cin % 16 or cout % 64:
            raise ValueError('winograd f16x2 packing needs a 3x3 weight with cin % 16 == 0 and cout % 64 == 0 (no up2x)')
        # max |G g G^T| for the power-of-two scale (elementwise: G's rows are g0, (g0 + g1 + g2)/2, (g0 - g1 + g2)/2, g2), all 16 positions
        # in one tensor and ONE reduction / host read-back per weight
        g = w.double()
        r = torch.stack((g[:, :, 0], 0.5 * (g[:, :, 0] + g[:, :, 1] + g[:, :, 2]), 0.5 * (g[:, :, 0] - g[:, :, 1] + g[:, :, 2]), g[:, :, 2]), dim=2)
        u = torch.stack((r[..., 0], 0.5 * (r[..., 0] + r[..., 1] + r[..., 2]), 0.5 * (r[..., 0] - r[..., 1] + r[..., 2]), r[..., 2]), dim=3)
        umax = float(u.abs().max())
        scale = 1.0 if umax == 0.0 or not math.isfinite(umax) else 2.0 ** (14 - math.frexp(umax)[1] + 1)
        packed = torch.empty(16 * cin * cout, dtype=torch.float32, device=w.device)
        fn = 'cf_pack_conv_weight_winograd_bf16' if code == WBF16 else 'cf_pack_conv_weight_winograd_f16x2'   # (WF16 reads the hi slot)
        L.check(getattr(lib, fn)(L.ptr(w), cout, cin, cout, cin, scale, L.ptr(packed, dtype=None), L.stream_ptr()), fn)
        operand = {WSPLIT: OPERAND_F16X2, WF16: 2, WBF16: 1}[code]
        return PackedWeight(packed, b, cout, cin, 9, cout, cin, bf16=operand, wino=True, scale=scale)
    if stride2 and (code != SPLIT or up2x):
        raise ValueError('stride2 packing exists for the split-half kernel (bf16=SPLIT) only')
    if code == SPLIT and (w.dim() == 2 or tuple(w.shape[2:]) == (1, 1)):
        # 1x1 on images (the ResBlock skip convolutions): the streaming form of the split-half convolution kernel
        if up2x or cin % 32 or cout % 64:
            raise ValueError('f16x2 1x1 packing needs cin % 32 == 0 and cout % 64 == 0')
        wmax = float(w.abs().max())
        scale = 1.0 if wmax == 0.0 or not math.isfinite(wmax) else 2.0 ** (14 - math.frexp(wmax)[1] + 1)
        packed = torch.empty(cin * cout, dtype=torch.float32, device=w.device)
        L.check(lib.cf_pack_conv_weight_f16x2(L.ptr(w), cout, cin, 3, cout, cin, scale, L.ptr(packed, dtype=None), L.stream_ptr()),
                'cf_pack_conv_weight_f16x2')
        return PackedWeight(packed, b, cout, cin, 1, cout, cin, bf16=OPERAND_F16X2, scale=scale, conv1=True)
    if code == SPLIT:
        if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or cin % (16 if stride2 else 32) or cout % 64:
            raise ValueError('f16x2 packing needs a 3x3 weight with cin % 32 == 0 (stride 2: % 16) and cout % 64 == 0')
        # power-of-two scale that puts max|w'| (folded taps: at most 4 summed) into [2^14, 2^15): lo halves stay normal
        wmax = float(w.abs().max()) * (4.0 if up2x else 1.0)
        scale = 1.0 if wmax == 0.0 or not math.isfinite(wmax) else 2.0 ** (14 - math.frexp(wmax)[1] + 1)
        packed = torch.empty((16 if (up2x or stride2) else 9) * cin * cout, dtype=torch.float32, device=w.device)
        form = 2 if stride2 else int(bool(up2x))
        L.check(lib.cf_pack_conv_weight_f16x2(L.ptr(w), cout, cin, form, cout, cin, scale, L.ptr(packed, dtype=None), L.stream_ptr()),
                'cf_pack_conv_weight_f16x2')
        return PackedWeight(packed, b, cout, cin, 9, cout, cin, bf16=OPERAND_F16X2, up2x=bool(up2x), scale=scale, s2=bool(stride2))
    if w.dim() == 4:
        if w.shape[2] != w.shape[3] or w.shape[2] not in (1, 3):
            raise ValueError(f'unsupported kernel size {tuple(w.shape[2:])}')
        taps = w.shape[2] * w.shape[3]
    elif w.dim() == 2:
        taps = 1
    else:
        raise ValueError('weight must be 2-D or 4-D')
    if code or up2x:   # 16-bit operands and the folded upsample exist for 3x3 convs whose channels fill whole K slabs
        slab = 32 if code else 16
        if taps != 9 or cin % slab:
            raise ValueError(f"{('fp32', 'bf16', 'f16')[code]}{' up2x' if up2x else ''} packing needs a 3x3 weight with "
                             f'cin % {slab} == 0')
    if not code and not up2x:
        cout_pad, cin_pad = _cout_pad(cout), (cin + 15) // 16 * 16
        packed = torch.empty(lib.cf_packed_weight_elems(cin_pad, taps, cout_pad), dtype=torch.float32, device=w.device)
        L.check(lib.cf_pack_conv_weight(L.ptr(w), cout, cin, taps, cout_pad, cin_pad, L.ptr(packed, dtype=None), L.stream_ptr()),
                'cf_pack_conv_weight')
        return PackedWeight(packed, b, cout, cin, taps, cout_pad, cin_pad)
    # bf16 kernels and every upsample kernel have N tiles of at least 64; the f16 general instantiation also has a 32-wide one
    cout_pad = _cout_pad(cout) if (code == 2 and not up2x) else max(64, _cout_pad(cout))
    dtype = (torch.float32, torch.bfloat16, torch.float16)[code]
    packed = torch.empty((16 if up2x else 9) * cin * cout_pad, dtype=dtype, device=w.device)
    name = 'cf_pack_conv_weight' + ('_up2x' if up2x else '') + ('', '_bf16', '_f16')[code]
    args = (L.ptr(w), cout, cin) + (() if up2x else (9,)) + (cout_pad, cin, L.ptr(packed, dtype=None), L.stream_ptr())
    L.check(getattr(lib, name)(*args), name)
    return PackedWeight(packed, b, cout, cin, 9, cout_pad, cin, bf16=code, up2x=bool(up2x))


def pack_weight_cat(weights, biases):
    """Row-concatenate several (cout_i, cin[,1,1]) weights into one GEMM (e.g. q|k|v)."""
    w = torch.cat([x.detach().reshape(x.shape[0], x.shape[1]) for x in weights], dim=0)
    b = None if biases is None else torch.cat([x.detach() for x in biases], dim=0)
    return pack_weight(w, b)


def _nhwc_ld(t, what):
    """Channel stride (floats per pixel) of an NHWC tensor that is dense or a channel slice `buf[..., a:b]` of a dense one."""
    B, H, W, C = t.shape
    ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else max(C, t.stride(0) // max(H * W, 1)))
    ok = t.stride(3) == 1 and ld >= C and (W == 1 or t.stride(2) == ld) and (H == 1 or t.stride(1) == W * ld) and \
        (B == 1 or t.stride(0) == H * W * ld)
    if not ok:
        raise ValueError(f'{what}: expected a dense NHWC tensor or a channel slice of one, got shape {tuple(t.shape)} '
                         f'strides {t.stride()}')
    return ld


# Split-K (cf_conv_desc.split_k) for 1x1 / Linear layers on small token images: WHETHER a layer takes the split-K kernel (64x64
# tiles, K cut into virtual chunks of 128 added in a fixed order) depends on its per-image shape only, so a face's bits are the same
# alone and inside any batch; HOW MANY workgroups then share a tile's chunks is chosen from the number of tiles in flight and does
# not change the result (see cf_common.h).  CODEFORMER_HIP_SPLITK = largest split count (0: layers keep the large-tile kernel).
SPLITK_MAX = int(os.environ.get('CODEFORMER_HIP_SPLITK', '8'))
SPLITK_IN_WORKGROUP = -1    # CF_SPLITK_IN_WORKGROUP of the header (ABI v21): split-half token GEMMs of one to a few faces
# Token GEMMs (split-half operands) of at most this many OUTPUTS (rows x columns) take the in-workgroup split: one to four faces (256 tokens
# each) for every layer, eight faces for the 512-column layers.  Measured per launch inside a captured graph (tools/gemm_chunk_probe.py):
# 256 x 512 x 512: 5.6 us against 15.0 (cross-workgroup split), 1024 x 512 x 1024: 13.9 / 17.2, 2048 x 1024 x 512: 22.3 / 30.2 -- and 4096 x 512 x
# 1024: 34.1 against 26.7 for the token-tile kernel, which keeps the large launches.  In the network (tools/latency.py, graph replay, one
# box): one face 6.71 -> 6.35 ms, two 8.34 -> 7.93, four 11.66 -> 11.38.
GEMM_IN_WG_MAX_OUTPUTS = int(os.environ.get('CODEFORMER_HIP_GEMM_IN_WG_OUTPUTS', str(1 << 20)))
_COUNTERS = {}


def splitk_for(pw, ho, wo, cin, batch=1):
    """Split count for a 1x1 / Linear or a Winograd 3x3 on `batch` images of ho x wo pixels; 0: the layer is not a split-K layer."""
    if SPLITK_MAX <= 0 or (pw.bf16 and not pw.wino and pw.taps != 1) or ho * wo > 1024 or cin % 128:
        return 0
    if pw.wino == 2:
        return 0   # (F(4x4,3x3) has no split-K form)
    if pw.wino:
        if ho % 8 or wo % 16 or ho * wo > 256:   # the 16x16 latents only: from 32x32 up a batch fills the CUs without splitting
            return 0
        tiles = batch * (ho // 8) * (wo // 16) * (pw.cout_pad // 64)
    elif pw.taps == 1 and (ho * wo) % 64 == 0 and pw.cout_pad % 64 == 0:
        if int(pw.bf16) == OPERAND_F16X2 and not pw.conv1 and batch * ho * wo * pw.cout_pad <= GEMM_IN_WG_MAX_OUTPUTS and cin <= 1024:
            return SPLITK_IN_WORKGROUP   # few tokens: the chunks of a 32x32 tile shared by the waves of one workgroup (cf_gemm_split.hip: same bits)
        tiles = batch * (ho * wo // 64) * (pw.cout_pad // 64)
    else:
        return 0
    # Workgroups a split launch may occupy (tools/sk_probe2.py, MI355X): the four-wave Winograd kernel runs two workgroups per CU, but
    # a second round of chain + finish costs more than the shorter chain saves (16 faces: 1 workgroup per tile 73 us, 2 -> 99 us);
    # the Linear kernel's finish outweighs its short K chain beyond 128 workgroups (8 faces, 512 -> 512: 1 -> 21 us, 2 -> 34 us).
    cap = 256 if pw.wino else 128
    v = cin // 128
    for ns in (8, 4, 2):
        if ns <= SPLITK_MAX and v % ns == 0 and tiles * ns <= cap:
            return ns
    return 1


def _counters(device, n):
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _COUNTERS.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 4096), dtype=torch.int32, device=device)   # every split-K launch leaves its counters at zero again
        _COUNTERS[key] = t
    return t


def conv2d(x, pw, *, x2=None, stride=1, upsample=False, prologue=PRO_NONE, scale=None, shift=None,
           epilogue=EPI_NONE, res=None, sft_scale=None, sft_w=0.0, in_nchw=False, out_nchw=False, emit_stats=False,
           out=None, pad_mode=PAD_ZERO, pad_lo=0, split_k=None, act=None, x_alt=None, alt_from=0, stats_into=None):
    """Implicit-GEMM conv (3x3 / 1x1).  x: (B,H,W,C0) [x2: (B,H,W,C1) concatenated after x]; returns (B,Ho,Wo,cout)
    (or (B,cout,Ho,Wo) when out_nchw).  With in_nchw, x is (B,C<=4,H,W).
    emit_stats: also write the GroupNorm(32) partial statistics of the output in the epilogue and attach them to the
    returned tensor (`._cf_stats`), so a following groupnorm_tables() does not re-read the tensor.
    x, x2 and `out` (optional destination) may be channel slices `buf[..., a:b]` of wider NHWC buffers -- the dense-block
    pattern of RRDBNet, where torch.cat never materialises; res / sft_scale (= res2 of EPI_AXPY2) then share out's stride.
    EPI_LEAKY / EPI_AXPY / EPI_AXPY2 use sft_w as alpha (see cf_epilogue in the header).
    pad_mode: PAD_ZERO, PAD_REFLECT (ReflectionPad2d(1) + unpadded 3x3) or PAD_EDGE (with upsample: reflection padding of the
    upsampled image); pad_lo=1 with stride 2: one padded row / column on every side instead of right / bottom only.
    act: (B, 2) range-scale table of an un-normalised input (act_scale(x)); dropped when the layer's kernel has fp32 operands.
    stats_into: a StatsPair -- the statistics partials of this launch go into its shared buffer (two sibling convolutions whose range-scale
    tables are then ONE launch: act_scale_pair).
    x_alt / alt_from (split-half token GEMMs only): output columns >= alt_from read their rows from x_alt (same shape as x) -- two Linear
    layers on two token matrices in one launch (q|k on LN(x) + pos, v on LN(x))."""
    lib = L.load()
    io = _act_dtype(x, 'x')            # float32, or bfloat16 = bf16 storage of every activation of the launch (io_bf16)
    if in_nchw:
        _f32(x)
        B, c0, H, W = x.shape
        ld0 = 0
        if not x.is_contiguous():
            raise ValueError('in_nchw input must be contiguous')
    else:
        B, H, W, c0 = x.shape
        ld0 = _nhwc_ld(x, 'x')
    c1 = ld1 = 0
    if x2 is not None:
        if x2.shape[:3] != x.shape[:3]:
            raise ValueError('x2 spatial shape mismatch')
        c1 = x2.shape[3]
        if x2.dtype != io:
            raise TypeError(f'x2 is {x2.dtype}, x is {io}: both halves of a concatenated input share the storage type (ops.to_bf16)')
        ld1 = _nhwc_ld(x2, 'x2')
    if c0 + c1 != pw.cin and not (c1 == 0 and c0 == pw.cin_pad):
        raise ValueError(f'input channels {c0}+{c1} != weight cin {pw.cin}')
    f43_up = bool(upsample) and pw.wino == 2 and not pw.bf16     # fp32 F(4,3) with the upsampling gather: the PLAIN 3x3 packing (WF43F), not the folded one
    if bool(upsample) != bool(pw.up2x) and not f43_up:
        raise ValueError('conv2d(upsample=True) needs a weight packed with up2x=True, or the fp32 F(4,3) packing (and vice versa)')
    if pw.taps == 1 and int(pw.bf16) == OPERAND_F16X2 and bool(pw.conv1) != (H * W > TOKEN_IMAGE_MAX):
        raise ValueError(f'1x1 with f16x2 operands: images of more than {TOKEN_IMAGE_MAX} pixels take a weight packed with bf16=SPLIT, '
                         'token matrices one packed with bf16=GSPLIT')
    if bool(pw.s2) != (stride == 2 and int(pw.bf16) == OPERAND_F16X2):
        raise ValueError('a weight packed with stride2=True serves conv2d(stride=2) only (and f16x2 operands at stride 2 need it)')
    if stride == 2:
        Ho, Wo = H // 2, W // 2
    else:
        Ho, Wo = (H * 2, W * 2) if upsample else (H, W)
    shape = (B, pw.cout, Ho, Wo) if out_nchw else (B, Ho, Wo, pw.cout)
    ldo = 0
    odt = torch.float32 if out_nchw else io      # (the NCHW network output stays fp32)
    if out is None:
        out = torch.empty(shape, dtype=odt, device=x.device)
    else:
        if tuple(out.shape) != shape or out.dtype != odt or out.device != x.device:
            raise ValueError(f'out: expected {odt} {shape} on {x.device}')
        if out_nchw:
            if not out.is_contiguous():
                raise ValueError('out_nchw destination must be contiguous')
        else:
            ldo = _nhwc_ld(out, 'out')
    for t in (res, sft_scale):
        if t is not None and tuple(t.shape) != (B, Ho, Wo, pw.cout):
            raise ValueError(f'epilogue operand shape {tuple(t.shape)} != {(B, Ho, Wo, pw.cout)}')
        if t is not None and t.dtype != io:
            raise TypeError(f'epilogue operand is {t.dtype}, the launch stores {io}')
        if t is not None and _nhwc_ld(t, 'epilogue operand') != (ldo or pw.cout):
            raise ValueError('epilogue operands must have the channel stride of the output')
    for t in (scale, shift):
        if t is not None and tuple(t.shape) != (B, c0 + c1):
            raise ValueError(f'prologue table shape {tuple(t.shape)} != {(B, c0 + c1)}')
    d = L.ConvDesc(
        in0=L.ptr(x, not in_nchw, dtype=io), in1=L.ptr(x2, True, dtype=io), c0=c0, c1=c1, batch=B, hin=H, win=W, hout=Ho, wout=Wo, cout=pw.cout,
        cout_pad=pw.cout_pad, taps=pw.taps, stride=stride, upsample=int(bool(upsample)), in_nchw=int(bool(in_nchw)),
        out_nchw=int(bool(out_nchw)), prologue=prologue, epilogue=epilogue, pro_scale=L.ptr(scale),
        pro_shift=L.ptr(shift), weight=L.ptr(pw.w, dtype=None), bias=L.ptr(pw.bias), res=L.ptr(res, True, dtype=io),
        sft_scale=L.ptr(sft_scale, True, dtype=io), sft_w=float(sft_w), out=L.ptr(out, not out_nchw, dtype=odt), bf16_mfma=int(pw.bf16),
        ld_in0=ld0, ld_in1=ld1, ld_out=ldo, pad_mode=int(pad_mode), pad_lo=int(pad_lo), winograd=int(pw.wino),
        acc_scale=1.0 / pw.scale, io_bf16=int(io == torch.bfloat16))
    if x_alt is not None:
        if tuple(x_alt.shape) != tuple(x.shape) or x_alt.dtype != torch.float32 or int(pw.bf16) != OPERAND_F16X2 or pw.taps != 1 or pw.conv1 or x2 is not None:
            raise ValueError('x_alt: a second float32 token matrix of the same shape, for a split-half token GEMM (bf16=GSPLIT weight)')
        d.in0_alt, d.alt_cout0 = L.ptr(x_alt), int(alt_from)
    if act is not None and needs_act_scale(pw):
        if tuple(act.shape) != (B, 2) or prologue not in (PRO_NONE, PRO_LEAKY):
            raise ValueError('act: expected a (B, 2) table and a none / leaky prologue')
        d.act_scale = L.ptr(act)
    if split_k is None:
        dense = not in_nchw and not out_nchw and ld0 == c0 and (c1 == 0 or ld1 == c1) and ldo in (0, pw.cout)
        split_k = splitk_for(pw, Ho, Wo, c0 + c1, B) if (stride == 1 and dense) else 0
    if split_k:
        d.split_k = int(split_k)
        if split_k > 1:
            nbytes, tiles = lib.cf_conv2d_workspace_bytes(ctypes.byref(d)), lib.cf_conv2d_tiles(ctypes.byref(d))
            if nbytes < 0 or tiles <= 0:
                raise RuntimeError(f'cf_conv2d_workspace_bytes failed: {L.last_error()}')
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
            d.workspace, d.counters = L.ptr(ws), L.ptr(_counters(x.device, tiles), dtype=torch.int32)
    if emit_stats and not out_nchw and pw.cout % GN_GROUPS == 0 and pw.cout // GN_GROUPS >= 2:
        d.stats_cpg = pw.cout // GN_GROUPS
        parts = lib.cf_conv2d_stats_parts(ctypes.byref(d))
        if parts <= 0:
            raise RuntimeError(f'cf_conv2d_stats_parts failed ({parts}): {L.last_error()}')
        n_part = B * GN_GROUPS * parts * 2
        part = torch.empty(n_part, dtype=torch.float64, device=x.device) if stats_into is None else stats_into.take(n_part, x.device)
        d.stats_out = L.ptr(part, dtype=torch.float64)
        out._cf_stats = GNStats(part, parts, d.stats_cpg)
    if PROFILE is None:
        L.check(lib.cf_conv2d(ctypes.byref(d), L.stream_ptr()), 'cf_conv2d')
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.cf_conv2d(ctypes.byref(d), L.stream_ptr()), 'cf_conv2d')
    e1.record()
    cin = c0 + c1
    flops = 2.0 * B * Ho * Wo * pw.cout * cin * (4 if (upsample and not pw.wino) else pw.taps)   # executed MACs (folded taps for up2x); Winograd
    # launches are booked at the direct convolution's 9 taps (the algorithmic work), not at their 4 MFMA multiplies per output
    esz = x.element_size()
    nbytes = float(esz * (x.numel() + (0 if x2 is None else x2.numel()) + (0 if res is None else res.numel()) + (0 if sft_scale is None else sft_scale.numel()))
                   + 4 * pw.cout * cin * pw.taps + out.element_size() * out.numel())
    kind = ('conv3x3_s2' if stride == 2 else ('conv_up2x' if upsample else ('conv3x3_wino' if pw.wino else 'conv3x3'))) \
        if pw.taps == 9 else 'gemm1x1'
    if pw.taps == 9 and stride == 1 and ((in_nchw and c0 <= 4) or (out_nchw and pw.cout <= 4)):
        kind = 'conv3x3_io'   # the network's first / last conv: vector-ALU kernels, HBM-bound (conv3x3_few_cin / few_cout)
    if pw.bf16:
        kind += ('', '_bf16', '_f16', '_f16x2')[int(pw.bf16)]
    if io == torch.bfloat16 and not pw.bf16:
        kind += '_bf16io'     # fp32 operands on bf16 tensors (the 1x1 skips and the RGB head of the bf16 mode)
    if pw.conv1:
        kind = 'conv1x1_stream_f16x2'   # 1x1 on images through the split-half convolution kernel (HBM-bound), not the token GEMM
    if pw.wino == 2:
        kind = 'conv3x3_wino43_f16x2' if pw.bf16 else 'conv3x3_wino43'   # F(4x4,3x3) on split halves / on fp32 operands (cf_wf43.hip)
    elif pw.wino and pw.bf16 and pw.cout % 128 == 0 and Ho * Wo >= 1024 and not split_k:
        kind += '_8w'      # the eight-wave 128-channel kernel (cf_wsplit.hip; the rule of cf_wsplit_covers)
    PROFILE.append((kind, flops, nbytes, e0, e1, (B, H, W, cin, pw.cout)))
    return out


def linear(x, pw, *, epilogue=EPI_NONE, res=None, x_alt=None, alt_from=0):
    """x: (M, K) -> (M, N) through the 1x1 path of the same kernel (M must be a multiple of 256).
    x_alt / alt_from: see conv2d (columns >= alt_from are x_alt @ W^T: two Linear layers, two inputs, one launch)."""
    M, K = x.shape
    if M % 256:
        raise ValueError(f'linear: rows {M} must be a multiple of 256')
    # present the token matrix as (M/256) "images" of 16x16 tokens: tile selection keys on the per-image shape only
    r4 = None if res is None else res.view(M // 256, 16, 16, pw.cout)
    y = conv2d(x.view(M // 256, 16, 16, K), pw, epilogue=epilogue, res=r4, x_alt=None if x_alt is None else x_alt.view(M // 256, 16, 16, K), alt_from=alt_from)
    return y.view(M, pw.cout)


FINALIZE_FUSED = os.environ.get('CODEFORMER_HIP_FINALIZE_FUSED', '1') != '0'   # 0: one cf_groupnorm_finalize per tensor + separate range-scale launches (A/B; same bits)


def groupnorm_tables(xs, gamma, beta, eps=GN_EPS, groups=GN_GROUPS, act_growth=None):
    """GroupNorm(groups) statistics of the channel-concatenation of xs (each (B,H,W,Ci)) folded with the affine
    parameters into per-(b,c) scale / shift tables (B, sum Ci), to be applied by a conv prologue.

    Tensors that carry epilogue statistics (`._cf_stats`, see conv2d(emit_stats=True)) are not read again; the others
    go through the stand-alone statistics pass.
    act_growth: the caller will also ask act_scale(*xs, growth=act_growth) (the block input that feeds both norm1 and the 1x1 skip
    convolution): when every tensor carries statistics the SAME launch writes that table (cf_groupnorm_finalize2) and parks it where
    act_scale finds it -- bitwise the table of the separate launch."""
    lib = L.load()
    B, H, W, _ = xs[0].shape
    ctot = sum(t.shape[3] for t in xs)
    if ctot % groups:
        raise ValueError(f'{ctot} channels not divisible by {groups} groups')
    cpg = ctot // groups
    hw = H * W
    dev = xs[0].device
    scale = torch.empty(B, ctot, dtype=torch.float32, device=dev)
    shift = torch.empty_like(scale)
    g_ptr, b_ptr, sc_ptr, sh_ptr = L.ptr(gamma), L.ptr(beta), L.ptr(scale), L.ptr(shift)
    stats = []
    for t in xs:
        _act_dtype(t)
        c = t.shape[3]
        if c % cpg:
            raise ValueError('concat boundary splits a group')
        st = getattr(t, '_cf_stats', None)
        if st is None or cpg % st.cpg:
            # ~128 KB of input per block, enough blocks to cover 256 CUs several times, at most 256 partials per group
            nblk = max(1, min(256, (hw * c * 4 + (1 << 17) - 1) >> 17))
            part = torch.empty(B * (c // cpg) * nblk * 2, dtype=torch.float64, device=dev)
            _f32(t)   # (the stand-alone pass reads fp32 tensors; bf16 tensors always carry the partials of their producer)
            L.check(lib.cf_groupnorm_stats(L.ptr(t), B, hw, c, cpg, L.ptr(part, dtype=torch.float64), nblk, L.stream_ptr()), 'cf_groupnorm_stats')
            st = GNStats(part, nblk, cpg)
            stats.append((st, False))
        else:
            stats.append((st, True))
    if FINALIZE_FUSED and len(xs) <= 2:
        # one launch for the tensor (or both halves of the concatenation), and the range-scale table with it when every half carries
        # its producer's statistics (that is where act_scale would take the bound from as well)
        want_act = act_growth is not None and RANGE_SCALE and ACT_FUSED and all(own for _, own in stats)
        act = torch.empty(B, 2, dtype=torch.float32, device=dev) if want_act else None
        cells = L.ptr(_act_cells(dev, B), dtype=torch.int32) if want_act else None
        sa, ca = stats[0][0], xs[0].shape[3]
        sb, cb = (stats[1][0], xs[1].shape[3]) if len(xs) == 2 else (None, 0)
        L.check(lib.cf_groupnorm_finalize2(L.ptr(sa.part, dtype=torch.float64), sa.parts, ca, sa.cpg, cpg // sa.cpg,
                                           None if sb is None else L.ptr(sb.part, dtype=torch.float64), 0 if sb is None else sb.parts, cb,
                                           1 if sb is None else sb.cpg, 1 if sb is None else cpg // sb.cpg, B, hw * cpg, g_ptr, b_ptr, float(eps),
                                           sc_ptr, sh_ptr, ctot, float(act_growth or 0.0), cells, L.ptr(act), L.stream_ptr()), 'cf_groupnorm_finalize2')
        if want_act:
            _park_act(xs, float(act_growth), act)
        return scale, shift
    coff = 0
    for t, (st, _) in zip(xs, stats):
        c = t.shape[3]
        L.check(lib.cf_groupnorm_finalize(L.ptr(st.part, dtype=torch.float64), B, st.parts, c, st.cpg, cpg // st.cpg, hw * cpg,
                                          g_ptr + 4 * coff, b_ptr + 4 * coff, float(eps), sc_ptr + 4 * coff,
                                          sh_ptr + 4 * coff, ctot, L.stream_ptr()), 'cf_groupnorm_finalize')
        coff += c
    return scale, shift


def _act_key(x, growth):
    ver = tensor_version(x)
    if ver is None and getattr(x, '_cf_stats', None) is not None:
        ver = 'inference'     # (see act_scale: a conv output of this forward)
    return None if ver is None else (float(growth), ver)


def _park_act(xs, growth, act):
    """Leave a range-scale table computed elsewhere (groupnorm_tables) where act_scale(x[, x2]) looks first."""
    k0 = _act_key(xs[0], growth)
    if k0 is None:
        return
    if len(xs) == 1:
        xs[0]._cf_act = (k0, act)
    else:
        k1 = _act_key(xs[1], growth)
        if k1 is not None:
            xs[0]._cf_act_pair = (k0, k1, weakref.ref(xs[1]), act)    # (a weak reference, not an id: ids are recycled)


def layernorm(x, gamma, beta, eps=1e-5, pos=None):
    """x: (rows, C).  Returns LN(x) and, when pos (npos, C) is given, also LN(x)+pos[row % npos]."""
    lib = L.load()
    rows, C = x.shape
    y = torch.empty_like(x)
    ypos = torch.empty_like(x) if pos is not None else None
    npos = 0 if pos is None else pos.shape[0]
    L.check(lib.cf_layernorm(L.ptr(_f32(x)), rows, C, L.ptr(gamma), L.ptr(beta), float(eps), L.ptr(pos), npos, L.ptr(y),
                             L.ptr(ypos), L.stream_ptr()), 'cf_layernorm')
    return (y, ypos) if pos is not None else y


def attention(q, k, v, batch, heads, head_dim, scale):
    """q,k,v: 2-D views (batch*256, >= heads*head_dim) that may be column slices of wider row-major matrices
    (row stride taken from .stride(0)).  Returns (batch*256, heads*head_dim)."""
    lib = L.load()
    rows = batch * 256
    for t in (q, k, v):
        if t.shape[0] != rows or t.stride(1) != 1 or t.shape[1] != heads * head_dim:
            raise ValueError('attention operand must be (batch*256, heads*head_dim) with unit column stride')
    out = torch.empty(rows, heads * head_dim, dtype=torch.float32, device=q.device)
    L.check(lib.cf_attention(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                             L.ptr(out), out.stride(0), batch, heads, head_dim, 256, float(scale), L.stream_ptr()),
            'cf_attention')
    return out


def argmax_rows(logits):
    lib = L.load()
    rows, n = logits.shape
    idx = torch.empty(rows, dtype=torch.int64, device=logits.device)
    L.check(lib.cf_argmax_rows(L.ptr(_f32(logits)), rows, n, L.ptr(idx, dtype=torch.int64), L.stream_ptr()), 'cf_argmax_rows')
    return idx


def codebook_gather(idx, codebook, batch, ntok, lq=None, eps=1e-5):
    """idx: (batch*ntok,) int64 -> (batch, ntok, dim); AdaIN against lq (batch, ntok, dim) when given."""
    lib = L.load()
    ncodes, dim = codebook.shape
    out = torch.empty(batch, ntok, dim, dtype=torch.float32, device=codebook.device)
    L.check(lib.cf_codebook_gather_adain(L.ptr(idx, dtype=torch.int64), L.ptr(_f32(codebook)), ncodes, L.ptr(lq), batch, ntok, dim,
                                         int(lq is not None), float(eps), L.ptr(out), L.stream_ptr()),
            'cf_codebook_gather_adain')
    return out


def vq_nearest(z_tokens, codebook, pw_codebook=None):
    """z_tokens: (rows, dim), rows % 256 == 0.  Returns (idx int64 (rows,), dmin (rows,), (scores, zz, ee))."""
    lib = L.load()
    rows, dim = z_tokens.shape
    ncodes = codebook.shape[0]
    pw = pw_codebook or pack_weight(codebook)
    scores = linear(z_tokens, pw)
    zz = torch.empty(rows, dtype=torch.float32, device=z_tokens.device)
    ee = torch.empty(ncodes, dtype=torch.float32, device=z_tokens.device)
    L.check(lib.cf_row_sqnorm(L.ptr(z_tokens), rows, dim, L.ptr(zz), L.stream_ptr()), 'cf_row_sqnorm')
    L.check(lib.cf_row_sqnorm(L.ptr(_f32(codebook)), ncodes, dim, L.ptr(ee), L.stream_ptr()), 'cf_row_sqnorm')
    idx = torch.empty(rows, dtype=torch.int64, device=z_tokens.device)
    dmin = torch.empty(rows, dtype=torch.float32, device=z_tokens.device)
    L.check(lib.cf_vq_argmin(L.ptr(scores), L.ptr(zz), L.ptr(ee), rows, ncodes, L.ptr(idx, dtype=torch.int64), L.ptr(dmin), L.stream_ptr()),
            'cf_vq_argmin')
    return idx, dmin, (scores, zz, ee)


def to_nhwc(x):
    """(B,C,H,W) -> (B,H,W,C) by the HIP transpose kernel."""
    lib = L.load()
    B, C, H, W = x.shape
    x = _f32(x).contiguous()
    y = torch.empty(B, H, W, C, dtype=torch.float32, device=x.device)
    L.check(lib.cf_nchw_to_nhwc(L.ptr(x), B, C, H * W, L.ptr(y), L.stream_ptr()), 'cf_nchw_to_nhwc')
    return y


def pixel_unshuffle_nhwc(x, scale, c_pad=None):
    """(B,C,H*s,W*s) NCHW -> (B,H,W,c_pad) channels-last pixel-unshuffle (arch_util.py:190-206), channels zero-padded to a
    multiple of 16 so the result feeds conv2d directly.  scale=1 is a padded layout change."""
    lib = L.load()
    B, C, Hs, Ws = x.shape
    if Hs % scale or Ws % scale:
        raise ValueError(f'pixel_unshuffle: {Hs}x{Ws} not divisible by {scale}')
    cu = C * scale * scale
    c_pad = (cu + 15) // 16 * 16 if c_pad is None else c_pad
    x = _f32(x).contiguous()
    y = torch.empty(B, Hs // scale, Ws // scale, c_pad, dtype=torch.float32, device=x.device)
    L.check(lib.cf_pixel_unshuffle_nhwc(L.ptr(x), B, C, Hs // scale, Ws // scale, scale, c_pad, L.ptr(y), L.stream_ptr()),
            'cf_pixel_unshuffle_nhwc')
    return y


def to_nchw(x):
    """(B,H,W,C) -> (B,C,H,W)."""
    lib = L.load()
    B, H, W, C = x.shape
    y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    L.check(lib.cf_nhwc_to_nchw(L.ptr(_f32(x)), B, C, H * W, L.ptr(y), L.stream_ptr()), 'cf_nhwc_to_nchw')
    return y


def img_u8_to_tensor(img):
    """uint8 (B,H,W,3) BGR on device -> fp32 (B,3,H,W) RGB in [-1,1]."""
    lib = L.load()
    B, H, W, _ = img.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=img.device)
    L.check(lib.cf_img_u8_to_tensor(L.ptr(img, dtype=torch.uint8), B, H, W, L.ptr(out), L.stream_ptr()), 'cf_img_u8_to_tensor')
    return out


def tensor_to_img_u8(t):
    """fp32 (B,3,H,W) RGB -> uint8 (B,H,W,3) BGR with tensor2img(min_max=(-1,1)) rounding."""
    lib = L.load()
    B, _, H, W = t.shape
    img = torch.empty(B, H, W, 3, dtype=torch.uint8, device=t.device)
    L.check(lib.cf_tensor_to_img_u8(L.ptr(_f32(t).contiguous()), B, H, W, L.ptr(img, dtype=torch.uint8), L.stream_ptr()), 'cf_tensor_to_img_u8')
    return img


def mask_composite(x, y):
    """Inpainting composite: where the normalised input pixel is pure white (x0+x1+x2 == 3) take y, else keep x."""
    lib = L.load()
    B, _, H, W = x.shape
    out = torch.empty_like(x)
    L.check(lib.cf_mask_composite(L.ptr(_f32(x).contiguous()), L.ptr(_f32(y).contiguous()), B, H, W, L.ptr(out), L.stream_ptr()),
            'cf_mask_composite')
    return out


_FBA_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def fused_bias_act(x, bias, negative_slope=0.2, scale=math.sqrt(2.0), ref=None, act=3, grad=0):
    """x: (N,C,...) NCHW-style; every mode of the reference op (fused_bias_act_kernel.cu:20-50): act 1 linear | 3 leaky ReLU; grad 0
    forward y = act(x + bias[c]) * scale, 1 first derivative gated by the sign of `ref` (the forward output), 2 zeros.
    float32 / float16 / bfloat16; bias / ref may be None (or empty, as the reference passes them)."""
    lib = L.load()
    if x.dtype not in _FBA_DTYPES:
        raise TypeError(f'fused_bias_act: float32 / float16 / bfloat16 tensors (got {x.dtype})')
    x = x.contiguous()
    bias = None if bias is None or bias.numel() == 0 else bias.detach().to(x.dtype).contiguous()
    ref = None if ref is None or ref.numel() == 0 else ref.detach().to(x.dtype).contiguous()
    if ref is not None and ref.shape != x.shape:
        raise ValueError('fused_bias_act: ref must have the shape of x')
    if bias is not None and bias.numel() != x.shape[1]:
        raise ValueError('fused_bias_act: bias must have one value per channel (dim 1)')
    hw = 1
    for s in x.shape[2:]:
        hw *= s
    y = torch.empty_like(x)
    L.check(lib.cf_fused_bias_act_ex(x.data_ptr(), 0 if bias is None else bias.data_ptr(), 0 if ref is None else ref.data_ptr(),
                                     x.numel(), x.shape[1] if x.dim() > 1 else 1, hw, int(act), int(grad), float(negative_slope),
                                     float(scale), _FBA_DTYPES[x.dtype], y.data_ptr(), L.stream_ptr()), 'cf_fused_bias_act_ex')
    return y


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """x: (N,C,H,W); kernel: (kh,kw)."""
    lib = L.load()
    x = _f32(x).contiguous()
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    oh = (h * up + p0 + p1 - kh) // down + 1
    ow = (w * up + p0 + p1 - kw) // down + 1
    y = torch.empty(n, c, oh, ow, dtype=torch.float32, device=x.device)
    L.check(lib.cf_upfirdn2d(L.ptr(x), n * c, h, w, L.ptr(_f32(kernel).contiguous()), kh, kw, up, up, down, down, p0, p1,
                             p0, p1, L.ptr(y), L.stream_ptr()), 'cf_upfirdn2d')
    return y


# ---- alignment warp / paste-back primitives (cf_paste.hip) -- see codeformer_amd/facelib/paste.py for the orchestration --------------
def _c6(m):
    """6 doubles in host memory for the ctypes call (a 2x3 destination->source matrix)."""
    vals = [float(v) for v in (m.reshape(-1).tolist() if hasattr(m, 'reshape') else m)]
    if len(vals) != 6:
        raise ValueError('affine matrix must have 6 elements')
    return (ctypes.c_double * 6)(*vals)


def warp_affine_u8(src, inv_dev, dst, region=None, border=(0, 0, 0)):
    """src: uint8 (sh,sw,3) (shared) or (n,sh,sw,3); inv_dev: float64 CUDA (n,6) destination->source matrices; dst: uint8 (n,dh,dw,3)
    written inside region = (rx, ry, rw, rh) (default: everything)."""
    lib = L.load()
    n, dh, dw, _ = dst.shape
    shared = src.dim() == 3
    sh, sw = (src.shape[0], src.shape[1]) if shared else (src.shape[1], src.shape[2])
    if (not shared and src.shape[0] != n) or tuple(inv_dev.shape) != (n, 6):
        raise ValueError('warp_affine_u8: batch mismatch')
    rx, ry, rw, rh = region or (0, 0, dw, dh)
    L.check(lib.cf_warp_affine_u8(L.ptr(src, dtype=torch.uint8), 0 if shared else sh * sw * 3, sh, sw, L.ptr(inv_dev, dtype=torch.float64), n,
                                  L.ptr(dst, dtype=torch.uint8), dh, dw, rx, ry, rw, rh, int(border[0]), int(border[1]), int(border[2]),
                                  L.stream_ptr()), 'cf_warp_affine_u8')
    return dst


def warp_affine_f32(src, inv, region):
    lib = L.load()
    rx, ry, rw, rh = region
    out = torch.empty(rh, rw, dtype=torch.float32, device=src.device)
    L.check(lib.cf_warp_affine_f32(L.ptr(src), src.shape[0], src.shape[1], _c6(inv), L.ptr(out), rx, ry, rw, rh, L.stream_ptr()),
            'cf_warp_affine_f32')
    return out


def erode(x, k):
    lib = L.load()
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.cf_erode_f32(L.ptr(x), L.ptr(tmp), L.ptr(out), x.shape[0], x.shape[1], int(k), L.stream_ptr()), 'cf_erode_f32')
    return out


def gaussian_blur(x, taps_dev, region_xy=(0, 0), frame_hw=None):
    """x: (rh,rw) float32 region at (rx, ry) of a frame of size frame_hw (default: the region is the frame); taps_dev: CUDA float32."""
    lib = L.load()
    ch, cw = frame_hw or (x.shape[0], x.shape[1])
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.cf_gaussian_blur_f32(L.ptr(x), L.ptr(tmp), L.ptr(out), x.shape[0], x.shape[1], int(region_xy[0]), int(region_xy[1]), ch, cw,
                                     L.ptr(taps_dev), taps_dev.numel(), L.stream_ptr()), 'cf_gaussian_blur_f32')
    return out


def sum_partials(x, out64):
    """64 fp64 partial sums of x into out64 (CUDA float64, 64 elements); the caller adds them after its read-back."""
    lib = L.load()
    L.check(lib.cf_sum_f32(L.ptr(x), x.numel(), L.ptr(out64, dtype=torch.float64), L.stream_ptr()), 'cf_sum_f32')


def paste_blend(canvas, face, inv, ero, soft, region, parse=None):
    lib = L.load()
    rx, ry, rw, rh = region
    L.check(lib.cf_paste_blend(L.ptr(canvas), canvas.shape[0], canvas.shape[1], L.ptr(face, dtype=torch.uint8), face.shape[0], face.shape[1],
                               _c6(inv), L.ptr(ero), L.ptr(soft), L.ptr(parse), rx, ry, rw, rh, L.stream_ptr()), 'cf_paste_blend')


def resize_linear_u8(src, dh, dw):
    lib = L.load()
    out = torch.empty(dh, dw, 3, dtype=torch.float32, device=src.device)
    L.check(lib.cf_resize_linear_u8(L.ptr(src, dtype=torch.uint8), src.shape[0], src.shape[1], L.ptr(out), dh, dw, L.stream_ptr()),
            'cf_resize_linear_u8')
    return out


def resize_area_u8(src, dh, dw):
    """cv2.resize(src uint8 (sh, sw, 3) on the device, (dw, dh), INTER_AREA) -> uint8 (dh, dw, 3); shrinking only."""
    lib = L.load()
    if src.dim() != 3 or src.shape[2] != 3 or not src.is_contiguous():
        raise ValueError('resize_area_u8 expects a contiguous uint8 (H, W, 3) image')
    out = torch.empty(dh, dw, 3, dtype=torch.uint8, device=src.device)
    L.check(lib.cf_resize_area_u8(L.ptr(src, dtype=torch.uint8), src.shape[0], src.shape[1], L.ptr(out, dtype=torch.uint8), dh, dw, L.stream_ptr()),
            'cf_resize_area_u8')
    return out


def f32_to_u8_trunc(x):
    lib = L.load()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    L.check(lib.cf_f32_to_u8_trunc(L.ptr(x), x.numel(), L.ptr(out, dtype=torch.uint8), L.stream_ptr()), 'cf_f32_to_u8_trunc')
    return out


def label_lut(labels, lut):
    lib = L.load()
    out = torch.empty(labels.shape, dtype=torch.float32, device=labels.device)
    arr = (ctypes.c_float * len(lut))(*[float(v) for v in lut])
    L.check(lib.cf_label_lut_f32(L.ptr(labels, dtype=torch.int64), labels.numel(), arr, len(lut), L.ptr(out), L.stream_ptr()), 'cf_label_lut_f32')
    return out


def scale_clear_border_(x, border, scale):
    lib = L.load()
    B, H, W = x.shape
    L.check(lib.cf_scale_clear_border_f32(L.ptr(x), B, H, W, int(border), float(scale), L.stream_ptr()), 'cf_scale_clear_border_f32')
    return x
