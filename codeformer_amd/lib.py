"""ctypes binding of libcodeformer_hip.so (the C ABI declared in include/codeformer_hip.h).

This is the only place where Python touches the native library.  There is NO fallback: if the shared object is
missing or fails to load, every op raises -- a GPU run can never silently take another path.
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# CF_LIB_PATH overrides the in-tree library (used only by tools/ablate.sh timing experiments)
LIB_PATH = os.environ.get('CF_LIB_PATH') or os.path.join(_PKG, 'libcodeformer_hip.so')

c_float_p = ctypes.c_void_p  # device pointers are passed as integers

ABI_VERSION = 22
PRO_NONE, PRO_AFFINE, PRO_AFFINE_SWISH, PRO_LEAKY = 0, 1, 2, 3
EPI_NONE, EPI_RESIDUAL, EPI_SFT, EPI_GELU, EPI_LEAKY, EPI_AXPY, EPI_AXPY2 = 0, 1, 2, 3, 4, 5, 6
PAD_ZERO, PAD_REFLECT, PAD_EDGE = 0, 1, 2


class ConvDesc(ctypes.Structure):
    """Mirror of struct cf_conv_desc (include/codeformer_hip.h)."""
    _fields_ = [
        ('in0', ctypes.c_void_p), ('in1', ctypes.c_void_p),
        ('c0', ctypes.c_int32), ('c1', ctypes.c_int32),
        ('batch', ctypes.c_int32), ('hin', ctypes.c_int32), ('win', ctypes.c_int32),
        ('hout', ctypes.c_int32), ('wout', ctypes.c_int32),
        ('cout', ctypes.c_int32), ('cout_pad', ctypes.c_int32),
        ('taps', ctypes.c_int32), ('stride', ctypes.c_int32), ('upsample', ctypes.c_int32),
        ('in_nchw', ctypes.c_int32), ('out_nchw', ctypes.c_int32),
        ('prologue', ctypes.c_int32), ('epilogue', ctypes.c_int32),
        ('pro_scale', ctypes.c_void_p), ('pro_shift', ctypes.c_void_p),
        ('weight', ctypes.c_void_p), ('bias', ctypes.c_void_p),
        ('res', ctypes.c_void_p), ('sft_scale', ctypes.c_void_p),
        ('sft_w', ctypes.c_float),
        ('out', ctypes.c_void_p),
        ('stats_out', ctypes.c_void_p), ('stats_cpg', ctypes.c_int32), ('bf16_mfma', ctypes.c_int32),
        ('ld_in0', ctypes.c_int32), ('ld_in1', ctypes.c_int32), ('ld_out', ctypes.c_int32),
        ('pad_mode', ctypes.c_int32), ('pad_lo', ctypes.c_int32), ('winograd', ctypes.c_int32),
        ('acc_scale', ctypes.c_float),
        ('split_k', ctypes.c_int32), ('workspace', ctypes.c_void_p), ('counters', ctypes.c_void_p),
        ('act_scale', ctypes.c_void_p),
        ('io_bf16', ctypes.c_int32),
        ('in0_alt', ctypes.c_void_p), ('alt_cout0', ctypes.c_int32),
    ]


# name -> (restype, argtypes); every symbol of include/codeformer_hip.h
_I, _P, _F, _L = ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_int64
SIGNATURES = {
    'cf_version': (_I, []),
    'cf_last_error': (ctypes.c_char_p, []),
    'cf_build_id': (ctypes.c_char_p, []),
    'cf_device_cu_count': (_I, []),
    'cf_device_init': (_I, []),
    'cf_conv2d': (_I, [ctypes.POINTER(ConvDesc), _P]),
    'cf_conv2d_stats_parts': (_I, [ctypes.POINTER(ConvDesc)]),
    'cf_conv2d_workspace_bytes': (_L, [ctypes.POINTER(ConvDesc)]),
    'cf_conv2d_tiles': (_I, [ctypes.POINTER(ConvDesc)]),
    'cf_pack_conv_weight': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'cf_packed_weight_elems': (_L, [_I, _I, _I]),
    'cf_pack_conv_weight_bf16': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'cf_pack_conv_weight_f16': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'cf_pack_conv_weight_up2x': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'cf_pack_conv_weight_winograd': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'cf_pack_conv_weight_winograd_f16x2': (_I, [_P, _I, _I, _I, _I, _F, _P, _P]),
    'cf_pack_conv_weight_winograd_bf16': (_I, [_P, _I, _I, _I, _I, _F, _P, _P]),
    'cf_pack_conv_weight_winograd43_f16x2': (_I, [_P, _I, _I, _I, _I, _F, _P, _P]),
    'cf_pack_conv_weight_winograd43': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'cf_pack_linear_weight_f16x2': (_I, [_P, _I, _I, _F, _P, _P]),
    'cf_pack_conv_weight_up2x_bf16': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'cf_pack_conv_weight_up2x_f16': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'cf_pack_conv_weight_f16x2': (_I, [_P, _I, _I, _I, _I, _I, _F, _P, _P]),
    'cf_groupnorm_stats': (_I, [_P, _I, _I, _I, _I, _P, _I, _P]),
    'cf_act_scale_from_stats': (_I, [_P, _I, _I, _F, _P, _P, _P]),
    'cf_act_scale_from_tensor': (_I, [_P, _I, _L, _F, _P, _P, _P]),
    'cf_act_scale_fused': (_I, [_P, _I, _P, _I, _P, _L, _I, _F, _P, _P, _P]),
    'cf_groupnorm_finalize': (_I, [_P, _I, _I, _I, _I, _I, _L, _P, _P, _F, _P, _P, _I, _P]),
    'cf_groupnorm_finalize2': (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _L, _P, _P, _F, _P, _P, _I, _F, _P, _P, _P]),
    'cf_layernorm': (_I, [_P, _I, _I, _P, _P, _F, _P, _I, _P, _P, _P]),
    'cf_attention': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    'cf_argmax_rows': (_I, [_P, _I, _I, _P, _P]),
    'cf_codebook_gather_adain': (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _F, _P, _P]),
    'cf_row_sqnorm': (_I, [_P, _I, _I, _P, _P]),
    'cf_vq_argmin': (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    'cf_nchw_to_nhwc': (_I, [_P, _I, _I, _I, _P, _P]),
    'cf_pixel_unshuffle_nhwc': (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'cf_nhwc_to_nchw': (_I, [_P, _I, _I, _I, _P, _P]),
    'cf_img_u8_to_tensor': (_I, [_P, _I, _I, _I, _P, _P]),
    'cf_tensor_to_img_u8': (_I, [_P, _I, _I, _I, _P, _P]),
    'cf_mask_composite': (_I, [_P, _P, _I, _I, _I, _P, _P]),
    'cf_f32_to_bf16': (_I, [_P, _L, _P, _P]),
    'cf_fused_bias_act': (_I, [_P, _P, _L, _I, _I, _F, _F, _P, _P]),
    'cf_fused_bias_act_ex': (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _F, _F, _I, _P, _P]),
    'cf_upfirdn2d': (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'cf_warp_affine_u8': (_I, [_P, _L, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'cf_warp_affine_f32': (_I, [_P, _I, _I, ctypes.POINTER(ctypes.c_double), _P, _I, _I, _I, _I, _P]),
    'cf_erode_f32': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'cf_gaussian_blur_f32': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    'cf_sum_f32': (_I, [_P, _L, _P, _P]),
    'cf_paste_blend': (_I, [_P, _I, _I, _P, _I, _I, ctypes.POINTER(ctypes.c_double), _P, _P, _P, _I, _I, _I, _I, _P]),
    'cf_resize_linear_u8': (_I, [_P, _I, _I, _P, _I, _I, _P]),
    'cf_resize_area_u8': (_I, [_P, _I, _I, _P, _I, _I, _P]),
    'cf_f32_to_u8_trunc': (_I, [_P, _L, _P, _P]),
    'cf_label_lut_f32': (_I, [_P, _L, ctypes.POINTER(ctypes.c_float), _I, _P, _P]),
    'cf_scale_clear_border_f32': (_I, [_P, _I, _I, _I, _I, _F, _P]),
}

_lib = None
_cuda = None      # torch.cuda.is_available(), asked once
_devices = set()   # devices cf_device_init() has run on (the library itself keeps no such state)


class NativeLibraryError(RuntimeError):
    pass


def load():
    """The shared object (loaded once), with cf_device_init() done on the current CUDA (ROCm) device.  Raises NativeLibraryError if
    the file is absent -- never falls back."""
    global _cuda
    lib = _lib if _lib is not None else _open()
    if _cuda is None:
        _cuda = torch.cuda.is_available()
    if _cuda:
        dev = torch.cuda.current_device()
        if dev not in _devices:   # per-device kernel attributes (dynamic LDS above 64 KB), once per device and process
            _init_device(lib, dev)
    return lib


def _init_device(lib, dev):
    if torch.cuda.is_current_stream_capturing():
        # hipFuncSetAttribute must not be issued inside a stream capture: the arch modules call ensure_device() before they capture;
        # a user capture that is the FIRST use of a device has to do the same
        raise NativeLibraryError(f'cf_device_init has not run on cuda:{dev} yet and the current stream is capturing: call '
                                 'codeformer_amd.lib.ensure_device(device) (or run one forward) before the capture')
    if lib.cf_device_init() != 0:
        raise NativeLibraryError(f'cf_device_init failed on cuda:{dev} ({torch.cuda.get_device_name(dev)}): '
                                 f'{lib.cf_last_error().decode("utf-8", "replace")}')
    _devices.add(dev)


def ensure_device(device):
    """Run cf_device_init() for `device` (torch.device / str / index) under that device's context.  The arch modules call this with the
    INPUT tensor's device at the top of every forward, so the per-device kernel attributes are set on the device the kernels will run
    on (not on whatever device happens to be current) and never inside a stream capture."""
    global _cuda
    lib = _lib if _lib is not None else _open()
    if _cuda is None:
        _cuda = torch.cuda.is_available()
    if not _cuda:
        return lib
    d = torch.device(device) if not isinstance(device, int) else torch.device('cuda', device)
    idx = torch.cuda.current_device() if d.index is None else d.index
    if idx not in _devices:
        with torch.cuda.device(idx):
            _init_device(lib, idx)
    return lib


def _open():
    global _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f'{LIB_PATH} not found: build it with `python -m codeformer_amd.build` (hipcc --offload-arch=gfx950). '
            'codeformer_amd has no non-HIP execution path for GPU tensors.')
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.cf_version() != ABI_VERSION:
        raise NativeLibraryError(f'ABI version mismatch: library {lib.cf_version()}, binding {ABI_VERSION}')
    _lib = lib
    return lib


def is_available():
    return os.path.exists(LIB_PATH)


def last_error():
    lib = _lib if _lib is not None else _open()   # (not load(): a failing cf_device_init reports through here)
    return lib.cf_last_error().decode('utf-8', 'replace')


def check(status, what):
    if status != 0:
        raise RuntimeError(f'{what} failed ({status}): {last_error()}')


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t, strided=False, dtype=torch.float32):
    """Device pointer of a tensor (None -> NULL).  The tensor must be a contiguous CUDA tensor of the dtype the kernel reads the
    bytes as -- float32 unless the call site names another one (int64 code indices, float64 statistics, uint8 images;
    dtype=None for opaque packed-weight buffers): a mismatch raises instead of being reinterpreted.
    strided=True: the caller validated the layout itself (e.g. a channel slice of an NHWC buffer)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError('codeformer_amd ops need CUDA (ROCm) tensors')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'codeformer_amd op expected a {dtype} tensor, got {t.dtype}')
    if not strided and not t.is_contiguous():
        raise ValueError('codeformer_amd ops need contiguous tensors')
    return t.data_ptr()
