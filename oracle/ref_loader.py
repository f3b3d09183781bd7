"""Import the reference's own arch files by path (build container only).

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box, so
this module is used solely by oracle/make_golden.py (run here, outputs committed
under tests/golden/) and by CPU tests that skip when the reference is absent.

The full ``basicsr`` package of the reference does not import on PyTorch-ROCm
(SURVEY.md F4/F5), so the three files the hot path needs are loaded under a stub
package exactly as SURVEY.md Appendix F describes.
"""
import importlib.util
import logging
import os
import sys
import types

REF = os.environ.get('CODEFORMER_REFERENCE', '/root/reference')


def available():
    return os.path.isfile(os.path.join(REF, 'basicsr/archs/codeformer_arch.py'))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    """Returns (registry_module, vqgan_arch_module, codeformer_arch_module).

    Installs stub ``basicsr*`` entries in sys.modules -- callers that also want
    this repo's own ``basicsr`` shim must run in a separate process.
    """
    if not available():
        raise RuntimeError(f'reference not found under {REF}')
    saved = {k: v for k, v in sys.modules.items() if k == 'basicsr' or k.startswith('basicsr.')}
    for k in saved:
        del sys.modules[k]
    for pkg in ('basicsr', 'basicsr.utils', 'basicsr.archs'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    sys.modules['basicsr.utils'].get_root_logger = lambda *a, **k: logging.getLogger('basicsr_ref')
    reg = _load('basicsr.utils.registry', f'{REF}/basicsr/utils/registry.py')
    vq = _load('basicsr.archs.vqgan_arch', f'{REF}/basicsr/archs/vqgan_arch.py')
    cf = _load('basicsr.archs.codeformer_arch', f'{REF}/basicsr/archs/codeformer_arch.py')
    # detach the stubs again so that this repo's own basicsr shim stays importable
    ref_mods = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'basicsr' or k.startswith('basicsr.')}
    sys.modules.update(saved)
    return reg, vq, cf, ref_mods


def load_reference_rrdbnet():
    """Returns (rrdbnet_arch_module, realesrgan_utils_module) of the reference.

    arch_util.py imports torchvision and the compiled DCN op at module level, realesrgan_utils.py imports cv2; none of
    them is touched by RRDBNet.forward or by RealESRGANer's tensor stages (pre_process / tile_process / post_process),
    so empty stand-ins are enough to import the files.  `RealESRGANer.enhance` (cv2 colour conversion) is NOT usable.
    """
    if not available():
        raise RuntimeError(f'reference not found under {REF}')
    saved = {k: v for k, v in sys.modules.items()
             if k == 'basicsr' or k.startswith('basicsr.') or k in ('torchvision', 'cv2')}
    for k in saved:
        del sys.modules[k]
    for pkg in ('basicsr', 'basicsr.utils', 'basicsr.archs', 'basicsr.ops', 'basicsr.ops.dcn', 'basicsr.utils.download_util',
                'basicsr.utils.misc', 'torchvision', 'cv2'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    sys.modules['torchvision'].__version__ = '0.0.0'
    sys.modules['basicsr.utils'].get_root_logger = lambda *a, **k: logging.getLogger('basicsr_ref')
    sys.modules['basicsr.ops.dcn'].ModulatedDeformConvPack = type('ModulatedDeformConvPack', (), {})
    sys.modules['basicsr.ops.dcn'].modulated_deform_conv = None
    sys.modules['basicsr.utils.download_util'].load_file_from_url = None
    sys.modules['basicsr.utils.misc'].get_device = lambda gpu_id=None: 'cpu'
    try:
        _load('basicsr.utils.registry', f'{REF}/basicsr/utils/registry.py')
        _load('basicsr.archs.arch_util', f'{REF}/basicsr/archs/arch_util.py')
        arch = _load('basicsr.archs.rrdbnet_arch', f'{REF}/basicsr/archs/rrdbnet_arch.py')
        esr = _load('basicsr.utils.realesrgan_utils', f'{REF}/basicsr/utils/realesrgan_utils.py')
    finally:
        for k in [k for k in sys.modules if k == 'basicsr' or k.startswith('basicsr.') or k in ('torchvision', 'cv2')]:
            del sys.modules[k]
        sys.modules.update(saved)
    return arch, esr


def load_reference_parsenet():
    """The reference's facelib/parsing/parsenet.py (imports only numpy + torch) as a stand-alone module."""
    if not available():
        raise RuntimeError(f'reference not found under {REF}')
    spec = importlib.util.spec_from_file_location('_ref_parsenet', f'{REF}/facelib/parsing/parsenet.py')
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_reference_img_util():
    """The reference's basicsr/utils/img_util.py (img2tensor / tensor2img, the tensor boundary of
    inference_codeformer.py:199-206) with stand-ins for the two imports this container lacks: `cv2.cvtColor` is only
    used for the BGR<->RGB swap (a channel reversal, exact for every dtype) and `torchvision.utils.make_grid` only for 4-D
    batches (never reached: the callers pass one face)."""
    if not available():
        raise RuntimeError(f'reference not found under {REF}')
    import numpy as np
    saved = {k: sys.modules.get(k) for k in ('cv2', 'torchvision', 'torchvision.utils')}
    cv2 = types.ModuleType('cv2')
    cv2.COLOR_BGR2RGB, cv2.COLOR_RGB2BGR = 4, 4
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])
    tv, tvu = types.ModuleType('torchvision'), types.ModuleType('torchvision.utils')
    tvu.make_grid = None
    tv.utils = tvu
    sys.modules.update({'cv2': cv2, 'torchvision': tv, 'torchvision.utils': tvu})
    try:
        spec = importlib.util.spec_from_file_location('_ref_img_util', f'{REF}/basicsr/utils/img_util.py')
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return m


def load_reference_retinaface():
    """The reference's facelib/detection/retinaface/{retinaface_net,retinaface_utils,retinaface}.py (last clause)
    with stand-ins for what this container lacks: torchvision (oracle/tv_stub.py: ResNet-50, IntermediateLayerGetter, nms -- restated
    from torchvision's published definitions), cv2 (never reached by `detect_faces(ndarray, use_origin_size=True)` /
    `batched_detect_faces(float tensor)`), basicsr.utils.misc.get_device -> cpu.  facelib.detection.align_trans (+ matlab_cp2tform)
    are the reference's own files.  Returns (retinaface_module, net_module, utils_module)."""
    if not available():
        raise RuntimeError(f'reference not found under {REF}')
    from . import tv_stub
    keys = ('facelib', 'basicsr', 'torchvision', 'cv2')
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in keys}
    for k in saved:
        del sys.modules[k]
    for pkg in ('facelib', 'facelib.detection', 'facelib.detection.retinaface', 'basicsr', 'basicsr.utils', 'basicsr.utils.misc', 'cv2'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    sys.modules['basicsr.utils.misc'].get_device = lambda gpu_id=None: 'cpu'
    sys.modules.update(tv_stub.modules())
    try:
        base = f'{REF}/facelib/detection'
        _load('facelib.detection.matlab_cp2tform', f'{base}/matlab_cp2tform.py')
        _load('facelib.detection.align_trans', f'{base}/align_trans.py')
        net = _load('facelib.detection.retinaface.retinaface_net', f'{base}/retinaface/retinaface_net.py')
        utils = _load('facelib.detection.retinaface.retinaface_utils', f'{base}/retinaface/retinaface_utils.py')
        rf = _load('facelib.detection.retinaface.retinaface', f'{base}/retinaface/retinaface.py')
    finally:
        for k in [k for k in sys.modules if k.split('.')[0] in keys]:
            del sys.modules[k]
        sys.modules.update(saved)
    return rf, net, utils


class torchvision_stub:
    """Context manager: oracle/tv_stub.py visible as `torchvision` (the reference's RetinaFace imports torchvision.models inside its
    constructor, retinaface.py:96)."""

    def __enter__(self):
        from . import tv_stub
        self.saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] == 'torchvision'}
        for k in self.saved:
            del sys.modules[k]
        sys.modules.update(tv_stub.modules())
        return self

    def __exit__(self, *exc):
        for k in [k for k in sys.modules if k.split('.')[0] == 'torchvision']:
            del sys.modules[k]
        sys.modules.update(self.saved)
        return False
