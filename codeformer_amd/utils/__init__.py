from .img_util import img2tensor, imread_bgr, imwrite, normalize_, tensor2img
from .logger import get_root_logger
from .misc import get_device, get_time_str, gpu_is_available, scandir, set_random_seed, sizeof_fmt
from .registry import ARCH_REGISTRY

__all__ = ['img2tensor', 'tensor2img', 'imwrite', 'imread_bgr', 'normalize_', 'get_root_logger', 'get_device',
           'gpu_is_available', 'scandir', 'set_random_seed', 'get_time_str', 'sizeof_fmt', 'ARCH_REGISTRY']
