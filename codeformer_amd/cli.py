"""Shared plumbing of the three aligned-face entrypoints (restoration / inpainting / colorization):
image list -> uint8 batches -> device tensors -> net -> uint8 faces -> PNGs.  On ROCm the uint8<->tensor boundary and the
inpainting composite run as HIP kernels; on CPU the host helpers of codeformer_amd.utils are used."""
import glob
import os

import numpy as np
import torch

from .utils.img_util import img2tensor, imread_bgr, normalize_, tensor2img

IMAGE_EXT = ('jpg', 'jpeg', 'png', 'JPG', 'JPEG', 'PNG')


def list_images(input_path):
    """(image paths, folder name) with the reference's glob (inference_codeformer.py:104-109)."""
    if input_path.endswith(IMAGE_EXT):
        return [input_path], None
    path = input_path[:-1] if input_path.endswith('/') else input_path
    return sorted(glob.glob(os.path.join(path, '*.[jpJP][pnPN]*[gG]'))), os.path.basename(path)


def faces_to_tensor(faces, device):
    """List of uint8 HxWx3 BGR -> (B,3,H,W) fp32 RGB in [-1,1] on `device` (img2tensor(face/255.) + normalize(0.5, 0.5))."""
    if device.type == 'cuda':
        from . import ops
        return ops.img_u8_to_tensor(torch.from_numpy(np.stack(faces)).to(device, non_blocking=True))
    ts = [normalize_(img2tensor(f / 255., bgr2rgb=True, float32=True), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)) for f in faces]
    return torch.stack(ts).to(device)


def tensor_to_faces(t):
    """(B,3,H,W) -> list of uint8 HWC BGR == tensor2img(t_i, rgb2bgr=True, min_max=(-1,1))."""
    if t.is_cuda:
        from . import ops
        return list(ops.tensor_to_img_u8(t).cpu().numpy())
    return [tensor2img(t[i], rgb2bgr=True, min_max=(-1, 1)) for i in range(t.shape[0])]


def inpaint_composite(x, y):
    """(1-mask)*x + mask*y with mask = pure-white input pixels (inference_inpainting.py:68-74)."""
    if x.is_cuda:
        from . import ops
        return ops.mask_composite(x, y)
    mask = (torch.sum(x, dim=1, keepdim=True) == 3).to(x.dtype)
    return (1 - mask) * x + mask * y


def load_faces(paths, size=None, require=None):
    faces = []
    for p in paths:
        img = imread_bgr(p)
        if require is not None:
            assert img.shape[:2] == require, f'Input resolution must be {require[0]}x{require[1]} for this entrypoint.'
        if size is not None and img.shape[:2] != size[::-1]:
            from .utils.img_util import resize_bilinear
            img = resize_bilinear(img, size)
        faces.append(img)
    return faces


def build_codeformer(device, ckpt_name, url, codebook_size, connect_list, random_init_seed=None):
    """ARCH_REGISTRY CodeFormer + checkpoint from weights/CodeFormer (never downloads; optional seeded random init)."""
    from .utils.download_util import load_file_from_url
    from .utils.registry import ARCH_REGISTRY
    from . import archs  # noqa: F401  (registers the arch classes)
    kw = dict(dim_embd=512, codebook_size=codebook_size, n_head=8, n_layers=9, connect_list=list(connect_list))
    try:
        ckpt = load_file_from_url(url=url, model_dir='weights/CodeFormer', progress=True, file_name=None)
    except FileNotFoundError:
        if random_init_seed is None:
            raise
        print(f'WARNING: {ckpt_name} not found -- using torch.manual_seed({random_init_seed}) random weights')
        torch.manual_seed(random_init_seed)
        return ARCH_REGISTRY.get('CodeFormer')(**kw).to(device).eval()
    net = ARCH_REGISTRY.get('CodeFormer')(**kw)
    net.load_state_dict(torch.load(ckpt, map_location='cpu')['params_ema'])
    return net.to(device).eval()
