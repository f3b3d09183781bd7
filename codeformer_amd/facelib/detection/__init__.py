"""Face detection models (reference: facelib/detection/__init__.py:14-57).

`init_detection_model(name, half, device)` with the reference's names.  Built: 'retinaface_resnet50' (the default of
inference_codeformer.py) and 'retinaface_mobile0.25'.  Not built: the YOLOv5-face detectors and dlib (optional alternatives of the
reference; they raise NotImplementedError here).  Weights are read from weights/facelib/ (no download: load_file_from_url returns an
existing file or raises FileNotFoundError).
"""
import os

import torch

from ...utils.download_util import load_file_from_url
from .retinaface.retinaface import RetinaFace

__all__ = ['RetinaFace', 'init_detection_model', 'init_retinaface_model']

_URLS = {
    'retinaface_resnet50': ('resnet50', 'https://github.com/sczhou/CodeFormer/releases/download/v0.1.0/detection_Resnet50_Final.pth'),
    'retinaface_mobile0.25': ('mobile0.25',
                              'https://github.com/sczhou/CodeFormer/releases/download/v0.1.0/detection_mobilenet0.25_Final.pth'),
}


def init_detection_model(model_name, half=False, device='cpu', model_path=None):
    if 'retinaface' in model_name:
        return init_retinaface_model(model_name, half, device, model_path)
    raise NotImplementedError(f'{model_name} is not implemented (built: {sorted(_URLS)}).')


def init_retinaface_model(model_name, half=False, device='cpu', model_path=None):
    if model_name not in _URLS:
        raise NotImplementedError(f'{model_name} is not implemented.')
    network, url = _URLS[model_name]
    model = RetinaFace(network_name=network, half=half, device='cpu')
    if model_path is None:
        model_path = load_file_from_url(url=url, model_dir=os.path.join('weights', 'facelib'), progress=True, file_name=None)
    sd = torch.load(model_path, map_location='cpu')
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}      # checkpoints saved from DataParallel
    model.load_state_dict(sd, strict=True)
    model = model.eval().to(device)
    return model.half() if half else model
