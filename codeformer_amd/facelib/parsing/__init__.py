"""Face parsing (reference: facelib/parsing/__init__.py:8-23)."""
import os

import torch

from ...utils.download_util import load_file_from_url
from .parsenet import ParseNet

__all__ = ['ParseNet', 'init_parsing_model']

_PARSENET_URL = 'https://github.com/sczhou/CodeFormer/releases/download/v0.1.0/parsing_parsenet.pth'


def init_parsing_model(model_name='bisenet', half=False, device='cuda', model_path=None):
    """Same call as the reference.  Only 'parsenet' (the model face_restoration_helper.py:125 asks for) exists here;
    weights are read from weights/facelib/ (no download: load_file_from_url returns an existing file or raises)."""
    if model_name != 'parsenet':
        raise NotImplementedError(f'{model_name} is not implemented (HIP path: parsenet only).')
    model = ParseNet(in_size=512, out_size=512, parsing_ch=19)
    if model_path is None:
        model_path = load_file_from_url(url=_PARSENET_URL, model_dir=os.path.join('weights', 'facelib'), progress=True,
                                        file_name=None)
    model.load_state_dict(torch.load(model_path, map_location='cpu'), strict=True)
    return model.eval().to(device)
