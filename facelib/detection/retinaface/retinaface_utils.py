from codeformer_amd.facelib.detection.retinaface.retinaface_utils import *  # noqa: F401,F403
from codeformer_amd.facelib.detection.retinaface.retinaface_utils import PriorBox, py_cpu_nms  # noqa: F401
