from codeformer_amd.facelib.detection.retinaface.retinaface_net import *  # noqa: F401,F403
