from codeformer_amd.facelib.detection.retinaface.retinaface import RetinaFace, generate_config  # noqa: F401
