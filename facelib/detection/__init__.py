from codeformer_amd.facelib.detection import RetinaFace, init_detection_model, init_retinaface_model  # noqa: F401
