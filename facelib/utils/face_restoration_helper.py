from codeformer_amd.facelib.utils.face_restoration_helper import FaceRestoreHelper, get_center_face, get_largest_face  # noqa: F401
