"""facelib.utils of the reference (facelib/utils/__init__.py): the restoration helper and the gray-image helpers."""
from ...utils.face_misc import adain_npy, bgr2gray, is_gray  # noqa: F401
from .face_restoration_helper import FaceRestoreHelper, get_center_face, get_largest_face  # noqa: F401
