from codeformer_amd.facelib.utils import *  # noqa: F401,F403
