from codeformer_amd.utils.img_util import *  # noqa: F401,F403
