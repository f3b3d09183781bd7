from codeformer_amd.utils.download_util import *  # noqa: F401,F403
