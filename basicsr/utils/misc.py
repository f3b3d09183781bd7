from codeformer_amd.utils.misc import *  # noqa: F401,F403
