from codeformer_amd.utils.logger import *  # noqa: F401,F403
