from codeformer_amd.utils.registry import *  # noqa: F401,F403
