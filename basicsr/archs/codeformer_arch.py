from codeformer_amd.archs.codeformer_arch import *  # noqa: F401,F403
