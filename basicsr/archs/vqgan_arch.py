from codeformer_amd.archs.vqgan_arch import *  # noqa: F401,F403
