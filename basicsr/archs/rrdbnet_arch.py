from codeformer_amd.archs.rrdbnet_arch import *  # noqa: F401,F403
from codeformer_amd.archs.rrdbnet_arch import RRDB, ResidualDenseBlock, RRDBNet  # noqa: F401
