from codeformer_amd.utils import *  # noqa: F401,F403
from codeformer_amd.utils import get_root_logger, img2tensor, imwrite, scandir, tensor2img  # noqa: F401
