"""Drop-in shim: the reference's entrypoints import `basicsr.*`; every name resolves to codeformer_amd.

Only the modules on the aligned-face inference path exist (archs, utils.{registry,misc,img_util,logger,
download_util}); training-side packages of the reference (data, losses, metrics, models, ops, train) are out of scope.
"""
from codeformer_amd import __version__  # noqa: F401
from . import archs  # noqa: F401,E402  (the reference's basicsr/__init__.py:3 star-imports archs, which fills ARCH_REGISTRY)
from . import utils  # noqa: F401,E402
