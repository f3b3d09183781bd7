"""Drop-in shim: the reference's entrypoints import `basicsr.*`; every name resolves to codeformer_amd.

Only the modules on the aligned-face inference path exist (archs, utils.{registry,misc,img_util,logger,
download_util}); training-side packages of the reference (data, losses, metrics, models, ops, train) are out of scope.
"""
from codeformer_amd import __version__  # noqa: F401
