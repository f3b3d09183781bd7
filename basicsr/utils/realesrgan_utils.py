from codeformer_amd.utils.realesrgan_utils import RealESRGANer  # noqa: F401
