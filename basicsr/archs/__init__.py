from codeformer_amd.archs import ARCH_REGISTRY, build_network  # noqa: F401
