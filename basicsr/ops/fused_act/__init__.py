from codeformer_amd.bundled import FusedLeakyReLU, fused_leaky_relu

__all__ = ['FusedLeakyReLU', 'fused_leaky_relu']
