from codeformer_amd.bundled import upfirdn2d

__all__ = ['upfirdn2d']
