"""codeformer_amd -- MI355X-native CodeFormer aligned-face inference path.

Layout
  csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/codeformer_hip.h)
  lib.py     ctypes binding (no fallback: missing library == error)
  ops.py     operator layer on channels-last fp32 tensors
  archs/     nn.Module mirrors of basicsr/archs/{vqgan_arch,codeformer_arch}.py (same names / state_dict keys)
  utils/     registry, device pick, image<->tensor helpers of the reference's basicsr/utils used by the path
  parallel.py  batch sharding over ranks + the single RCCL gather
"""
__version__ = '0.1.0'
