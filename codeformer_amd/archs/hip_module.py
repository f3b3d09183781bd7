"""Base class shared by the HIP-backed arch modules.

Public contract (same as the reference's nn.Modules): `forward(x)` takes / returns NCHW tensors.
  * CUDA (ROCm) tensors -> `forward_nhwc`, i.e. the HIP kernels behind the C ABI (channels-last inside);
  * CPU tensors         -> `forward_host`, stock torch ops, kept for the reference's CPU plumbing config
                           (inference_codeformer.py on a box without a GPU).  It is selected ONLY by the
                           tensor's device; a CUDA tensor never reaches it, and a missing native library raises.
"""
import torch
from torch import nn

from .. import ops


# Bumped whenever any packed weight is (re)built: captured HIP graphs hold raw pointers to the packed buffers, so a graph
# recorded under an older epoch must be re-captured.
PACK_EPOCH = [0]


class HipModule(nn.Module):

    def _packed(self, key, build, *params):
        """Cache a packed weight until one of its source parameters is modified, replaced or moved."""
        sig = tuple((p.data_ptr(), ops.tensor_version(p), str(p.device)) for p in params if p is not None)
        cache = self.__dict__.setdefault('_hip_cache', {})
        ent = cache.get(key)
        if ent is None or ent[0] != sig:
            ent = (sig, build())
            cache[key] = ent
            PACK_EPOCH[0] += 1
        return ent[1]

    def _pw_conv(self, name, bf16=False, up2x=False, f16=False, hw=None, c_split=None):
        """Packed weight of a conv / linear sub-module for operand code `bf16` (0 fp32, 1 bf16, 2 IEEE half, ops.WINOGRAD,
        ops.SPLIT).  The Winograd and split-half kernels need the conv's INPUT size `hw` (and the concat boundary `c_split`);
        shapes they do not cover fall back as ops.conv_code says."""
        conv = getattr(self, name) if isinstance(name, str) else name
        code = 2 if f16 else int(bf16)
        if code in (ops.WINOGRAD, ops.WINOGRAD_F43) + ops.SPLIT_CODES or (code in (1, 2) and hw is not None and not up2x):
            cout, cin = conv.weight.shape[:2]
            if hw is None or tuple(conv.weight.shape[2:]) != (3, 3):
                code = 0
            else:
                code = ops.conv_code(code, cin, cout, hw[0], hw[1], up2x=up2x, c_split=c_split)
        key = (name if isinstance(name, str) else id(conv), code, bool(up2x))
        return self._packed(key, lambda: ops.pack_weight(conv.weight, conv.bias, bf16=code, up2x=up2x), conv.weight, conv.bias)

    def _range_code(self, norm, code, n):
        """Operand code for a 3x3 conv whose input passes GroupNorm `norm` with n elements per group: `code`, or its exact-fp32
        replacement when |gamma| * sqrt(n - 1) + |beta| could leave the IEEE-half operand range (ops.gn_range_ok).  The two maxima are
        read back once per parameter version."""
        code = int(code)
        if code in (0, 1, ops.WINOGRAD, ops.WINOGRAD_F43):
            return code
        gmax, bmax = self._packed(('gn_range', id(norm)),
                                  lambda: (float(norm.weight.detach().abs().max()), float(norm.bias.detach().abs().max())),
                                  norm.weight, norm.bias)
        return code if ops.gn_range_ok(gmax, bmax, n) else ops.exact_code(code)

    def invalidate_packed_weights(self):
        """Drop every cached packed weight of this module tree.  The cache already follows load_state_dict / .to() /
        optimizer steps (parameter version + storage pointer); call this after editing `param.data` in place, which
        PyTorch does not version."""
        for m in self.modules():
            if isinstance(m, HipModule):
                m.__dict__.pop('_hip_cache', None)
        PACK_EPOCH[0] += 1

    def forward_nhwc(self, x):  # pragma: no cover - abstract
        raise NotImplementedError

    def forward_host(self, x):  # pragma: no cover - abstract
        raise NotImplementedError

    def forward(self, x):
        if x.is_cuda:
            ops.L.ensure_device(x.device)   # kernel attributes on the tensor's device, outside any capture
            with torch.no_grad(), torch.cuda.device(x.device):
                return ops.to_nchw(self.forward_nhwc(ops.to_nhwc(x.float())))
        return self.forward_host(x)
