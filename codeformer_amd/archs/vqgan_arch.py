"""VQGAN building blocks of CodeFormer, MI355X-native.

API mirror of the reference's basicsr/archs/vqgan_arch.py (same class names, constructor arguments, parameter
names/shapes and therefore state_dict keys; same default initialisation order, so `torch.manual_seed(s)` gives
bit-identical random weights).  The arithmetic is NOT torch's: on ROCm tensors every block runs hand-written HIP
kernels (codeformer_amd/csrc) on channels-last activations, with GroupNorm-apply, swish, nearest-upsample,
zero-padding, residual adds and biases fused into the implicit-GEMM convolution.

Reference lines each block follows are cited per class (paths relative to the reference repo).
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..ops import EPI_RESIDUAL, PRO_AFFINE, PRO_AFFINE_SWISH
from ..utils.logger import get_root_logger
from ..utils.registry import ARCH_REGISTRY
from .hip_module import HipModule

GN_GROUPS = 32
GN_EPS = 1e-6


def normalize(in_channels):
    """GroupNorm(32, C, eps=1e-6, affine) -- parameter container (basicsr/archs/vqgan_arch.py:14-15)."""
    return nn.GroupNorm(num_groups=GN_GROUPS, num_channels=in_channels, eps=GN_EPS, affine=True)


def swish(x):
    """x * sigmoid(x) (basicsr/archs/vqgan_arch.py:18-20); host path only -- fused into the conv gather on GPU."""
    return x * torch.sigmoid(x)


def _gn_tables(norm, *xs, act_growth=None):
    return ops.groupnorm_tables(list(xs), norm.weight, norm.bias, norm.eps, norm.num_groups, act_growth=act_growth)


class VectorQuantizer(HipModule):
    """Nearest-code quantiser (basicsr/archs/vqgan_arch.py:24-84).

    forward(z): L2-nearest code per spatial position (:33-70); get_codebook_feat(indices, shape): row gather (:72-84).
    GPU: z.E^T on the fp32-MFMA GEMM, argmin by wavefront shuffles (cf_vq_argmin), gather by cf_codebook_gather_adain.
    """

    def __init__(self, codebook_size, emb_dim, beta):
        super().__init__()
        self.codebook_size = codebook_size
        self.emb_dim = emb_dim
        self.beta = beta
        self.embedding = nn.Embedding(self.codebook_size, self.emb_dim)
        self.embedding.weight.data.uniform_(-1.0 / self.codebook_size, 1.0 / self.codebook_size)

    def _stats(self, z_q, z, idx, mean_distance):
        enc = torch.zeros(idx.shape[0], self.codebook_size, dtype=z.dtype, device=z.device)
        enc.scatter_(1, idx.view(-1, 1), 1)
        loss = torch.mean((z_q - z) ** 2) * (1.0 + self.beta)
        e_mean = enc.mean(dim=0)
        perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
        return loss, {'perplexity': perplexity, 'min_encodings': enc, 'min_encoding_indices': idx.view(-1, 1),
                      'mean_distance': mean_distance}

    def forward(self, z):
        if not z.is_cuda:
            return self.forward_host(z)
        with torch.no_grad():
            B, C, H, W = z.shape
            zt = ops.to_nhwc(z.float()).view(B * H * W, C)
            cb = self.embedding.weight
            pw = self._packed('codebook', lambda: ops.pack_weight(cb), cb)
            idx, _, (scores, zz, ee) = ops.vq_nearest(zt, cb, pw)
            zq_t = ops.codebook_gather(idx, cb, B, H * W)
            z_q = ops.to_nchw(zq_t.view(B, H, W, C))
            # mean over ALL code distances (a diagnostic, :42): mean(zz) + mean(ee) - 2 mean(z.E^T)
            loss, stats = self._stats(z_q, z, idx, zz.mean() + ee.mean() - 2.0 * scores.mean())
            return z_q, loss, stats

    def forward_host(self, z):
        zp = z.permute(0, 2, 3, 1).contiguous()
        zf = zp.view(-1, self.emb_dim)
        w = self.embedding.weight
        d = (zf ** 2).sum(dim=1, keepdim=True) + (w ** 2).sum(1) - 2 * torch.matmul(zf, w.t())
        idx = torch.argmin(d, dim=1)
        z_q = w[idx].view(zp.shape)
        z_q = (zp + (z_q - zp).detach()).permute(0, 3, 1, 2).contiguous()
        loss, stats = self._stats(z_q, z, idx, torch.mean(d))
        return z_q, loss, stats

    def get_codebook_feat(self, indices, shape):
        """indices: (B*T,) or (B,T,1) int64; shape: [B, H, W, C] or None (-> (B*T, C))."""
        if not indices.is_cuda:
            z_q = self.embedding.weight[indices.view(-1)]
            if shape is not None:
                z_q = z_q.view(shape).permute(0, 3, 1, 2).contiguous()
            return z_q
        with torch.no_grad():
            idx = indices.reshape(-1).contiguous()
            if shape is None:
                return ops.codebook_gather(idx, self.embedding.weight, 1, idx.numel()).view(idx.numel(), -1)
            B, H, W, C = shape
            return ops.to_nchw(ops.codebook_gather(idx, self.embedding.weight, B, H * W).view(B, H, W, C))


class Downsample(HipModule):
    """pad(0,1,0,1) + 3x3 stride-2 conv (basicsr/archs/vqgan_arch.py:117-126); the pad is a bounds check on GPU."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward_nhwc(self, x, bf16=False):
        # split-half operands: the stride-2 form of cf_split.hip (a 2x2 convolution of the space-to-depth view of x); the input is the
        # un-normalised residual stream, so it carries a range scale (computed only when the table is used: CODEFORMER_HIP_RANGE_SCALE=0
        # drops it).  Mode 2 (single IEEE-half operands) takes the same split-half stride-2 form; bf16 (1) keeps the exact kernel.
        cin, cout = self.conv.in_channels, self.conv.out_channels
        if int(bf16) in (2,) + ops.SPLIT_CODES and ops.split_s2_ok(cin, cout, x.shape[1], x.shape[2]):
            pw = self._packed(('conv', 's2'), lambda: ops.pack_weight(self.conv.weight, self.conv.bias, bf16=ops.SPLIT, stride2=True),
                              self.conv.weight, self.conv.bias)
            return ops.conv2d(x, pw, stride=2, emit_stats=True, act=ops.act_scale(x) if ops.needs_act_scale(pw) else None)
        return ops.conv2d(x, self._pw_conv('conv'), stride=2, emit_stats=True)

    def forward_host(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode='constant', value=0))


class Upsample(HipModule):
    """nearest x2 + 3x3 conv (basicsr/archs/vqgan_arch.py:129-138).  GPU: the upsampled tensor never exists -- every output
    parity class (oy&1, ox&1) is a 2x2 convolution of the SOURCE with taps pre-summed at pack time (2.25x fewer MACs)."""

    def __init__(self, in_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward_nhwc(self, x, bf16=False):
        if int(bf16) == 2:
            bf16 = ops.SPLIT   # the input is the un-normalised residual stream: single IEEE halves have no range scaling on the direct kernel
        c = self.conv.in_channels
        if int(bf16) == ops.WINOGRAD_F43 and x.dtype == torch.float32 and ops.f43_up_ok(c, self.conv.out_channels, 2 * x.shape[1], 2 * x.shape[2]):
            # precision 'fp32': Winograd F(4x4,3x3) on the virtually upsampled image (the gather reads source pixel (y >> 1, x >> 1)):
            # 2.25 fp32 products per output instead of the 4 of the folded sub-pixel form
            pw = self._packed(('conv', 'f43up'), lambda: ops.pack_weight(self.conv.weight, self.conv.bias, bf16=ops.WF43F), self.conv.weight, self.conv.bias)
            return ops.conv2d(x, pw, upsample=True, emit_stats=True)
        pw = self._pw_conv('conv', bf16, up2x=True, hw=x.shape[1:3])
        return ops.conv2d(x, pw, upsample=True, emit_stats=True, act=ops.act_scale(x) if ops.needs_act_scale(pw) else None)

    def forward_host(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class ResBlock(HipModule):
    """GN-swish-conv3x3-GN-swish-conv3x3 (+1x1 skip) + residual (basicsr/archs/vqgan_arch.py:141-164).

    GPU: two stats passes + two fused convs (+ one 1x1 GEMM); `forward_nhwc(x, x2)` treats (x, x2) as a channel
    concatenation without materialising it (used by Fuse_sft_block, codeformer_arch.py:152).
    """

    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = normalize(in_channels)
        oc = self.out_channels
        self.conv1 = nn.Conv2d(in_channels, oc, kernel_size=3, stride=1, padding=1)
        self.norm2 = normalize(oc)
        self.conv2 = nn.Conv2d(oc, oc, kernel_size=3, stride=1, padding=1)
        if self.in_channels != oc:
            self.conv_out = nn.Conv2d(in_channels, oc, kernel_size=1, stride=1, padding=0)

    def forward_nhwc(self, x, x2=None, bf16=False):
        """bf16: operand code of the two 3x3 convs (fp32 accumulate / storage); the 1x1 skip follows it only for split-half codes on images
        (_skip_nhwc), otherwise it stays on the fp32 GEMM."""
        xs = (x,) if x2 is None else (x, x2)
        # (a channel-changing block: the skip convolution's range-scale table of the same input rides in the finalize launch)
        sc, sh = _gn_tables(self.norm1, *xs, act_growth=4.0 if self._skip_streams(x, x2, bf16) else None)
        hw = x.shape[1:3]
        npix = hw[0] * hw[1]
        code1 = self._range_code(self.norm1, bf16, npix * self.in_channels // GN_GROUPS)
        code2 = self._range_code(self.norm2, bf16, npix * self.out_channels // GN_GROUPS)
        h = ops.conv2d(x, self._pw_conv('conv1', code1, hw=hw, c_split=None if x2 is None else x.shape[3]), x2=x2, prologue=PRO_AFFINE_SWISH, scale=sc, shift=sh, emit_stats=True)
        sc, sh = _gn_tables(self.norm2, h)
        if self.in_channels != self.out_channels:
            skip = self._skip_nhwc(x, x2, bf16)
        else:
            skip = x
        return ops.conv2d(h, self._pw_conv('conv2', code2, hw=hw), prologue=PRO_AFFINE_SWISH, scale=sc, shift=sh,
                          epilogue=EPI_RESIDUAL, res=skip, emit_stats=True)

    def _skip_streams(self, x, x2, code):
        """True when the 1x1 skip convolution of this block runs on the streaming split-half kernel (and therefore wants act_scale(x, x2))."""
        if self.in_channels == self.out_channels:
            return False
        c_split = None if x2 is None else x.shape[3]
        stored = x.dtype == torch.bfloat16 and getattr(x, '_cf_stats', None) is not None and (x2 is None or getattr(x2, '_cf_stats', None) is not None)
        return bool((int(code) in (2,) + ops.SPLIT_CODES or stored) and ops.RANGE_SCALE and
                    ops.split_1x1_ok(self.in_channels, self.out_channels, x.shape[1], x.shape[2], c_split))

    def _skip_nhwc(self, x, x2, code):
        """The 1x1 skip convolution.  With split-half operands requested and an image of more than ops.TOKEN_IMAGE_MAX pixels it streams
        through the split-half convolution kernel (these layers are HBM-bound; the fp32 MFMA GEMM holds them at 2-4 TB/s); the input
        is the un-normalised block input, so it carries a range scale -- one table for both halves of a concatenated input."""
        # bf16 storage (x is a bf16 tensor that carries its producer's statistics): the same streaming kernel -- a bf16 value is its own hi
        # half, so the product is exact on the activation side; the fp32-MFMA GEMM ran these HBM-bound layers at 0.9 ms (128 -> 64 @512^2 x16)
        if self._skip_streams(x, x2, code):
            pw = self._packed(('conv_out', 'f16x2'), lambda: ops.pack_weight(self.conv_out.weight, self.conv_out.bias, bf16=ops.SPLIT),
                              self.conv_out.weight, self.conv_out.bias)
            return ops.conv2d(x, pw, x2=x2, act=ops.act_scale(x, x2))
        return ops.conv2d(x, self._pw_conv('conv_out'), x2=x2)

    def forward_host(self, x_in):
        x = self.conv1(swish(self.norm1(x_in)))
        x = self.conv2(swish(self.norm2(x)))
        if self.in_channels != self.out_channels:
            x_in = self.conv_out(x_in)
        return x + x_in


class AttnBlock(HipModule):
    """Single-head spatial self-attention (basicsr/archs/vqgan_arch.py:167-226).

    GPU: GN stats -> ONE fused q|k|v 1x1 GEMM with the GN-apply prologue -> cf_attention (256 keys, d=512,
    scale C^-0.5 on the scores) -> proj_out GEMM with the residual epilogue.
    """

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def forward_nhwc(self, x):
        B, H, W, C = x.shape
        if H * W != 256:
            raise ValueError(f'AttnBlock HIP path is built for 16x16 feature maps (got {H}x{W})')
        sc, sh = _gn_tables(self.norm, x)
        qkv_w = (self.q.weight, self.k.weight, self.v.weight)
        qkv_b = (self.q.bias, self.k.bias, self.v.bias)
        pw = self._packed('qkv', lambda: ops.pack_weight_cat(qkv_w, qkv_b), *qkv_w, *qkv_b)
        qkv = ops.conv2d(x, pw, prologue=PRO_AFFINE, scale=sc, shift=sh).view(B * 256, 3 * C)
        o = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, 1, C, int(C) ** (-0.5))
        return ops.conv2d(o.view(B, H, W, C), self._pw_conv('proj_out'), epilogue=EPI_RESIDUAL, res=x, emit_stats=True)

    def forward_host(self, x):
        h_ = self.norm(x)
        q, k, v = self.q(h_), self.k(h_), self.v(h_)
        b, c, h, w = q.shape
        w_ = torch.bmm(q.reshape(b, c, h * w).permute(0, 2, 1), k.reshape(b, c, h * w)) * (int(c) ** (-0.5))
        w_ = F.softmax(w_, dim=2).permute(0, 2, 1)
        h_ = torch.bmm(v.reshape(b, c, h * w), w_).reshape(b, c, h, w)
        return x + self.proj_out(h_)


class _Conv3x3(HipModule):
    """Plain 3x3 conv wrapper so that nn.Conv2d entries of the block lists get a HIP path; keeps the parameter
    names of nn.Conv2d (weight, bias) so the state_dict keys stay `blocks.<i>.weight`."""

    def __init__(self, cin, cout):
        super().__init__()
        conv = nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1)
        self.weight, self.bias = conv.weight, conv.bias
        self.in_channels, self.out_channels = cin, cout

    def pw(self, bf16=False):
        return self._packed(('w', int(bf16)), lambda: ops.pack_weight(self.weight, self.bias, bf16=int(bf16)), self.weight,
                            self.bias)

    def forward_nhwc(self, x, bf16=False, **kw):
        plain = not kw.get('out_nchw') and not kw.get('in_nchw')
        # operand code: 0 fp32 / 1 bf16 / 2 f16 / ops.WINOGRAD / ops.SPLIT, reduced to what this layer's shape supports
        h, w = (x.shape[2], x.shape[3]) if kw.get('in_nchw') else (x.shape[1], x.shape[2])
        unnormalised = plain and kw.get('prologue', ops.PRO_NONE) in (ops.PRO_NONE, ops.PRO_LEAKY)
        if unnormalised and int(bf16) == 2:
            bf16 = ops.SPLIT   # (single IEEE halves on the direct kernel have no range scaling; the split kernels do)
        code = ops.conv_code(bf16, self.in_channels, self.out_channels, h, w, plain=plain)
        pw = self.pw(code)
        if unnormalised and ops.needs_act_scale(pw):
            kw['act'] = ops.act_scale(x)
        return ops.conv2d(x, pw, **kw)

    def forward_host(self, x):
        return F.conv2d(x, self.weight, self.bias, stride=1, padding=1)


class _GroupNorm(nn.GroupNorm):
    """GroupNorm entry of a block list.  On GPU it is never run alone: the NEXT conv applies it in its gather
    (Encoder block 23->24, Generator block 23->24: GN then conv with no activation in between)."""

    def forward(self, x):
        if x.is_cuda:
            xn = ops.to_nhwc(x.float())
            sc, sh = ops.groupnorm_tables([xn], self.weight, self.bias, self.eps, self.num_groups)
            B, C = sc.shape
            return x * sc.view(B, C, 1, 1) + sh.view(B, C, 1, 1)  # module-level convenience only (not on the hot path)
        return super().forward(x)


def _run_blocks_nhwc(blocks, x, taps=None, first_nchw=False, last_nchw=False, bf16=False, storage_bf16=False):
    """Execute a block list on channels-last activations, folding GroupNorm entries into the following conv.

    taps: optional {block index: callable(x)} invoked after that block (encoder feature taps / generator fusions;
    a callable may return a replacement activation).
    storage_bf16 (generator, operand code 1 only -- precision 'bf16', BASELINE configs 3 / 5): every activation of more than
    ops.TOKEN_IMAGE_MAX pixels per image lives in HBM as bf16 (cf_conv_desc.io_bf16).  The switch happens in front of the first Upsample
    whose output is that large: its fp32 input (32x32) is copied to bf16 once (ops.to_bf16) and from there every kernel reads and writes
    bf16 -- the storage type of a launch is the dtype of its input tensor; the final 64 -> 3 conv writes the fp32 NCHW image.
    """
    pending = None
    n = len(blocks)
    for i, blk in enumerate(blocks):
        if storage_bf16 and int(bf16) == 1 and isinstance(blk, Upsample) and x.dtype == torch.float32 and \
                4 * x.shape[1] * x.shape[2] > ops.TOKEN_IMAGE_MAX:
            x = ops.to_bf16(x)
        if isinstance(blk, _GroupNorm):
            pending = ops.groupnorm_tables([x], blk.weight, blk.bias, blk.eps, blk.num_groups) + (blk,)
        elif isinstance(blk, _Conv3x3):
            kw = {}
            code = bf16
            if pending is not None:
                kw.update(prologue=PRO_AFFINE, scale=pending[0], shift=pending[1])
                code = blk._range_code(pending[2], bf16, x.shape[1] * x.shape[2] * x.shape[3] // pending[2].num_groups)
                pending = None
            if i == 0 and first_nchw:
                kw['in_nchw'] = True
            if i == n - 1 and last_nchw:
                kw['out_nchw'] = True
            kw['emit_stats'] = i != n - 1      # every inner conv feeds a GroupNorm of the next block
            x = blk.forward_nhwc(x, bf16=code, **kw)
        else:
            if pending is not None:
                raise RuntimeError('GroupNorm must be followed by a conv in the block list')
            x = blk.forward_nhwc(x, bf16=bf16) if isinstance(blk, (ResBlock, Upsample, Downsample)) else blk.forward_nhwc(x)
        if taps and i in taps:
            r = taps[i](x)
            if r is not None:
                x = r
    return x


class Encoder(HipModule):
    """VQGAN encoder (basicsr/archs/vqgan_arch.py:229-273): 25 blocks for the CodeFormer configuration."""

    def __init__(self, in_channels, nf, emb_dim, ch_mult, num_res_blocks, resolution, attn_resolutions):
        super().__init__()
        self.nf = nf
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.attn_resolutions = attn_resolutions
        res = resolution
        widths = [nf * m for m in (1,) + tuple(ch_mult)]
        seq = [_Conv3x3(in_channels, nf)]
        ch = nf
        for level in range(self.num_resolutions):
            ch = widths[level]
            for _ in range(num_res_blocks):
                seq.append(ResBlock(ch, widths[level + 1]))
                ch = widths[level + 1]
                if res in attn_resolutions:
                    seq.append(AttnBlock(ch))
            if level != self.num_resolutions - 1:
                seq.append(Downsample(ch))
                res //= 2
        seq += [ResBlock(ch, ch), AttnBlock(ch), ResBlock(ch, ch), _GroupNorm(GN_GROUPS, ch, eps=GN_EPS, affine=True),
                _Conv3x3(ch, emb_dim)]
        self.blocks = nn.ModuleList(seq)

    def forward(self, x):
        if not x.is_cuda:
            return self.forward_host(x)
        with torch.no_grad():
            return ops.to_nchw(self.forward_nhwc(x.float().contiguous()))

    def forward_nhwc(self, x_nchw, taps=None, bf16=False):
        """x_nchw: the (B,3,H,W) network input (read directly by the first conv); returns NHWC.
        bf16: operand code for the 3x3 convs (only 0 = direct fp32 or ops.WINOGRAD = fp32 Winograd make sense here: the code
        indices depend on the encoder)."""
        return _run_blocks_nhwc(self.blocks, x_nchw, taps, first_nchw=True, bf16=bf16)

    def forward_host(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


class Generator(HipModule):
    """VQGAN decoder (basicsr/archs/vqgan_arch.py:276-323): 25 blocks for the CodeFormer configuration."""

    def __init__(self, nf, emb_dim, ch_mult, res_blocks, img_size, attn_resolutions):
        super().__init__()
        self.nf = nf
        self.ch_mult = ch_mult
        self.num_resolutions = len(self.ch_mult)
        self.num_res_blocks = res_blocks
        self.resolution = img_size
        self.attn_resolutions = attn_resolutions
        self.in_channels = emb_dim
        self.out_channels = 3
        ch = nf * ch_mult[-1]
        res = img_size // 2 ** (self.num_resolutions - 1)
        seq = [_Conv3x3(emb_dim, ch), ResBlock(ch, ch), AttnBlock(ch), ResBlock(ch, ch)]
        for level in reversed(range(self.num_resolutions)):
            width = nf * ch_mult[level]
            for _ in range(res_blocks):
                seq.append(ResBlock(ch, width))
                ch = width
                if res in attn_resolutions:
                    seq.append(AttnBlock(ch))
            if level != 0:
                seq.append(Upsample(ch))
                res *= 2
        seq += [_GroupNorm(GN_GROUPS, ch, eps=GN_EPS, affine=True), _Conv3x3(ch, self.out_channels)]
        self.blocks = nn.ModuleList(seq)

    def forward(self, x):
        if not x.is_cuda:
            return self.forward_host(x)
        with torch.no_grad():
            return self.forward_nhwc(ops.to_nhwc(x.float()))

    def forward_nhwc(self, x, taps=None, bf16=False, storage_bf16=False):
        """x: (B,16,16,C) NHWC latent; returns the (B,3,H,W) NCHW image (written directly by the last conv).
        bf16=True: every 3x3 conv except the final 64->3 one uses bf16 MFMA operands; storage_bf16: see _run_blocks_nhwc."""
        return _run_blocks_nhwc(self.blocks, x, taps, last_nchw=True, bf16=bf16, storage_bf16=storage_bf16)

    def forward_host(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


@ARCH_REGISTRY.register()
class VQAutoEncoder(HipModule):
    """Stage-I VQGAN autoencoder (basicsr/archs/vqgan_arch.py:326-389)."""

    def __init__(self, img_size, nf, ch_mult, quantizer='nearest', res_blocks=2, attn_resolutions=[16],
                 codebook_size=1024, emb_dim=256, beta=0.25, gumbel_straight_through=False, gumbel_kl_weight=1e-8,
                 model_path=None):
        super().__init__()
        logger = get_root_logger()
        self.in_channels = 3
        self.nf = nf
        self.n_blocks = res_blocks
        self.codebook_size = codebook_size
        self.embed_dim = emb_dim
        self.ch_mult = ch_mult
        self.resolution = img_size
        self.attn_resolutions = attn_resolutions
        self.quantizer_type = quantizer
        self.encoder = Encoder(self.in_channels, nf, emb_dim, ch_mult, res_blocks, img_size, attn_resolutions)
        if quantizer == 'nearest':
            self.beta = beta
            self.quantize = VectorQuantizer(codebook_size, emb_dim, beta)
        else:
            # GumbelQuantizer is a training-time alternative (vqgan_arch.py:87-114); outside the inference hot path.
            raise NotImplementedError(f"quantizer '{quantizer}' is not part of the MI355X inference path")
        self.generator = Generator(nf, emb_dim, ch_mult, res_blocks, img_size, attn_resolutions)
        if model_path is not None:
            chkpt = torch.load(model_path, map_location='cpu')
            key = 'params_ema' if 'params_ema' in chkpt else ('params' if 'params' in chkpt else None)
            if key is None:
                raise ValueError('Wrong params!')
            self.load_state_dict(chkpt[key])
            logger.info(f'vqgan is loaded from: {model_path} [{key}]')

    def forward(self, x):
        x = self.encoder(x)
        quant, codebook_loss, quant_stats = self.quantize(x)
        x = self.generator(quant)
        return x, codebook_loss, quant_stats
