"""CodeFormer network, MI355X-native.

API mirror of the reference's basicsr/archs/codeformer_arch.py: `ARCH_REGISTRY.get('CodeFormer')(dim_embd=512,
codebook_size=1024, n_head=8, n_layers=9, connect_list=[...])`, identical state_dict keys / init order, and
`net(x, w=..., adain=...)` -> `(out, logits, lq_feat)` (`code_only=True` -> `(logits, lq_feat)`).

On ROCm tensors `CodeFormer.forward` (reference: codeformer_arch.py:223-280) runs entirely on the HIP kernels of
codeformer_amd/csrc with channels-last activations:
  encoder (fused GN/swish/conv stacks, taps kept by reference instead of .clone())
  -> 9 pre-LN Transformer layers on (B*256, 512) token matrices (LayerNorm kernel, fp32-MFMA GEMMs with bias/GELU/
     residual epilogues, 8-head attention kernel)
  -> logits GEMM -> wavefront-shuffle argmax (== softmax+topk(1), lowest index on ties) -> codebook gather (+AdaIN)
  -> generator with the controllable feature transform fused into conv gathers / epilogues.
"""
import os
import threading

import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ..ops import EPI_GELU, EPI_RESIDUAL, EPI_SFT, PRO_LEAKY
from ..utils.registry import ARCH_REGISTRY
from .hip_module import PACK_EPOCH, HipModule
from .vqgan_arch import ResBlock, VQAutoEncoder


# The q | k and v projections of a Transformer layer as one split-half token GEMM (round 6; CODEFORMER_HIP_QKV_ONE_LAUNCH=0: two launches, A/B --
# the same bits either way)
QKV_ONE_LAUNCH = os.environ.get('CODEFORMER_HIP_QKV_ONE_LAUNCH', '1') != '0'

# Graphed forwards share static input / output buffers per captured shape, so capture and replay are serialised (one lock for the
# process: module attributes must stay picklable / deep-copyable).  The eager path (more than `graph_max_batch` faces) is re-entrant.
_GRAPH_LOCK = threading.RLock()
_CAPTURE_STREAMS = {}    # device -> the stream every graph of this process is captured on (its scratch words exist before the first capture)


def calc_mean_std(feat, eps=1e-5):
    """Per-(b,c) mean and sqrt(unbiased var + eps) of a 4-D tensor (codeformer_arch.py:12-26); host helper."""
    assert feat.dim() == 4, 'The input feature should be 4D tensor.'
    b, c = feat.shape[:2]
    flat = feat.reshape(b, c, -1)
    std = (flat.var(dim=2) + eps).sqrt().view(b, c, 1, 1)
    mean = flat.mean(dim=2).view(b, c, 1, 1)
    return mean, std


def adaptive_instance_normalization(content_feat, style_feat):
    """AdaIN (codeformer_arch.py:29-43).  NCHW in/out; on GPU the statistics + re-normalisation run in
    cf_codebook_gather_adain only inside CodeFormer.forward -- this free function is the host helper."""
    size = content_feat.size()
    style_mean, style_std = calc_mean_std(style_feat)
    content_mean, content_std = calc_mean_std(content_feat)
    normalized = (content_feat - content_mean.expand(size)) / content_std.expand(size)
    return normalized * style_std.expand(size) + style_mean.expand(size)


class TransformerSALayer(HipModule):
    """Pre-LN self-attention + MLP layer (codeformer_arch.py:99-134).

    GPU path works on batch-major token matrices X[(b*256 + t), 512] (the NHWC view of the 16x16 latent), i.e.
    the reference's (T, B, C) sequence-first tensor with the first two axes swapped.
    """

    def __init__(self, embed_dim, nhead=8, dim_mlp=2048, dropout=0.0, activation='gelu'):
        super().__init__()
        if activation != 'gelu':
            raise RuntimeError(f'activation should be gelu for the MI355X path, not {activation}.')
        self.self_attn = nn.MultiheadAttention(embed_dim, nhead, dropout=dropout)
        self.linear1 = nn.Linear(embed_dim, dim_mlp)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_mlp, embed_dim)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.embed_dim, self.nhead = embed_dim, nhead

    def with_pos_embed(self, tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_tokens(self, X, pos, batch, code=0, code_ln=None):
        """X: (batch*256, E) tokens, pos: (256, E) or None.  Returns the layer output, same shape.
        code: operand code of the five GEMMs (0 exact fp32 MFMA, ops.GSPLIT split-half operands); code_ln: the code of the three that
        read a LayerNorm output (q|k, v, MLP-up; default: `code`) -- their inputs are bounded by the norm's parameters (ln_code)."""
        E, H = self.embed_dim, self.nhead
        sa = self.self_attn
        w, b = sa.in_proj_weight, sa.in_proj_bias
        code_ln = code if code_ln is None else code_ln
        c1 = ln_code(self, self.norm1, code_ln, pos)
        c2 = ln_code(self, self.norm2, code_ln)
        # out-proj reads convex combinations of v = Wv LN1(x) + bv, MLP-down reads GELU(W1 LN2(x) + b1) with |GELU(h)| <= |h|: both inherit
        # a bound from the norm's parameters and one weight matrix (bounded_code); `code` (gemm_precision='f16x2') overrides the check
        c_o = code or bounded_code(self, 'o', code_ln, self.norm1, w[2 * E:], b[2 * E:])
        c_d = code or bounded_code(self, 'd', code_ln, self.norm2, self.linear1.weight, self.linear1.bias)
        pw_o = self._pw_conv(sa.out_proj, bf16=c_o)
        if pos is not None:
            t2, t2p = ops.layernorm(X, self.norm1.weight, self.norm1.bias, self.norm1.eps, pos=pos)
        else:
            t2 = t2p = ops.layernorm(X, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        if c1 == ops.GSPLIT and QKV_ONE_LAUNCH and (2 * E) % 128 == 0:
            # split-half token GEMM: the whole in_proj (3 E rows) in ONE launch -- the q | k columns contract LN(x) + pos, the v columns LN(x)
            # (cf_conv_desc.in0_alt; per output element the arithmetic of the two launches below -- with ONE power-of-two pack scale for the whole
            #  in_proj instead of one per part, so q, k, v agree with the two-launch form to the last bits of the 22-bit operands, not bitwise)
            pw_qkv = self._packed(('qkv', c1), lambda: ops.pack_weight(w, b, bf16=c1), w, b)
            qkv = ops.linear(t2p, pw_qkv, x_alt=t2, alt_from=2 * E)
            q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
        else:
            pw_qk = self._packed(('qk', c1), lambda: ops.pack_weight(w[:2 * E], b[:2 * E], bf16=c1), w, b)
            pw_v = self._packed(('v', c1), lambda: ops.pack_weight(w[2 * E:], b[2 * E:], bf16=c1), w, b)
            qk = ops.linear(t2p, pw_qk)                      # q | k share the (LN(x)+pos) input
            v = ops.linear(t2, pw_v)                         # v = LN(x) without pos (codeformer_arch.py:125-126)
            q, k = qk[:, :E], qk[:, E:]
        hd = E // H
        a = ops.attention(q, k, v, batch, H, hd, float(hd) ** -0.5)
        X = ops.linear(a, pw_o, epilogue=EPI_RESIDUAL, res=X)
        t2 = ops.layernorm(X, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.linear(t2, self._pw_conv('linear1', bf16=c2), epilogue=EPI_GELU)
        return ops.linear(h, self._pw_conv('linear2', bf16=c_d), epilogue=EPI_RESIDUAL, res=X)

    def forward(self, tgt, tgt_mask=None, tgt_key_padding_mask=None, query_pos=None):
        """tgt: (T=256, B, E) sequence-first like the reference."""
        if tgt.is_cuda:
            if tgt_mask is not None or tgt_key_padding_mask is not None:
                raise NotImplementedError('attention masks are not part of the CodeFormer inference path')
            with torch.no_grad():
                T, B, E = tgt.shape
                X = tgt.float().transpose(0, 1).contiguous().view(B * T, E)
                pos = None
                if query_pos is not None:
                    pos = query_pos.float()[:, 0, :].contiguous()   # the reference broadcasts one table over the batch
                Y = self.forward_tokens(X, pos, B)
                return Y.view(B, T, E).transpose(0, 1).contiguous()
        tgt2 = self.norm1(tgt)
        q = k = self.with_pos_embed(tgt2, query_pos)
        tgt2 = self.self_attn(q, k, value=tgt2, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
        tgt = tgt + self.dropout1(tgt2)
        tgt2 = self.norm2(tgt)
        tgt2 = self.linear2(self.dropout(F.gelu(self.linear1(tgt2))))
        return tgt + self.dropout2(tgt2)


def ln_code(module, norm, code, pos=None):
    """Operand code of a Linear layer that reads LayerNorm `norm`'s output (+ the position table): `code`, or 0 (exact fp32) when the bound
    max_i(|gamma_i| sqrt(C - 1) + |beta_i|) + max|pos| could leave the IEEE-half range of the split-half token GEMM, which has no range
    scale.  The maxima are read back once per parameter version (HipModule._packed)."""
    if int(code) != ops.GSPLIT:
        return int(code)
    params = (norm.weight, norm.bias) + (() if pos is None else (pos,))
    bound = module._packed(('ln_range', id(norm), pos is not None),
                           lambda: float((norm.weight.detach().abs() * math.sqrt(norm.weight.numel() - 1) + norm.bias.detach().abs()).max())
                           + (0.0 if pos is None else float(pos.detach().abs().max())), *params)
    return ops.GSPLIT if bound < 32768.0 else 0


def bounded_code(module, tag, code, norm, weight, bias):
    """Operand code of a Linear layer whose input is bounded by max_n(|W| L + |b|)_n with L = |gamma| sqrt(C - 1) + |beta| the bound of
    LayerNorm `norm`'s output and (W, b) the Linear layer in between (out-proj: the value projection -- attention outputs are convex
    combinations of values; MLP-down: the MLP-up layer -- |GELU(h)| <= |h|): `code` while that stays inside the IEEE-half range of the
    split-half token GEMM, else 0 (exact fp32).  One matrix-vector product per parameter version."""
    if int(code) != ops.GSPLIT:
        return int(code)

    def bound():
        L = norm.weight.detach().abs().float() * math.sqrt(norm.weight.numel() - 1) + norm.bias.detach().abs().float()
        v = weight.detach().abs().float() @ L
        if bias is not None:
            v = v + bias.detach().abs().float()
        return float(v.max())
    params = (norm.weight, norm.bias, weight) + (() if bias is None else (bias,))
    return ops.GSPLIT if module._packed(('lin_range', tag), bound, *params) < 32768.0 else 0


class Fuse_sft_block(HipModule):
    """Controllable feature transform (codeformer_arch.py:136-157):
    e = ResBlock(cat[enc, dec]); out = dec + w * (dec * scale(e) + shift(e)).

    GPU: the concat is two base pointers in the conv gather; LeakyReLU(0.2) is the gather prologue of the second
    conv of each branch; the SFT combine is the epilogue of the last `shift` conv.
    """

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.encode_enc = ResBlock(2 * in_ch, out_ch)
        self.scale = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=3, padding=1), nn.LeakyReLU(0.2, True),
                                   nn.Conv2d(out_ch, out_ch, kernel_size=3, padding=1))
        self.shift = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=3, padding=1), nn.LeakyReLU(0.2, True),
                                   nn.Conv2d(out_ch, out_ch, kernel_size=3, padding=1))

    def forward_nhwc(self, enc, dec, w=1, bf16=False):
        e = self.encode_enc.forward_nhwc(enc, dec, bf16=bf16)
        hw = e.shape[1:3]
        if int(bf16) == 2 and not ops.wsingle_ok(e.shape[3], e.shape[3], hw[0], hw[1]):
            bf16 = ops.SPLIT   # (single IEEE halves on the direct kernel have no range scaling; un-normalised inputs need it)
        # The four convs below read UN-NORMALISED tensors (e; the outputs of scale.0 / shift.0): on the 16-bit-operand kernels each
        # gets a per-image power-of-two range scale derived from the statistics its producer wrote (ops.act_scale).
        pws = [self._pw_conv(m, bf16, hw=hw) for m in (self.scale[0], self.scale[2], self.shift[0], self.shift[2])]
        rs = [ops.needs_act_scale(p) for p in pws]
        act_e = ops.act_scale(e) if (rs[0] or rs[2]) else None
        if rs[1] and rs[3] and ops.ACT_FUSED:
            # both second convolutions want a table: the two first convolutions write their statistics into one buffer and ONE launch turns
            # them into both tables (bitwise act_scale of each) -- 4 launches fewer per forward
            pair = ops.StatsPair()
            s0 = ops.conv2d(e, pws[0], act=act_e, emit_stats=True, stats_into=pair)
            h = ops.conv2d(e, pws[2], act=act_e, emit_stats=True, stats_into=pair)
            act_s, act_h = ops.act_scale_pair(pair, e.shape[0])
        else:
            s0 = ops.conv2d(e, pws[0], act=act_e, emit_stats=rs[1])
            h = ops.conv2d(e, pws[2], act=act_e, emit_stats=rs[3])
            act_s, act_h = (ops.act_scale(s0) if rs[1] else None), (ops.act_scale(h) if rs[3] else None)
        s = ops.conv2d(s0, pws[1], prologue=PRO_LEAKY, act=act_s)
        return ops.conv2d(h, pws[3], prologue=PRO_LEAKY, epilogue=EPI_SFT, res=dec, sft_scale=s, sft_w=float(w), emit_stats=True, act=act_h)

    def forward(self, enc_feat, dec_feat, w=1):
        if enc_feat.is_cuda:
            with torch.no_grad():
                out = self.forward_nhwc(ops.to_nhwc(enc_feat.float()), ops.to_nhwc(dec_feat.float()), w)
                return ops.to_nchw(out)
        enc = self.encode_enc(torch.cat([enc_feat, dec_feat], dim=1))
        return dec_feat + w * (dec_feat * self.scale(enc) + self.shift(enc))


@ARCH_REGISTRY.register()
class CodeFormer(VQAutoEncoder):

    def __init__(self, dim_embd=512, n_head=8, n_layers=9, codebook_size=1024, latent_size=256,
                 connect_list=['32', '64', '128', '256'], fix_modules=['quantize', 'generator'], vqgan_path=None):
        super().__init__(512, 64, [1, 2, 2, 4, 4, 8], 'nearest', 2, [16], codebook_size)
        if vqgan_path is not None:
            self.load_state_dict(torch.load(vqgan_path, map_location='cpu')['params_ema'])
        if fix_modules is not None:
            for name in fix_modules:
                for p in getattr(self, name).parameters():
                    p.requires_grad = False

        # Operand format of the convolutions (3x3 stride 1 / 2 / folded upsample and, for the split-half codes, the 1x1 skips on images).
        # Tensors, accumulation, attention, statistics and the code argmax are fp32 in every mode; the Transformer's Linear layers: see
        # gemm_precision.
        #   'f16x2' (default): fp32 operands split into hi + lo IEEE halves (22 significant bits), three f16 MFMAs per product, fp32
        #            accumulation, in the Winograd F(2x2,3x3) domain where the layer allows it (cf_wsplit.hip / cf_winograd.hip H2; folded
        #            upsample, stride-2 and image-sized 1x1 convs: cf_split.hip), for generator, CFT and -- see encoder_precision -- the encoder.  Per layer at or below
        #            the fp64-error of the exact kernels; whole network vs the reference on real crops: pixels 7.2e-5 (exact path
        #            7.1e-5; tolerance 1e-3), logits 6.7e-6 (exact: 7.6e-6; tolerance 1e-4), code indices identical.  1.5x the exact
        #            path's faces/s.
        #   'fp32':  everything on exact fp32 MFMA operands (Winograd where eligible: F(2x2,3x3), and F(4x4,3x3) in generator / CFT, see below).
        #   'bf16' (BASELINE configs 3/5) / 'fp16': single 16-bit operands with fp32 accumulate in generator + CFT -- in the Winograd domain
        #            (one MFMA per transform-domain product, cf_wsplit.hip) for the 128-channel-tile layers from 32x32 up, the direct 16-bit
        #            kernel elsewhere; the encoder runs as in the default mode (split halves, fp32-grade), so logits and code indices are
        #            bitwise those of 'f16x2' (pixel gates in tests/test_gpu_real_images.py).
        self.precision = os.environ.get('CODEFORMER_HIP_PRECISION', 'f16x2')
        # precision 'bf16' as BASELINE configs 3 / 5 define it: "bf16 storage + fp32 accumulate in generator + CFT" (round 6).  With the
        # switch on (default) every generator / fusion-block activation of more than 1024 pixels per image -- the 64x64 .. 512x512 levels,
        # 99.6 % of the decoder's bytes -- lives in HBM as bf16: the producing epilogue rounds once, after its GroupNorm partials were taken
        # from the fp32 accumulators; consumers widen on load; the four encoder taps the fusion blocks read get a bf16 copy.  Encoder,
        # Transformer, argmax, the 16x16 / 32x32 latents (AttnBlocks, AdaIN) stay fp32, so logits and code indices are bitwise those of the
        # default mode.  CODEFORMER_HIP_BF16_STORAGE=0 / False: bf16 OPERANDS on fp32 tensors (rounds 2-5).
        self.bf16_storage = os.environ.get('CODEFORMER_HIP_BF16_STORAGE', '1') != '0'
        # Exact-fp32 convolutions (precision='fp32', and in every mode the layers the split kernel does not take): evaluate 3x3
        # stride-1 convolutions with Winograd F(2x2,3x3) -- the same function in fp32 with 2.25x fewer multiplies (cf_winograd.hip).
        # Set False (or CODEFORMER_HIP_WINOGRAD=0) for the direct evaluation everywhere.
        self.winograd = os.environ.get('CODEFORMER_HIP_WINOGRAD', '1') != '0'
        # precision 'f16x2' and 'fp32': generator / CFT layers the F(4x4,3x3) kernel covers (ops.f43_ok; which shapes: CODEFORMER_HIP_F43) run
        # there (split-half / IEEE-fp32 operands) -- 2.25 instead of 4 transform-domain products per output.  Its error against fp64 is ~5x that of F(2x2,3x3): far inside
        # the pixel tolerance; the encoder's use of it has its own switch and gate (next attribute).  False / CODEFORMER_HIP_F43=0: F(2x2,3x3) everywhere.
        self.winograd_f43 = ops.F43_LAYERS != '0'
        # The same kernel for the ENCODER's covered layers (the 64-channel 512^2, 128-channel 256^2 / 128^2 stages and, for fp32 operands, the
        # 256-channel 64^2 stage): on by default since round 5 (CODEFORMER_HIP_F43_ENCODER=0 / False: F(2x2,3x3) there, the round-2..4 encoder).
        # The encoder decides the code indices and F(4x4,3x3) carries ~5x the per-layer error of F(2x2,3x3), so the switch was admitted on
        # a measured margin, not on "no index changed": for every golden token whose reference top-2 gap is >= 1e-5 the ratio
        # (reference gap) / (2 x max |our logit - reference logit|) must stay >= 5 (review of round 4) -- measured 7.0 (split halves) /
        # 7.3 (fp32 operands) with the switch on against 9.9 / 8.7 with it off, over the seeded face, the three real crops, the masked face,
        # the four range variants and the 32-face sweep; no index differs anywhere; logits move by 5-11e-6 (tolerance 1e-4).  It buys 4 % of
        # the step in the default mode and 9 % in 'fp32'.  Gate: tests/test_gpu_real_images.py::test_encoder_logit_margin (tools/logit_margin.py,
        # profiles/r05_logit_margin.txt).
        self.winograd_f43_encoder = os.environ.get('CODEFORMER_HIP_F43_ENCODER', '1') == '1'
        # Also evaluate the ENCODER's 3x3 stride-1 convolutions with Winograd, in every precision mode (the encoder is always
        # fp32, so logits / indices stay bitwise identical across 'fp32' / 'bf16' / 'fp16').  Measured against the reference:
        # logits 4.3e-6 (direct kernel 5.5e-6), lq_feat 1.0e-5 (1.4e-5), indices exact on every seeded face incl. one whose
        # top-2 gap is 1.7e-5 -- the Winograd form sums fewer products per output and is, if anything, the more accurate one.
        # Round 2 added the reference's own crops (three PNGs, one masked face) and an 8-face sweep to the gate: indices equal the
        # reference's on every token whose reference gap is >= 1e-5 (tests/test_gpu_real_images.py).
        self.winograd_encoder = os.environ.get('CODEFORMER_HIP_WINOGRAD_ENCODER', '1') != '0'
        # Operand format of the ENCODER's 3x3 stride-1 convolutions: 'fp32' (exact fp32 MFMA, Winograd where eligible), 'f16x2' (split
        # halves on every layer but the first conv -- the 16x16 latents included, on the four-wave Winograd kernel with split-K; only with
        # winograd = False do the latents stay on the exact direct kernel) or 'auto' = 'fp32' when
        # precision is 'fp32', 'f16x2' otherwise (a 16-bit generator does not ask for an exact-fp32 encoder at 0.54 of the fp32 MFMA peak).  The code indices hang on the encoder, so this was measured before it became the
        # default (tools/encoder_split_check.py, profiles/r02_encoder_split_check.txt): against the reference's logits on its own
        # crops the split encoder is as close as the exact one (max 6.7e-6 / 5.4e-6 / 5.7e-6 vs 7.6e-6 / 5.5e-6 / 6.0e-6; the
        # reference's own 1-vs-8-thread noise is 2.6e-6), the smallest (reference top-2 gap) / (2 x our logit error) over all tokens
        # is 9.1 (exact: 8.8), and every index of every golden agrees.  Set 'fp32' to keep logits bitwise equal across precisions.
        self.encoder_precision = os.environ.get('CODEFORMER_HIP_ENCODER_PRECISION', 'auto')
        # Operand format of the Transformer's Linear layers (feat_emb, q|k / v / out projections, MLP, logits head): 'fp32' = exact fp32
        # MFMA GEMM; 'f16x2' = split-half operands (cf_gemm_split.hip) everywhere -- opt-in: the token GEMM has no range scale and the
        # inputs of out-proj / MLP-down / feat_emb are not bounded by anything the host can check; 'auto' (default) = split halves for the
        # Linear layers whose input the host can bound from parameters -- those behind a LayerNorm (q|k, v, MLP-up, the logits head:
        # |gamma| sqrt(C - 1) + |beta| (+ |pos|), ln_code) and those one Linear layer further (out-proj, MLP-down: bounded_code), 46 of 47
        # launches -- while the bound stays inside the half range and neither `precision` nor `encoder_precision` is 'fp32' (so that
        # encoder_precision = 'fp32' alone keeps the logits bitwise identical across modes); exact fp32 elsewhere (feat_emb reads the
        # un-normalised encoder feature).
        # Against fp64 the split-half GEMM is closer than the fp32-MFMA one (1.2e-6 vs 2.0e-6); logits / indices vs the reference unchanged.
        self.gemm_precision = os.environ.get('CODEFORMER_HIP_GEMM_PRECISION', 'auto')
        # Optional HIP-graph replay of the whole forward (one graph per input shape / w / flags): takes the ~250 host launches
        # per call off the critical path.  Measured: no gain at B=1..16 on an otherwise idle host (the kernels, not the launches,
        # bound even B=1), so it is off by default; useful when the host thread is busy (decode / encode of PNGs).
        # Round 3 re-measurement: with the faster kernels a one-face call is host-bound in places when run eagerly (the Transformer section:
        # 1.6 ms of host enqueue for 0.9 ms of GPU work), and replay takes the whole call from 6.76 to 6.55 ms on a box where the host is the
        # slower side.  Round 4: 'auto' (default) replays a graph for batches of at most `graph_max_batch` faces -- the reference's own call
        # pattern is one face per call (inference_codeformer.py:197-206) -- behind a parameter signature (sum of the parameters' versions
        # and storage addresses, ~70 us per call): a versioned in-place update, load_state_dict, .to() or invalidate_packed_weights() all
        # re-capture; only an un-versioned `.data` edit needs the explicit invalidate_packed_weights() the packed-weight cache needs too.
        # '1': every batch size; '0': never.  At most four captured graphs are kept (each pins the activations of its shape).
        self.use_hip_graphs = {'0': False, '1': True}.get(os.environ.get('CODEFORMER_HIP_GRAPHS', 'auto'), 'auto')
        self.graph_max_batch = 4
        self._graphs = {}
        self.connect_list = connect_list
        self.n_layers = n_layers
        self.n_head = n_head
        self.dim_embd = dim_embd
        self.dim_mlp = dim_embd * 2
        self.latent_size = latent_size

        self.position_emb = nn.Parameter(torch.zeros(latent_size, self.dim_embd))
        self.feat_emb = nn.Linear(256, self.dim_embd)
        self.ft_layers = nn.Sequential(*[
            TransformerSALayer(embed_dim=dim_embd, nhead=n_head, dim_mlp=self.dim_mlp, dropout=0.0)
            for _ in range(self.n_layers)])
        self.idx_pred_layer = nn.Sequential(nn.LayerNorm(dim_embd), nn.Linear(dim_embd, codebook_size, bias=False))

        self.channels = {'16': 512, '32': 256, '64': 256, '128': 128, '256': 128, '512': 64}
        # encoder tap after the 2nd ResBlock of a level; generator fusion after the 1st ResBlock of a level
        self.fuse_encoder_block = {'512': 2, '256': 5, '128': 8, '64': 11, '32': 14, '16': 18}
        self.fuse_generator_block = {'16': 6, '32': 9, '64': 12, '128': 15, '256': 18, '512': 21}
        self.fuse_convs_dict = nn.ModuleDict()
        for f_size in self.connect_list:
            ch = self.channels[f_size]
            self.fuse_convs_dict[f_size] = Fuse_sft_block(ch, ch)

    # ------------------------------------------------------------------ GPU (HIP) path
    def _forward_hip(self, x, w, code_only, adain):
        B, _, Himg, Wimg = x.shape
        if (Himg, Wimg) != (512, 512):
            raise ValueError(f'CodeFormer expects aligned 512x512 faces, got {Himg}x{Wimg}')
        x = x.float().contiguous()
        enc_feat = {}
        enc_taps = {self.fuse_encoder_block[f]: (lambda t: enc_feat.__setitem__(str(t.shape[2]), t))
                    for f in self.connect_list}
        if self.encoder_precision not in ('auto', 'fp32', 'f16x2'):
            raise ValueError(f"encoder_precision must be 'auto', 'fp32' or 'f16x2', got {self.encoder_precision!r}")
        enc_code = ops.WINOGRAD if (self.winograd and self.winograd_encoder) else 0
        if self.encoder_precision == 'f16x2' or (self.encoder_precision == 'auto' and self.precision != 'fp32'):
            enc_code = ops.SPLIT if enc_code == ops.WINOGRAD else ops.SPLIT_DIRECT
        if self.winograd_f43_encoder and self.winograd_f43:
            enc_code = {ops.SPLIT: ops.SPLIT_F43, ops.WINOGRAD: ops.WINOGRAD_F43}.get(enc_code, enc_code)
        lq = self.encoder.forward_nhwc(x, enc_taps, bf16=enc_code)        # (B,16,16,256) channels-last
        T = lq.shape[1] * lq.shape[2]
        tokens = lq.view(B * T, lq.shape[3])

        if self.gemm_precision not in ('auto', 'fp32', 'f16x2'):
            raise ValueError(f"gemm_precision must be 'auto', 'fp32' or 'f16x2', got {self.gemm_precision!r}")
        gcode = ops.GSPLIT if self.gemm_precision == 'f16x2' else 0
        # 'auto' follows the encoder: encoder_precision = 'fp32' alone keeps logits bitwise identical across the precision modes
        gcode_ln = ops.GSPLIT if (self.gemm_precision == 'f16x2' or (self.gemm_precision == 'auto' and self.precision != 'fp32' and self.encoder_precision != 'fp32')) else 0
        X = ops.linear(tokens, self._pw_conv('feat_emb', bf16=gcode))
        for layer in self.ft_layers:
            X = layer.forward_tokens(X, self.position_emb, B, code=gcode, code_ln=gcode_ln)
        ln, head = self.idx_pred_layer[0], self.idx_pred_layer[1]
        logits2d = ops.linear(ops.layernorm(X, ln.weight, ln.bias, ln.eps), self._pw_conv(head, bf16=ln_code(self, ln, gcode_ln)))
        logits = logits2d.view(B, T, -1)
        lq_feat = ops.to_nchw(lq)
        if code_only:
            return logits, lq_feat

        idx = ops.argmax_rows(logits2d)                                     # == topk(softmax(logits), 1)
        quant = ops.codebook_gather(idx, self.quantize.embedding.weight, B, T,
                                    lq=tokens.view(B, T, -1) if adain else None)
        quant = quant.view(B, lq.shape[1], lq.shape[2], -1)

        if self.precision not in ('fp32', 'f16x2', 'bf16', 'fp16'):
            raise ValueError(f"precision must be 'fp32', 'f16x2', 'bf16' or 'fp16', got {self.precision!r}")
        bf16 = {'fp32': 0, 'f16x2': ops.SPLIT, 'bf16': 1, 'fp16': 2}[self.precision]   # operand code of the generator + CFT 3x3 convs
        if bf16 == 0 and self.winograd:
            bf16 = ops.WINOGRAD
        if bf16 == ops.SPLIT and not self.winograd:
            bf16 = ops.SPLIT_DIRECT
        if bf16 == ops.WINOGRAD and self.winograd_f43:
            bf16 = ops.WINOGRAD_F43   # the same layers on fp32 operands (four v_mfma_f32_16x16x4_f32 per 16 channels instead of three f16 MFMAs)
        if bf16 == ops.SPLIT and self.winograd_f43:
            bf16 = ops.SPLIT_F43   # (the encoder gets this code through winograd_f43_encoder only, above)
        gen_taps = None
        if w > 0:
            def fuse(t):
                f = str(t.shape[2])
                enc = enc_feat[f]
                if t.dtype == torch.bfloat16:    # bf16 storage: the fusion block concatenates [enc, dec] -- the tap gets its bf16 copy
                    enc = ops.to_bf16(enc)
                return self.fuse_convs_dict[f].forward_nhwc(enc, t, w, bf16=bf16)
            gen_taps = {self.fuse_generator_block[f]: fuse for f in self.connect_list}
        out = self.generator.forward_nhwc(quant, gen_taps, bf16=bf16, storage_bf16=bool(self.bf16_storage) and bf16 == 1)      # (B,3,512,512) NCHW
        self.last_indices = idx.view(B, T)
        return out, logits, lq_feat

    # ------------------------------------------------------------------ host (CPU tensors) path
    def _forward_host(self, x, w, detach_16, code_only, adain):
        enc_feat = {}
        out_list = [self.fuse_encoder_block[f] for f in self.connect_list]
        for i, block in enumerate(self.encoder.blocks):
            x = block(x)
            if i in out_list:
                enc_feat[str(x.shape[-1])] = x.clone()
        lq_feat = x
        pos_emb = self.position_emb.unsqueeze(1).repeat(1, x.shape[0], 1)
        query = self.feat_emb(lq_feat.flatten(2).permute(2, 0, 1))
        for layer in self.ft_layers:
            query = layer(query, query_pos=pos_emb)
        logits = self.idx_pred_layer(query).permute(1, 0, 2)
        if code_only:
            return logits, lq_feat
        _, top_idx = torch.topk(F.softmax(logits, dim=2), 1, dim=2)
        quant = self.quantize.get_codebook_feat(top_idx, shape=[x.shape[0], 16, 16, 256])
        if detach_16:
            quant = quant.detach()
        if adain:
            quant = adaptive_instance_normalization(quant, lq_feat)
        x = quant
        fuse_list = [self.fuse_generator_block[f] for f in self.connect_list]
        for i, block in enumerate(self.generator.blocks):
            x = block(x)
            if i in fuse_list and w > 0:
                f = str(x.shape[-1])
                x = self.fuse_convs_dict[f](enc_feat[f].detach(), x, w)
        return x, logits, lq_feat

    def _forward_graphed(self, x, w, code_only, adain):
        """Capture-once / replay-many execution of _forward_hip on the current stream.  Outputs are copies, so callers may
        keep them across calls.  A graph is re-captured when any packed weight was rebuilt since its capture."""
        with _GRAPH_LOCK:   # static buffers per shape: two threads replaying one graph would race on them
            key = (tuple(x.shape), float(w), bool(code_only), bool(adain), self.precision, bool(self.bf16_storage), self.encoder_precision, self.gemm_precision, bool(self.winograd), bool(self.winograd_encoder), bool(self.winograd_f43), bool(self.winograd_f43_encoder), str(x.device), ops.switches())
            ent = self._graphs.get(key)
            sig = self._param_signature()
            if ent is None or ent['epoch'] != PACK_EPOCH[0] or ent['sig'] != sig:
                static_x = x.float().contiguous().clone()
                for _ in range(2):                      # warm-up: packs weights, sets kernel attributes, primes the allocator
                    self._forward_hip(static_x, w, code_only, adain)
                torch.cuda.synchronize(x.device)
                graph = torch.cuda.CUDAGraph()
                # The zero-initialised scratch words of the range-scale / split-K kernels are kept per (device, stream).  Created inside the
                # capture they would be two fill launches of every replay: create them for the capture stream first, outside the capture.
                cap = _CAPTURE_STREAMS.get(str(x.device))
                if cap is None:
                    cap = _CAPTURE_STREAMS[str(x.device)] = torch.cuda.Stream(device=x.device)
                with torch.cuda.stream(cap):
                    ops._act_cells(x.device, 4 * max(int(x.shape[0]), 1))
                    ops._counters(x.device, 4096)
                cap.synchronize()
                with torch.cuda.graph(graph, stream=cap):
                    outs = self._forward_hip(static_x, w, code_only, adain)
                ent = {'graph': graph, 'x': static_x, 'outs': outs, 'epoch': PACK_EPOCH[0], 'sig': sig, 'idx': getattr(self, 'last_indices', None)}
                self._graphs.pop(key, None)
                while len(self._graphs) >= 4:           # oldest first: a graph pins the activations of its shape
                    self._graphs.pop(next(iter(self._graphs)))
                self._graphs[key] = ent
            ent['x'].copy_(x)
            ent['graph'].replay()
            if ent['idx'] is not None:
                self.last_indices = ent['idx'].clone()   # (a fresh tensor per call, as the eager path returns: the next replay overwrites the graph's own)
            return tuple(o.clone() for o in ent['outs'])

    def _param_signature(self):
        """(sum of versions, sum of storage addresses, sum of object ids) over the parameters and buffers CURRENTLY registered in the
        module tree: changes with every versioned in-place update, every re-allocation and every replaced Parameter object
        (`m.weight = nn.Parameter(...)`, parametrize, an EMA swap).  The list of modules is cached (rebuilt by load_state_dict / _apply /
        invalidate_packed_weights, and here when a sub-module was replaced: the ids of all children are part of the walk); their
        `_parameters` / `_buffers` dictionaries are read on every call (~0.15 ms)."""
        for _ in range(2):
            mods = self.__dict__.get('_sig_modules')
            if mods is None:
                mods = list(self.modules())
                self.__dict__['_sig_modules'] = mods
                self.__dict__['_sig_children'] = sum(id(c) for m in mods for c in m._modules.values() if c is not None)
            ver = ptr = ident = kids = 0
            for m in mods:
                for t in m._parameters.values():
                    if t is not None:
                        ver += ops.tensor_version(t) or 0
                        ptr += t.data_ptr()
                        ident += id(t)
                for t in m._buffers.values():
                    if t is not None:
                        ver += ops.tensor_version(t) or 0
                        ptr += t.data_ptr()
                        ident += id(t)
                for c in m._modules.values():
                    if c is not None:
                        kids += id(c)
            if kids == self.__dict__['_sig_children']:
                break
            self.__dict__.pop('_sig_modules', None)   # a sub-module was replaced or added: walk the new tree
        return (ver, ptr, ident)

    def load_state_dict(self, *args, **kwargs):
        self._graphs.clear()                       # captured graphs point at the old packed weights
        self.__dict__.pop('_sig_modules', None)
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):         # .to() / .cuda() / .float(): parameters move, graphs are stale
        if getattr(self, '_graphs', None):
            self._graphs.clear()
        self.__dict__.pop('_sig_modules', None)
        return super()._apply(fn, *args, **kwargs)

    def invalidate_packed_weights(self):
        self._graphs.clear()
        self.__dict__.pop('_sig_modules', None)
        super().invalidate_packed_weights()

    def forward(self, x, w=0, detach_16=True, code_only=False, adain=False):
        if x.is_cuda:
            ops.L.ensure_device(x.device)   # kernel attributes on the tensor's device, before (never inside) a capture
            with torch.no_grad(), torch.cuda.device(x.device):
                graphed = self.use_hip_graphs is True or (self.use_hip_graphs == 'auto' and x.shape[0] <= self.graph_max_batch)
                if graphed and ops.PROFILE is None and not torch.cuda.is_current_stream_capturing():
                    return self._forward_graphed(x, w, code_only, adain)
                return self._forward_hip(x, w, code_only, adain)
        return self._forward_host(x, w, detach_16, code_only, adain)
