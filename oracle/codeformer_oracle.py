"""CPU restatement of the reference CodeFormer forward (fp32, torch CPU ops).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The product path
(codeformer_amd/) never imports this file.

Every arithmetic step of the reference path is executed by PyTorch ATen ops
(the reference has no custom kernels on this path), so the
restatement is written against ``torch.nn.functional`` on CPU tensors, driven
by a plain ``state_dict`` (no nn.Module).  It does NOT import /root/reference,
so it travels to the GPU box.  Each function cites the reference lines it
follows (paths relative to /root/reference).

Pinning: the reference ships no tests/golden vectors, so this
oracle is pinned against outputs of the reference's own arch files executed in
the build container (oracle/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py) and against the known-answer values recorded in
SURVEY.md section 8(c).
"""
import math

import torch
import torch.nn.functional as F

GN_GROUPS = 32
GN_EPS = 1e-6        # basicsr/archs/vqgan_arch.py:14-15
LN_EPS = 1e-5        # nn.LayerNorm default, basicsr/archs/codeformer_arch.py:108-109
ADAIN_EPS = 1e-5     # basicsr/archs/codeformer_arch.py:12

# Channel plan of VQAutoEncoder(512, 64, [1,2,2,4,4,8], 'nearest', 2, [16], cb)
# basicsr/archs/codeformer_arch.py:166
NF = 64
CH_MULT = (1, 2, 2, 4, 4, 8)
RESOLUTION = 512
ATTN_RES = (16,)
EMB_DIM = 256


def encoder_plan():
    """Block list of Encoder.__init__ (basicsr/archs/vqgan_arch.py:229-267).

    Returns a list of (kind, cin, cout) with kind in
    {'conv','res','attn','down','norm'}.
    """
    plan = [('conv', 3, NF)]
    curr = RESOLUTION
    in_mult = (1,) + CH_MULT
    cin = NF
    for i in range(len(CH_MULT)):
        cin = NF * in_mult[i]
        cout = NF * CH_MULT[i]
        for _ in range(2):
            plan.append(('res', cin, cout))
            cin = cout
            if curr in ATTN_RES:
                plan.append(('attn', cin, cin))
        if i != len(CH_MULT) - 1:
            plan.append(('down', cin, cin))
            curr //= 2
    plan += [('res', cin, cin), ('attn', cin, cin), ('res', cin, cin),
             ('norm', cin, cin), ('conv', cin, EMB_DIM)]
    return plan


def generator_plan():
    """Block list of Generator.__init__ (basicsr/archs/vqgan_arch.py:276-316)."""
    cin = NF * CH_MULT[-1]
    curr = RESOLUTION // 2 ** (len(CH_MULT) - 1)
    plan = [('conv', EMB_DIM, cin), ('res', cin, cin), ('attn', cin, cin), ('res', cin, cin)]
    for i in reversed(range(len(CH_MULT))):
        cout = NF * CH_MULT[i]
        for _ in range(2):
            plan.append(('res', cin, cout))
            cin = cout
            if curr in ATTN_RES:
                plan.append(('attn', cin, cin))
        if i != 0:
            plan.append(('up', cin, cin))
            curr *= 2
    plan += [('norm', cin, cin), ('conv', cin, 3)]
    return plan


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def group_norm(x, sd, p):
    # basicsr/archs/vqgan_arch.py:14-15
    return F.group_norm(x, GN_GROUPS, sd[p + '.weight'], sd[p + '.bias'], GN_EPS)


def swish(x):
    # basicsr/archs/vqgan_arch.py:18-20
    return x * torch.sigmoid(x)


def conv(x, sd, p, stride=1, padding=1):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride=stride, padding=padding)


def res_block(x_in, sd, p):
    # basicsr/archs/vqgan_arch.py:153-164
    x = group_norm(x_in, sd, p + '.norm1')
    x = swish(x)
    x = conv(x, sd, p + '.conv1')
    x = group_norm(x, sd, p + '.norm2')
    x = swish(x)
    x = conv(x, sd, p + '.conv2')
    if (p + '.conv_out.weight') in sd:
        x_in = conv(x_in, sd, p + '.conv_out', padding=0)
    return x + x_in


def attn_block(x, sd, p):
    # basicsr/archs/vqgan_arch.py:202-226
    h_ = group_norm(x, sd, p + '.norm')
    q = conv(h_, sd, p + '.q', padding=0)
    k = conv(h_, sd, p + '.k', padding=0)
    v = conv(h_, sd, p + '.v', padding=0)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k)
    w_ = w_ * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    w_ = w_.permute(0, 2, 1)
    h_ = torch.bmm(v, w_).reshape(b, c, h, w)
    h_ = conv(h_, sd, p + '.proj_out', padding=0)
    return x + h_


def downsample(x, sd, p):
    # basicsr/archs/vqgan_arch.py:122-126 -- pad right/bottom only, stride 2
    x = F.pad(x, (0, 1, 0, 1), mode='constant', value=0)
    return conv(x, sd, p + '.conv', stride=2, padding=0)


def upsample(x, sd, p):
    # basicsr/archs/vqgan_arch.py:134-138
    x = F.interpolate(x, scale_factor=2.0, mode='nearest')
    return conv(x, sd, p + '.conv')


def run_block(kind, x, sd, p):
    if kind == 'conv':
        return conv(x, sd, p)
    if kind == 'res':
        return res_block(x, sd, p)
    if kind == 'attn':
        return attn_block(x, sd, p)
    if kind == 'down':
        return downsample(x, sd, p)
    if kind == 'up':
        return upsample(x, sd, p)
    if kind == 'norm':
        return group_norm(x, sd, p)
    raise ValueError(kind)


def calc_mean_std(feat, eps=ADAIN_EPS):
    # basicsr/archs/codeformer_arch.py:12-26 (unbiased variance)
    b, c = feat.shape[:2]
    var = feat.view(b, c, -1).var(dim=2) + eps
    std = var.sqrt().view(b, c, 1, 1)
    mean = feat.view(b, c, -1).mean(dim=2).view(b, c, 1, 1)
    return mean, std


def adain(content, style):
    # basicsr/archs/codeformer_arch.py:29-43
    size = content.size()
    s_mean, s_std = calc_mean_std(style)
    c_mean, c_std = calc_mean_std(content)
    normalized = (content - c_mean.expand(size)) / c_std.expand(size)
    return normalized * s_std.expand(size) + s_mean.expand(size)


def layer_norm(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], LN_EPS)


def mha(q_in, k_in, v_in, sd, p, n_head):
    """nn.MultiheadAttention forward, need_weights=True path, no masks, dropout 0.

    Reference call site basicsr/archs/codeformer_arch.py:126; arithmetic in
    torch/nn/functional.py multi_head_attention_forward (q != k is v-distinct
    => in_proj_weight.chunk(3); q scaled by sqrt(1/head_dim) BEFORE q.k^T;
    softmax over keys).  Inputs are (L, B, E) sequence-first.
    """
    L, B, E = q_in.shape
    hd = E // n_head
    w_q, w_k, w_v = sd[p + '.in_proj_weight'].chunk(3)
    b_q, b_k, b_v = sd[p + '.in_proj_bias'].chunk(3)
    q = F.linear(q_in, w_q, b_q)
    k = F.linear(k_in, w_k, b_k)
    v = F.linear(v_in, w_v, b_v)
    q = q.view(L, B * n_head, hd).transpose(0, 1)
    k = k.view(L, B * n_head, hd).transpose(0, 1)
    v = v.view(L, B * n_head, hd).transpose(0, 1)
    q = q * math.sqrt(1.0 / float(hd))
    attn = torch.bmm(q, k.transpose(-2, -1))
    attn = F.softmax(attn, dim=-1)
    out = torch.bmm(attn, v)
    out = out.transpose(0, 1).contiguous().view(L * B, E)
    out = F.linear(out, sd[p + '.out_proj.weight'], sd[p + '.out_proj.bias'])
    return out.view(L, B, E)


def transformer_layer(tgt, pos, sd, p, n_head):
    # basicsr/archs/codeformer_arch.py:118-134
    t2 = layer_norm(tgt, sd, p + '.norm1')
    q = k = t2 + pos
    t2 = mha(q, k, t2, sd, p + '.self_attn', n_head)
    tgt = tgt + t2
    t2 = layer_norm(tgt, sd, p + '.norm2')
    t2 = F.linear(F.gelu(F.linear(t2, sd[p + '.linear1.weight'], sd[p + '.linear1.bias'])),
                  sd[p + '.linear2.weight'], sd[p + '.linear2.bias'])
    return tgt + t2


def fuse_sft(enc, dec, w, sd, p):
    # basicsr/archs/codeformer_arch.py:151-157
    e = res_block(torch.cat([enc, dec], dim=1), sd, p + '.encode_enc')
    scale = conv(F.leaky_relu(conv(e, sd, p + '.scale.0'), 0.2), sd, p + '.scale.2')
    shift = conv(F.leaky_relu(conv(e, sd, p + '.shift.0'), 0.2), sd, p + '.shift.2')
    return dec + w * (dec * scale + shift)


def get_codebook_feat(indices, codebook, shape):
    # basicsr/archs/vqgan_arch.py:72-84: one-hot x codebook == exact row gather
    z_q = codebook[indices.view(-1)]
    return z_q.view(shape).permute(0, 3, 1, 2).contiguous()


def vq_nearest(z, codebook):
    """VectorQuantizer.forward indices + z_q (basicsr/archs/vqgan_arch.py:33-70)."""
    zp = z.permute(0, 2, 3, 1).contiguous()
    zf = zp.view(-1, codebook.shape[1])
    d = (zf ** 2).sum(dim=1, keepdim=True) + (codebook ** 2).sum(1) - 2 * torch.matmul(zf, codebook.t())
    idx = torch.argmin(d, dim=1)
    z_q = codebook[idx].view(zp.shape).permute(0, 3, 1, 2).contiguous()
    return z_q, idx, d


def vq_forward(z, codebook, beta):
    """VectorQuantizer.forward with its statistics (basicsr/archs/vqgan_arch.py:33-70): distances, mean_distance (:42),
    argmin (:44), one-hot min_encodings (:49-50), z_q = one-hot x codebook (:53), loss = mse + beta * mse (:55),
    straight-through z_q (:57), perplexity = exp(-sum(e_mean * log(e_mean + 1e-10))) (:60-61)."""
    zp = z.permute(0, 2, 3, 1).contiguous()
    zf = zp.view(-1, codebook.shape[1])
    d = (zf ** 2).sum(dim=1, keepdim=True) + (codebook ** 2).sum(1) - 2 * torch.matmul(zf, codebook.t())
    mean_distance = torch.mean(d)
    idx = torch.argmin(d, dim=1).unsqueeze(1)
    enc = torch.zeros(idx.shape[0], codebook.shape[0]).to(z)
    enc.scatter_(1, idx, 1)
    z_q = torch.matmul(enc, codebook).view(zp.shape)
    loss = torch.mean((z_q - zp) ** 2) + beta * torch.mean((z_q - zp) ** 2)
    z_q = zp + (z_q - zp)
    e_mean = torch.mean(enc, dim=0)
    perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
    return {'z_q': z_q.permute(0, 3, 1, 2).contiguous(), 'loss': loss, 'perplexity': perplexity, 'min_encodings': enc,
            'min_encoding_indices': idx, 'mean_distance': mean_distance}


def inpaint_composite(x, y):
    """Inpainting composite (inference_inpainting.py:68-74): mask = pixels whose three normalised input channels sum to
    exactly 3 (pure white brush); result = (1 - mask) * x + mask * y, evaluated in fp32 in that order."""
    mask = torch.zeros(x.shape[0], 1, x.shape[2], x.shape[3], dtype=x.dtype)
    mask[torch.sum(x, dim=1, keepdim=True) == 3] = 1.0
    return (1 - mask) * x + mask * y


FUSE_ENC_BLOCK = {'512': 2, '256': 5, '128': 8, '64': 11, '32': 14, '16': 18}   # codeformer_arch.py:203
FUSE_GEN_BLOCK = {'16': 6, '32': 9, '64': 12, '128': 15, '256': 18, '512': 21}  # codeformer_arch.py:205


@torch.no_grad()
def encoder_forward(x, sd, connect_list):
    # basicsr/archs/codeformer_arch.py:225-230
    taps = {}
    out_list = [FUSE_ENC_BLOCK[f] for f in connect_list]
    for i, (kind, _, _) in enumerate(encoder_plan()):
        x = run_block(kind, x, sd, f'encoder.blocks.{i}')
        if i in out_list:
            taps[str(x.shape[-1])] = x.clone()
    return x, taps


@torch.no_grad()
def transformer_forward(lq_feat, sd, n_head=8, n_layers=9):
    # basicsr/archs/codeformer_arch.py:235-245
    B = lq_feat.shape[0]
    pos = sd['position_emb'].unsqueeze(1).repeat(1, B, 1)
    q = F.linear(lq_feat.flatten(2).permute(2, 0, 1), sd['feat_emb.weight'], sd['feat_emb.bias'])
    for l in range(n_layers):
        q = transformer_layer(q, pos, sd, f'ft_layers.{l}', n_head)
    logits = F.linear(layer_norm(q, sd, 'idx_pred_layer.0'), sd['idx_pred_layer.1.weight'])
    return logits.permute(1, 0, 2)


@torch.no_grad()
def generator_forward(x, sd, taps, w, connect_list):
    # basicsr/archs/codeformer_arch.py:269-277
    fuse_list = [FUSE_GEN_BLOCK[f] for f in connect_list]
    for i, (kind, _, _) in enumerate(generator_plan()):
        x = run_block(kind, x, sd, f'generator.blocks.{i}')
        if i in fuse_list:
            f = str(x.shape[-1])
            if w > 0:
                x = fuse_sft(taps[f], x, w, sd, f'fuse_convs_dict.{f}')
    return x


@torch.no_grad()
def codeformer_forward(x, sd, w=0.0, adain_flag=False, code_only=False,
                       connect_list=('32', '64', '128', '256'), n_head=8, n_layers=9,
                       return_idx=False):
    """CodeFormer.forward (basicsr/archs/codeformer_arch.py:223-280).

    x: (B,3,512,512) fp32 CPU tensor.  Returns (out, logits, lq_feat) like the
    reference; with return_idx also the (B,256) int64 code indices.
    """
    x = x.float()
    lq_feat, taps = encoder_forward(x, sd, connect_list)
    logits = transformer_forward(lq_feat, sd, n_head, n_layers)
    if code_only:
        return logits, lq_feat
    soft = F.softmax(logits, dim=2)
    _, top_idx = torch.topk(soft, 1, dim=2)
    B = x.shape[0]
    quant = get_codebook_feat(top_idx, sd['quantize.embedding.weight'], [B, 16, 16, 256])
    if adain_flag:
        quant = adain(quant, lq_feat)
    out = generator_forward(quant, sd, taps, w, connect_list)
    if return_idx:
        return out, logits, lq_feat, top_idx.view(B, -1)
    return out, logits, lq_feat


def tensor2img_u8(t, min_max=(-1.0, 1.0)):
    """tensor2img for one (3,H,W) RGB tensor -> uint8 HWC BGR
    (basicsr/utils/img_util.py:38-94: clamp, (x-min)/(max-min), *255, round-half-even)."""
    t = t.float().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    img = t.numpy().transpose(1, 2, 0)[:, :, ::-1]
    return (img * 255.0).round().astype('uint8')


def fused_bias_act(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """fused_leaky_relu restatement
    (basicsr/ops/fused_act/src/fused_bias_act_kernel.cu:20-50, act=3 grad=0)."""
    shape = [1, -1] + [1] * (x.dim() - 2)
    return F.leaky_relu(x + bias.view(*shape), negative_slope) * scale


def fused_bias_act_modes(x, bias, ref, act, grad, alpha, scale):
    """Every mode of the op (fused_bias_act_kernel.cu:26-47): x += bias[channel]; act*10+grad: 10/11 y = x, 12/32 y = 0,
    30 y = x > 0 ? x : x*alpha, 31 y = ref > 0 ? x : x*alpha; out = y*scale.  Arithmetic in x's dtype, as scalar_t."""
    if bias is not None:
        x = x + bias.view(*([1, -1] + [1] * (x.dim() - 2)))
    mode = act * 10 + grad
    if mode in (12, 32):
        y = torch.zeros_like(x)
    elif mode == 30:
        y = torch.where(x > 0, x, x * alpha)
    elif mode == 31:
        y = torch.where(ref > 0, x, x * alpha)
    else:
        y = x
    return y * scale


def upfirdn2d(inp, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d restatement following upfirdn2d_native
    (basicsr/ops/upfirdn2d/upfirdn2d.py:156-186); input (N,C,H,W), square up/down,
    pad = (pad0, pad1) applied to both axes."""
    n, c, in_h, in_w = inp.shape
    kh, kw = kernel.shape
    x = inp.reshape(-1, in_h, 1, in_w, 1)
    x = F.pad(x, [0, up - 1, 0, 0, 0, up - 1])
    x = x.reshape(-1, in_h * up, in_w * up)
    p0, p1 = pad
    x = F.pad(x, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    x = x[:, max(-p0, 0):x.shape[1] - max(-p1, 0), max(-p0, 0):x.shape[2] - max(-p1, 0)]
    x = x.unsqueeze(1)
    wk = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw)
    x = F.conv2d(x, wk)
    x = x[:, :, ::down, ::down]
    return x.reshape(n, c, x.shape[2], x.shape[3])
