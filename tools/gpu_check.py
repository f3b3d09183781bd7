"""On-GPU diagnostic harness: `python tools/gpu_check.py <group> [...]` (groups run in-process, one report line per
check, never stops at the first failure).  `tools/gpu_run_all.sh` runs every group in its own process so that a
faulting kernel cannot take the other groups down.  Checker only -- imports oracle/ like the tests do.
"""
import json
import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from codeformer_amd import ops  # noqa: E402
from codeformer_amd.ops import (EPI_GELU, EPI_NONE, EPI_RESIDUAL, EPI_SFT, PRO_AFFINE, PRO_AFFINE_SWISH, PRO_LEAKY,  # noqa: E402
                                PRO_NONE)
from oracle import codeformer_oracle as O  # noqa: E402
from oracle.synth import seeded_input, seeded_randn, synth_state_dict  # noqa: E402

DEV = 'cuda'
RESULTS = []


def rnd(shape, seed, scale=1.0):
    return seeded_randn(shape, seed) * scale


def report(name, got, ref, atol, rtol=0.0):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    if got.shape != ref.shape:
        print(f'[FAIL] {name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}', flush=True)
        RESULTS.append((name, False, float('inf')))
        return False
    diff = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    ok = bool((diff <= tol).all()) and bool(torch.isfinite(got).all())
    mx = float(diff.max())
    msg = f'[{"ok" if ok else "FAIL"}] {name}: max|d|={mx:.3e} mean|d|={float(diff.mean()):.3e} ref_absmax={float(ref.abs().max()):.3e}'
    if not ok:
        bad = (diff > tol).nonzero()
        msg += f' nbad={bad.shape[0]}/{diff.numel()} first_bad={bad[0].tolist() if bad.shape[0] else None}'
        if bad.shape[0]:
            i = tuple(bad[0].tolist())
            msg += f' got={float(got[i]):.6f} ref={float(ref[i]):.6f}'
    print(msg, flush=True)
    RESULTS.append((name, ok, mx))
    return ok


def run(name, fn):
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        print(f'[FAIL] {name}: EXCEPTION {type(e).__name__}: {e}', flush=True)
        traceback.print_exc()
        RESULTS.append((name, False, float('nan')))
    torch.cuda.synchronize()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------------------------------
def g_basic():
    def t_transpose():
        x = rnd((2, 48, 20, 24), 1).to(DEV)
        report('to_nhwc', ops.to_nhwc(x), nhwc(x), 0)
        report('to_nchw', ops.to_nchw(nhwc(x)), x, 0)
    run('transpose', t_transpose)

    def t_pack():
        w = rnd((70, 40, 3, 3), 2).to(DEV)
        pw = ops.pack_weight(w, None)
        ref = torch.zeros(9, 48 // 16, 128, 16, device=DEV)
        wp = F.pad(w, (0, 0, 0, 0, 0, 8, 0, 58))  # cin->48, cout->128
        ref = wp.permute(2, 3, 1, 0).reshape(9, 3, 16, 128).permute(0, 1, 3, 2).contiguous()
        report('pack_weight_3x3', pw.w.view(9, 3, 128, 16), ref, 0)
        w2 = rnd((64, 32), 3).to(DEV)
        pw2 = ops.pack_weight(w2, None)
        ref2 = w2.t().reshape(1, 2, 16, 64).permute(0, 1, 3, 2).contiguous()
        report('pack_weight_lin', pw2.w.view(1, 2, 64, 16), ref2, 0)
    run('pack', t_pack)

    def t_ln():
        x = rnd((512, 512), 4, 2.0) + 0.5
        g, b, pos = rnd((512,), 5), rnd((512,), 6), rnd((256, 512), 7)
        y, yp = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5, pos=pos.to(DEV))
        ref = F.layer_norm(x.double(), (512,), g.double(), b.double(), 1e-5)
        report('layernorm', y, ref, 2e-6, 2e-6)
        report('layernorm+pos', yp, ref + pos.double().repeat(2, 1), 2e-6, 2e-6)
    run('layernorm', t_ln)

    def t_argmax():
        x = rnd((512, 1024), 8)
        x[3, 100] = x[3, 900] = 50.0   # tie -> lowest index
        x[4, 1023] = 60.0
        x[5, 0] = 60.0
        idx = ops.argmax_rows(x.to(DEV))
        report('argmax', idx.double(), x.argmax(-1).double(), 0)
        print('   tie row idx =', int(idx[3]), '(expect 100)')
    run('argmax', t_argmax)

    def t_gather():
        cb = rnd((1024, 256), 9, 1e-3)
        idx = torch.randint(0, 1024, (2 * 256,), generator=torch.Generator().manual_seed(10))
        lq = rnd((2, 256, 16, 16), 11, 0.5) + 0.1
        q = O.get_codebook_feat(idx, cb, [2, 16, 16, 256])
        got = ops.codebook_gather(idx.to(DEV), cb.to(DEV), 2, 256)
        report('gather', nchw(got.view(2, 16, 16, 256)), q, 0)
        got = ops.codebook_gather(idx.to(DEV), cb.to(DEV), 2, 256, lq=nhwc(lq).reshape(2, 256, 256).to(DEV))
        report('gather+adain', nchw(got.view(2, 16, 16, 256)), O.adain(q.double(), lq.double()), 2e-6, 2e-6)
    run('gather', t_gather)

    def t_gn():
        for (C, H, seed) in ((64, 32, 12), (128, 16, 13), (512, 16, 14), (256, 64, 15)):
            x = rnd((2, C, H, H), seed, 1.5) + 0.3
            g, b = rnd((C,), seed + 100), rnd((C,), seed + 200)
            sc, sh = ops.groupnorm_tables([nhwc(x).to(DEV)], g.to(DEV), b.to(DEV))
            y = x.double() * sc.double().cpu().view(2, C, 1, 1) + sh.double().cpu().view(2, C, 1, 1)
            report(f'groupnorm C{C} H{H}', y, F.group_norm(x.double(), 32, g.double(), b.double(), 1e-6), 3e-6, 3e-6)
        x1, x2 = rnd((2, 128, 32, 32), 16) + 1.0, rnd((2, 128, 32, 32), 17, 2.0)
        g, b = rnd((256,), 18), rnd((256,), 19)
        sc, sh = ops.groupnorm_tables([nhwc(x1).to(DEV), nhwc(x2).to(DEV)], g.to(DEV), b.to(DEV))
        xc = torch.cat([x1, x2], 1).double()
        y = xc * sc.double().cpu().view(2, 256, 1, 1) + sh.double().cpu().view(2, 256, 1, 1)
        report('groupnorm concat', y, F.group_norm(xc, 32, g.double(), b.double(), 1e-6), 3e-6, 3e-6)
    run('groupnorm', t_gn)

    def t_gn_epilogue():
        """GroupNorm tables from conv-epilogue statistics == tables from the stand-alone pass == fp64 group_norm."""
        cases = [(64, 64, 32, 3, 1, False), (64, 128, 32, 3, 1, False), (128, 256, 16, 3, 1, False), (64, 64, 64, 3, 2, False),
                 (64, 64, 16, 3, 1, True), (128, 128, 16, 1, 1, False), (512, 512, 16, 3, 1, False), (128, 64, 32, 1, 1, False)]
        for i, (cin, cout, H, k, stride, up) in enumerate(cases):
            x = rnd((2, cin, H, H), 300 + i)
            w = rnd((cout, cin, k, k), 320 + i, 1.0 / (cin * k * k) ** 0.5)
            bias = rnd((cout,), 340 + i, 0.3) + 0.5
            g, b = rnd((cout,), 360 + i), rnd((cout,), 380 + i)
            pw = ops.pack_weight(w.to(DEV), bias.to(DEV), up2x=up)
            y = ops.conv2d(nhwc(x).to(DEV), pw, stride=stride, upsample=up, emit_stats=True)
            assert getattr(y, '_cf_stats', None) is not None, 'conv2d did not attach statistics'
            sc, sh = ops.groupnorm_tables([y], g.to(DEV), b.to(DEV))
            yc = nchw(y).double().cpu()
            got = yc * sc.double().cpu().view(2, cout, 1, 1) + sh.double().cpu().view(2, cout, 1, 1)
            report(f'epilogue-stats GN cin{cin} cout{cout} H{H} k{k} s{stride} up{int(up)} (parts={y._cf_stats.parts})', got,
                   F.group_norm(yc, 32, g.double(), b.double(), 1e-6), 5e-6, 5e-6)
        # concat of two conv outputs: pairs of fine groups are merged by the finalize
        xa, xb = rnd((2, 64, 32, 32), 401), rnd((2, 64, 32, 32), 402)
        pwa = ops.pack_weight(rnd((128, 64, 3, 3), 403, 0.05).to(DEV), rnd((128,), 404).to(DEV))
        pwb = ops.pack_weight(rnd((128, 64, 3, 3), 405, 0.08).to(DEV), rnd((128,), 406).to(DEV))
        ya = ops.conv2d(nhwc(xa).to(DEV), pwa, emit_stats=True)
        yb = ops.conv2d(nhwc(xb).to(DEV), pwb, emit_stats=True)
        g, b = rnd((256,), 407), rnd((256,), 408)
        sc, sh = ops.groupnorm_tables([ya, yb], g.to(DEV), b.to(DEV))
        yc = torch.cat([nchw(ya), nchw(yb)], 1).double().cpu()
        got = yc * sc.double().cpu().view(2, 256, 1, 1) + sh.double().cpu().view(2, 256, 1, 1)
        report('epilogue-stats GN concat (gmerge=2)', got, F.group_norm(yc, 32, g.double(), b.double(), 1e-6), 5e-6, 5e-6)
    run('groupnorm-epilogue', t_gn_epilogue)


# ----------------------------------------------------------------------------------------------------
def conv_case(name, cin, cout, H, *, B=2, k=3, stride=1, upsample=False, c_split=None, prologue=PRO_NONE,
              epilogue=EPI_NONE, in_nchw=False, out_nchw=False, seed=0, atol=2e-5):
    def body():
        x = rnd((B, cin, H, H), seed + 1)
        w = rnd((cout, cin, k, k), seed + 2, 1.0 / (cin * k * k) ** 0.5)
        bias = rnd((cout,), seed + 3, 0.1)
        xd = x.double()
        sc = sh = None
        if prologue in (PRO_AFFINE, PRO_AFFINE_SWISH):
            sc, sh = rnd((B, cin), seed + 4, 0.5) + 1.0, rnd((B, cin), seed + 5, 0.3)
            xd = xd * sc.double().view(B, cin, 1, 1) + sh.double().view(B, cin, 1, 1)
            if prologue == PRO_AFFINE_SWISH:
                xd = xd * torch.sigmoid(xd)
        elif prologue == PRO_LEAKY:
            xd = F.leaky_relu(xd, 0.2)
        if upsample:
            xd = F.interpolate(xd, scale_factor=2.0, mode='nearest')
        if stride == 2:
            xd = F.pad(xd, (0, 1, 0, 1))
            ref = F.conv2d(xd, w.double(), bias.double(), stride=2)
        else:
            ref = F.conv2d(xd, w.double(), bias.double(), padding=k // 2)
        Ho = ref.shape[2]
        res = sft = None
        if epilogue == EPI_RESIDUAL:
            res = rnd((B, cout, Ho, Ho), seed + 6)
            ref = ref + res.double()
        elif epilogue == EPI_SFT:
            res, sft = rnd((B, cout, Ho, Ho), seed + 6), rnd((B, cout, Ho, Ho), seed + 7)
            ref = res.double() + 0.7 * (res.double() * sft.double() + ref)
        elif epilogue == EPI_GELU:
            ref = F.gelu(ref)
        pw = ops.pack_weight(w.to(DEV), bias.to(DEV), up2x=upsample)
        if in_nchw:
            xin, x2 = x.to(DEV), None
        elif c_split:
            xin, x2 = nhwc(x[:, :c_split]).to(DEV), nhwc(x[:, c_split:]).to(DEV)
        else:
            xin, x2 = nhwc(x).to(DEV), None
        got = ops.conv2d(xin, pw, x2=x2, stride=stride, upsample=upsample, prologue=prologue,
                         scale=None if sc is None else sc.to(DEV), shift=None if sh is None else sh.to(DEV),
                         epilogue=epilogue, res=None if res is None else nhwc(res).to(DEV),
                         sft_scale=None if sft is None else nhwc(sft).to(DEV), sft_w=0.7, in_nchw=in_nchw,
                         out_nchw=out_nchw)
        report(name, got if out_nchw else nchw(got), ref, atol, 1e-5)
    run(name, body)


def g_conv():
    conv_case('conv3x3 64->128 @32 (BN128)', 64, 128, 32, seed=10)
    conv_case('conv3x3 128->256 @16 (2 N tiles)', 128, 256, 16, seed=20)
    conv_case('conv3x3 64->64 @32 (BN64,BM256)', 64, 64, 32, seed=30)
    conv_case('conv3x3 64->3 @32 out_nchw (BN32)', 64, 3, 32, out_nchw=True, seed=40)
    conv_case('conv3x3 3->64 @32 in_nchw', 3, 64, 32, in_nchw=True, seed=50)
    conv_case('conv3x3 s2 128->128 @32', 128, 128, 32, stride=2, seed=60)
    conv_case('conv3x3 s2 64->64 @64', 64, 64, 64, stride=2, seed=70)
    conv_case('conv3x3 up 64->64 @16', 64, 64, 16, upsample=True, seed=80)
    conv_case('conv3x3 up 128->128 @16', 128, 128, 16, upsample=True, seed=90)
    conv_case('conv3x3 up 256->256 @32 B=3 (narrow off, 2 N tiles)', 256, 256, 32, upsample=True, B=3, seed=95)
    conv_case('conv3x3 up swish 64->128 @16 (prologue on the source)', 64, 128, 16, upsample=True, prologue=PRO_AFFINE_SWISH, seed=97)
    conv_case('conv3x3 cat 64+64->64 @32', 128, 64, 32, c_split=64, seed=100)
    conv_case('conv3x3 cat 128+128->128 @16', 256, 128, 16, c_split=128, seed=110)
    conv_case('conv3x3 affine+swish 64->128 @32', 64, 128, 32, prologue=PRO_AFFINE_SWISH, seed=120)
    conv_case('conv3x3 affine 512->256 @16', 512, 256, 16, prologue=PRO_AFFINE, seed=130)
    conv_case('conv3x3 leaky 128->128 @16', 128, 128, 16, prologue=PRO_LEAKY, seed=140)
    conv_case('conv3x3 +res 128->128 @16', 128, 128, 16, epilogue=EPI_RESIDUAL, seed=150)
    conv_case('conv3x3 leaky+sft 128->128 @32', 128, 128, 32, prologue=PRO_LEAKY, epilogue=EPI_SFT, seed=160)
    conv_case('conv1x1 64->128 @16', 64, 128, 16, k=1, seed=170)
    conv_case('conv1x1 128->64 @32 (BN64)', 128, 64, 32, k=1, seed=180)
    conv_case('conv1x1 cat 64+64->64 @32', 128, 64, 32, k=1, c_split=64, seed=190)
    conv_case('conv1x1 affine 512->1536 @16', 512, 1536, 16, k=1, prologue=PRO_AFFINE, seed=200)
    conv_case('conv1x1 gelu 512->1024 @16', 512, 1024, 16, k=1, epilogue=EPI_GELU, seed=210)
    conv_case('conv3x3 512->512 @16 (K=4608)', 512, 512, 16, seed=220, B=1)
    conv_case('conv3x3 B=3 128->128 @48 (non-pow2 tiles)', 128, 128, 48, B=3, seed=230)

    def t_linear():
        x, w, b = rnd((512, 256), 1), rnd((512, 256), 2, 1 / 16), rnd((512,), 3, 0.1)
        r = rnd((512, 512), 4)
        pw = ops.pack_weight(w.to(DEV), b.to(DEV))
        report('linear 256->512', ops.linear(x.to(DEV), pw), F.linear(x.double(), w.double(), b.double()), 2e-5, 1e-5)
        report('linear +res', ops.linear(x.to(DEV), pw, epilogue=EPI_RESIDUAL, res=r.to(DEV)),
               F.linear(x.double(), w.double(), b.double()) + r.double(), 2e-5, 1e-5)
    run('linear', t_linear)


def g_attn():
    def t(dh, heads, seed):
        B = 2
        E = dh * heads
        q, k, v = rnd((B * 256, E), seed, 0.7), rnd((B * 256, E), seed + 1, 0.7), rnd((B * 256, E), seed + 2)
        scale = dh ** -0.5
        qkv = torch.cat([q, k, v], 1).to(DEV)
        got = ops.attention(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], B, heads, dh, scale)
        qd = q.double().view(B, 256, heads, dh).transpose(1, 2)
        kd = k.double().view(B, 256, heads, dh).transpose(1, 2)
        vd = v.double().view(B, 256, heads, dh).transpose(1, 2)
        a = torch.softmax(qd @ kd.transpose(-1, -2) * scale, -1) @ vd
        report(f'attention dh={dh} heads={heads}', got, a.transpose(1, 2).reshape(B * 256, E), 5e-6, 1e-5)
    run('attn64', lambda: t(64, 8, 1))
    run('attn512', lambda: t(512, 1, 5))

    def t_sep():  # separate buffers with different leading dims (the MHA call pattern)
        B, E = 1, 512
        qk, v = rnd((256, 1024), 9, 0.5), rnd((256, 512), 10)
        got = ops.attention(qk.to(DEV)[:, :E], qk.to(DEV)[:, E:], v.to(DEV), B, 8, 64, 0.125)
        qd = qk[:, :E].double().view(1, 256, 8, 64).transpose(1, 2)
        kd = qk[:, E:].double().view(1, 256, 8, 64).transpose(1, 2)
        vd = v.double().view(1, 256, 8, 64).transpose(1, 2)
        a = torch.softmax(qd @ kd.transpose(-1, -2) * 0.125, -1) @ vd
        report('attention mixed ld', got, a.transpose(1, 2).reshape(256, 512), 5e-6, 1e-5)
    run('attn_sep', t_sep)


# ----------------------------------------------------------------------------------------------------
def g_blocks():
    from codeformer_amd.archs import codeformer_arch as CA
    from codeformer_amd.archs import vqgan_arch as VA
    gold = np.load(os.path.join(ROOT, 'tests/golden/blocks_seed7.npz'))
    shapes = json.load(open(os.path.join(ROOT, 'tests/golden/blocks_seed7_shapes.json')))
    bsd = synth_state_dict(shapes, 7)

    def sub(p):
        return {k[len(p) + 1:]: v for k, v in bsd.items() if k.startswith(p + '.')}

    def mk(mod, p):
        mod.load_state_dict(sub(p))
        return mod.eval().to(DEV)

    xr, xa = seeded_randn((1, 64, 32, 32), 71), seeded_randn((1, 512, 16, 16), 72)
    xt, pos = seeded_randn((256, 2, 512), 73), seeded_randn((256, 1, 512), 74).repeat(1, 2, 1)
    xe, xd, xs = seeded_randn((1, 128, 32, 32), 75), seeded_randn((1, 128, 32, 32), 76), seeded_randn((1, 64, 32, 32), 77)
    g = {k: torch.from_numpy(gold[k]) for k in gold.files}
    run('ResBlock', lambda: report('ResBlock 64->128 vs reference golden', mk(VA.ResBlock(64, 128), 'res')(xr.to(DEV)), g['out_res'], 2e-5, 1e-5))
    run('AttnBlock', lambda: report('AttnBlock 512 vs reference golden', mk(VA.AttnBlock(512), 'attn')(xa.to(DEV)), g['out_attn'], 2e-5, 1e-5))
    run('TransformerSALayer', lambda: report('TransformerSALayer vs reference golden',
                                             mk(CA.TransformerSALayer(512, 8, 1024, 0.0), 'tl')(xt.to(DEV), query_pos=pos.to(DEV)),
                                             g['out_tl'], 2e-5, 1e-5))
    run('Fuse_sft_block', lambda: report('Fuse_sft_block vs reference golden',
                                         mk(CA.Fuse_sft_block(128, 128), 'fuse')(xe.to(DEV), xd.to(DEV), 0.7), g['out_fuse'], 3e-5, 1e-5))
    run('Downsample', lambda: report('Downsample vs reference golden', mk(VA.Downsample(64), 'down')(xs.to(DEV)), g['out_down'], 2e-5, 1e-5))
    run('Upsample', lambda: report('Upsample vs reference golden', mk(VA.Upsample(64), 'up')(xs.to(DEV)), g['out_up'], 2e-5, 1e-5))

    def t_vq():
        gv = np.load(os.path.join(ROOT, 'tests/golden/vq_seed11.npz'))
        q = VA.VectorQuantizer(1024, 256, 0.25)
        q.embedding.weight.data.copy_(torch.from_numpy(gv['codebook']))
        q = q.to(DEV)
        zq, loss, st = q(torch.from_numpy(gv['z']).to(DEV))
        idx = st['min_encoding_indices'].view(-1).cpu().numpy()
        print(f'   vq idx equal: {(idx == gv["idx"]).mean():.4f} of {idx.size}')
        report('VectorQuantizer idx vs reference golden', torch.from_numpy(idx).double(), torch.from_numpy(gv['idx']).double(), 0)
        report('VectorQuantizer z_q', zq, torch.from_numpy(gv['zq']), 1e-7)
    run('VectorQuantizer', t_vq)


def build_net(codebook_size=1024, connect_list=('32', '64', '128', '256')):
    import codeformer_amd.archs  # noqa: F401
    from codeformer_amd.utils.registry import ARCH_REGISTRY
    torch.manual_seed(0)
    net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=codebook_size, n_head=8, n_layers=9,
                                          connect_list=list(connect_list)).eval()
    return net


def g_net():
    net = build_net().to(DEV)
    gold = np.load(os.path.join(ROOT, 'tests/golden/restoration_seed0_face0.npz'))
    x = seeded_input(1)

    def t_full():
        out, logits, lq = net(x.to(DEV), w=0.5, adain=True)
        torch.cuda.synchronize()
        report('lq_feat vs reference golden', lq, torch.from_numpy(gold['lq_feat']), 1e-4)
        report('logits vs reference golden', logits, torch.from_numpy(gold['logits']), 1e-4)
        idx = logits.argmax(-1).cpu().numpy()
        neq = int((idx != gold['idx']).sum())
        print(f'   code indices: {256 - neq}/256 equal to the reference (min ref gap {gold["gap"].min():.2e}); '
              f'kernel argmax == torch argmax: {bool((net.last_indices.cpu().numpy() == idx).all())}')
        RESULTS.append(('indices exact', neq == 0, neq))
        report('out (w=0.5) vs reference golden', out, torch.from_numpy(gold['out']), 1e-3)
        for w, key in ((0.0, 'out_w0.0_sub'), (1.0, 'out_w1.0_sub')):
            o = net(x.to(DEV), w=w, adain=True)[0]
            report(f'out (w={w}) vs reference golden (4x subsampled)', o[:, :, ::4, ::4], torch.from_numpy(gold[key]), 1e-3)
    run('full forward', t_full)

    def t_batch():
        xb = seeded_input(16)[:4].to(DEV)
        o4 = net(xb, w=0.5, adain=True)
        g1 = np.load(os.path.join(ROOT, 'tests/golden/restoration_seed0_b16_face1.npz'))
        report('batch4 face1 out vs golden (sub)', o4[0][1:2, :, ::4, ::4], torch.from_numpy(g1['out_sub']), 1e-3)
        report('batch4 face1 logits vs golden', o4[1][1:2], torch.from_numpy(g1['logits']), 1e-4)
        o1 = net(xb[1:2].contiguous(), w=0.5, adain=True)
        report('batch-of-4 vs batch-of-1 (bitwise)', o4[0][1:2], o1[0], 0)
        o4b = net(xb, w=0.5, adain=True)
        report('repeat run (bitwise)', o4b[0], o4[0], 0)
    run('batch', t_batch)

    def t_time():
        for B in (1, 4, 16):
            xb = seeded_input(16)[:B].to(DEV).contiguous()
            for _ in range(2):
                net(xb, w=0.5, adain=True)
            torch.cuda.synchronize()
            n = 3
            t0 = time.perf_counter()
            for _ in range(n):
                net(xb, w=0.5, adain=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f'   B={B}: {dt * 1e3:.1f} ms/forward = {B / dt:.1f} faces/s  ({739.99e9 * B / dt / 1e12:.1f} executed TFLOP/s fp32)', flush=True)
    run('timing', t_time)

    def t_graph_time():
        net.use_hip_graphs = True
        for B in (1, 4, 16):
            xb = seeded_input(16)[:B].to(DEV).contiguous()
            for _ in range(2):
                net(xb, w=0.5, adain=True)
            torch.cuda.synchronize()
            n = 5
            t0 = time.perf_counter()
            for _ in range(n):
                net(xb, w=0.5, adain=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f'   hip-graph replay B={B}: {dt * 1e3:.1f} ms/forward = {B / dt:.1f} faces/s', flush=True)
        net.use_hip_graphs = False
    run('graph timing', t_graph_time)


def g_bf16():
    """bf16-MFMA conv path: (1) kernel vs an fp64 conv of the SAME bf16-rounded operands (isolates layout/indexing
    errors from rounding: only accumulation order differs), (2) whole net in precision='bf16' vs the fp32 golden."""
    def bf(x):
        return x.to(torch.bfloat16).to(torch.float32)

    def case(name, cin, cout, H, *, B=2, upsample=False, c_split=None, prologue=PRO_NONE, epilogue=EPI_NONE, seed=0):
        def body():
            x = rnd((B, cin, H, H), seed + 1)
            w = rnd((cout, cin, 3, 3), seed + 2, 1.0 / (cin * 9) ** 0.5)
            bias = rnd((cout,), seed + 3, 0.1)
            xd = x.clone()
            sc = sh = None
            if prologue == PRO_AFFINE_SWISH:
                sc, sh = rnd((B, cin), seed + 4, 0.5) + 1.0, rnd((B, cin), seed + 5, 0.3)
                xd = xd * sc.view(B, cin, 1, 1) + sh.view(B, cin, 1, 1)
                xd = xd * torch.sigmoid(xd)
            elif prologue == PRO_LEAKY:
                xd = F.leaky_relu(xd, 0.2)
            xd = bf(xd).double()
            if upsample:
                xd = F.interpolate(xd, scale_factor=2.0, mode='nearest')
            ref = F.conv2d(xd, bf(w).double(), bias.double(), padding=1)
            Ho = ref.shape[2]
            res = sft = None
            if epilogue == EPI_RESIDUAL:
                res = rnd((B, cout, Ho, Ho), seed + 6)
                ref = ref + res.double()
            elif epilogue == EPI_SFT:
                res, sft = rnd((B, cout, Ho, Ho), seed + 6), rnd((B, cout, Ho, Ho), seed + 7)
                ref = res.double() + 0.7 * (res.double() * sft.double() + ref)
            pw = ops.pack_weight(w.to(DEV), bias.to(DEV), bf16=True, up2x=upsample)
            if c_split:
                xin, x2 = nhwc(x[:, :c_split]).to(DEV), nhwc(x[:, c_split:]).to(DEV)
            else:
                xin, x2 = nhwc(x).to(DEV), None
            got = ops.conv2d(xin, pw, x2=x2, upsample=upsample, prologue=prologue, scale=None if sc is None else sc.to(DEV),
                             shift=None if sh is None else sh.to(DEV), epilogue=epilogue,
                             res=None if res is None else nhwc(res).to(DEV), sft_scale=None if sft is None else nhwc(sft).to(DEV),
                             sft_w=0.7, emit_stats=True)
            # swish runs on fast exp/rcp in this mode and is then rounded to bf16: allow a few bf16 ulps of the activations
            # folded up2x taps are summed in fp32 and THEN rounded to bf16 (the reference here rounds each tap): bf16-ulp level
            tol = 2e-2 if (prologue == PRO_AFFINE_SWISH or upsample) else 2e-4
            report(name, nchw(got), ref, tol, 1e-4)
        run(name, body)

    case('bf16 conv 64->128 @32', 64, 128, 32, seed=10)
    case('bf16 conv 128->64 @32 (BN64)', 128, 64, 32, seed=20)
    case('bf16 conv 512->512 @16 (narrow)', 512, 512, 16, seed=30, B=1)
    case('bf16 conv up 128->128 @16', 128, 128, 16, upsample=True, seed=40)
    case('bf16 conv cat 64+64->64 @32', 128, 64, 32, c_split=64, seed=50)
    case('bf16 conv leaky+sft 128->128 @32', 128, 128, 32, prologue=PRO_LEAKY, epilogue=EPI_SFT, seed=60)
    case('bf16 conv swish+res 256->256 @16', 256, 256, 16, prologue=PRO_AFFINE_SWISH, epilogue=EPI_RESIDUAL, seed=70)

    def t_net():
        net = build_net().to(DEV)
        gold = np.load(os.path.join(ROOT, 'tests/golden/restoration_seed0_face0.npz'))
        x = seeded_input(1).to(DEV)
        net.precision = 'f16x2'
        o32 = net(x, w=0.5, adain=True)     # (16-bit modes share the default mode's encoder: split halves)
        net.precision = 'bf16'
        out, logits, lq = net(x, w=0.5, adain=True)
        report('bf16 mode: logits bitwise equal to the default mode', logits, o32[1], 0)
        neq = int((net.last_indices.cpu().numpy() != gold['idx']).sum())
        RESULTS.append(('bf16 mode indices exact', neq == 0, neq))
        d = (out.cpu().double() - torch.from_numpy(gold['out']).double()).abs()
        ref = torch.from_numpy(gold['out']).double()
        print(f'   bf16 mode out vs fp32 reference golden: max|d|={float(d.max()):.4f} mean|d|={float(d.mean()):.5f} '
              f'rms={float((d ** 2).mean().sqrt()):.5f} ref_std={float(ref.std()):.3f} indices_equal={neq == 0}', flush=True)
        report('bf16 mode out vs fp32 reference golden (bf16 gate)', out, torch.from_numpy(gold['out']), 0.25)
        RESULTS.append(('bf16 mode mean error gate', float(d.mean()) < 0.02, float(d.mean())))
        net.precision = 'fp16'   # IEEE-half operands: same split, same speed, 3 more mantissa bits
        out, logits, lq = net(x, w=0.5, adain=True)
        report('fp16 mode: logits bitwise equal to the default mode', logits, o32[1], 0)
        neq = int((net.last_indices.cpu().numpy() != gold['idx']).sum())
        RESULTS.append(('fp16 mode indices exact', neq == 0, neq))
        d = (out.cpu().double() - torch.from_numpy(gold['out']).double()).abs()
        print(f'   fp16 mode out vs fp32 reference golden: max|d|={float(d.max()):.4f} mean|d|={float(d.mean()):.5f}', flush=True)
        report('fp16 mode out vs fp32 reference golden (fp16 gate)', out, torch.from_numpy(gold['out']), 0.04)
        RESULTS.append(('fp16 mode mean error gate', float(d.mean()) < 0.003, float(d.mean())))
        xb = seeded_input(16).to(DEV)
        for prec in ('fp32', 'bf16', 'fp16'):
            net.precision = prec
            for _ in range(2):
                net(xb, w=0.5, adain=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                net(xb, w=0.5, adain=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print(f'   precision={prec}: B=16 {dt * 1e3:.1f} ms/forward = {16 / dt:.1f} faces/s', flush=True)
    run('bf16 net', t_net)


GROUPS = {'bf16': g_bf16, 'basic': g_basic, 'conv': g_conv, 'attn': g_attn, 'blocks': g_blocks, 'net': g_net}

if __name__ == '__main__':
    names = sys.argv[1:] or list(GROUPS)
    print('device:', torch.cuda.get_device_name(0), flush=True)
    for n in names:
        print(f'===== group {n} =====', flush=True)
        t0 = time.time()
        GROUPS[n]()
        print(f'----- {n} done in {time.time() - t0:.1f}s', flush=True)
    nfail = sum(1 for r in RESULTS if not r[1])
    print(f'SUMMARY {" ".join(names)}: {len(RESULTS) - nfail} ok, {nfail} FAIL')
    sys.exit(1 if nfail else 0)
