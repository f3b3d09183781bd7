"""RRDBNet / RealESRGANer row (reference: basicsr/archs/rrdbnet_arch.py, basicsr/utils/realesrgan_utils.py).

CPU (`-m "not gpu"`): the oracle restatement, this repo's module (init order, state_dict keys, host forward) and the
RealESRGANer tensor stages against goldens produced by the REFERENCE's own files (oracle/make_golden_rrdbnet.py).
GPU (`-m gpu`): the HIP path (dense-block slices, leaky / axpy / axpy2 epilogues, masked edge tiles, folded upsample,
pixel-unshuffle) against the same goldens and the oracle.

Tolerances (fp32 everywhere): oracle / host forward vs reference 2e-6 (same ATen kernels, other threading);
HIP vs reference 2e-5 on the 2-block nets (outputs ~0.015 mean, 0.2 max) and 2e-4 on the 23-block net (207 chained
convolutions, outputs up to 1.7): different accumulation order only.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _digests():
    with open(os.path.join(GOLD, 'rrdbnet_digests.json')) as f:
        return json.load(f)


def _sd_hash(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


def _build(case):
    from basicsr.archs.rrdbnet_arch import RRDBNet
    torch.manual_seed(case['seed'])
    return RRDBNet(3, 3, scale=case['scale'], num_feat=64, num_block=case['num_block'], num_grow_ch=32).eval()


def _input(shape, seed):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed))


CASES = ['x2_small', 'x4_small', 'x2_full']


@pytest.mark.parametrize('name', CASES)
def test_rrdbnet_oracle_matches_reference_goldens(name):
    """Pins oracle/rrdbnet_oracle.py AND the module's init order / keys / host forward to the reference's outputs."""
    from oracle import rrdbnet_oracle as RO
    case = _digests()[name]
    net = _build(case)
    assert len(net.state_dict()) == case['n_keys']
    assert _sd_hash(net.state_dict()) == case['state_dict'], 'seeded init differs from the reference constructor'
    gold = torch.from_numpy(np.load(os.path.join(GOLD, f'rrdbnet_{name}.npz'))['out'])
    x = _input(case['shape'], case['in_seed'])
    y = RO.rrdbnet_forward(net.state_dict(), x, case['scale'])
    assert tuple(y.shape) == tuple(gold.shape) == tuple(case['out_shape'])
    assert float((y - gold).abs().max()) <= 2e-6
    with torch.no_grad():
        assert float((net(x) - gold).abs().max()) <= 2e-6


def test_rrdbnet_registry_and_errors():
    import basicsr  # noqa: F401
    from basicsr.utils.registry import ARCH_REGISTRY
    from oracle import rrdbnet_oracle as RO
    cls = ARCH_REGISTRY.get('RRDBNet')
    net = cls(3, 3, scale=1, num_feat=32, num_block=1, num_grow_ch=16).eval()
    x = _input((1, 3, 16, 24), 1)
    with torch.no_grad():
        y = net(x)
    assert tuple(y.shape) == (1, 3, 16, 24)       # scale 1: unshuffle by 4, upsample by 4
    assert float((y - RO.rrdbnet_forward(net.state_dict(), x, 1)).abs().max()) <= 2e-6
    with pytest.raises(ValueError):
        net(torch.rand(1, 3, 18, 24))            # not divisible by the unshuffle factor (arch_util.py:202 asserts)


def test_realesrganer_tensor_stages_match_reference():
    """pre_process (reflect pre-pad + mod-pad) -> tile_process / process -> post_process, on CPU, vs the reference class."""
    from basicsr.utils.realesrgan_utils import RealESRGANer
    d = _digests()
    gold = np.load(os.path.join(GOLD, 'rrdbnet_esrganer.npz'))
    net = _build(d['x2_small'])
    e = d['esrganer']
    img = _input(tuple(e['img_shape']), e['in_seed']).numpy()
    up = RealESRGANer(2, None, model=net, tile=e['tile'], tile_pad=e['tile_pad'], pre_pad=e['pre_pad'], device='cpu')
    with torch.no_grad():
        up.pre_process(img)
        assert list(up.img.shape) == e['padded_shape']
        up.tile_process()
        tiled = up.post_process().clone()
        up.pre_process(img)
        up.process()
        whole = up.post_process().clone()
    assert list(tiled.shape) == e['out_shape']
    assert float((tiled - torch.from_numpy(gold['tiled'])).abs().max()) <= 2e-6
    assert float((whole - torch.from_numpy(gold['whole'])).abs().max()) <= 2e-6


def test_realesrganer_enhance_modes_cpu():
    """Image-level contract: BGR uint8 / gray / BGRA / 16-bit in, same mode out at x2 (realesrgan_utils.py:179-262)."""
    from basicsr.utils.realesrgan_utils import RealESRGANer
    from oracle import rrdbnet_oracle as RO
    net = _build(_digests()['x2_small'])
    up = RealESRGANer(2, None, model=net, tile=0, pre_pad=0, device='cpu')
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (18, 22, 3), dtype=np.uint8)
    out, mode = up.enhance(bgr)
    assert mode == 'RGB' and out.shape == (36, 44, 3) and out.dtype == np.uint8
    x = torch.from_numpy(np.ascontiguousarray(bgr[:, :, ::-1].transpose(2, 0, 1))).float()[None] / 255
    ref = RO.rrdbnet_forward(net.state_dict(), x, 2)[0].clamp(0, 1).numpy()[::-1].transpose(1, 2, 0)
    assert np.abs(out.astype(int) - np.round(ref * 255).astype(int)).max() <= 1
    out, mode = up.enhance(bgr[:, :, 0])
    assert mode == 'L' and out.shape == (36, 44)
    out, mode = up.enhance(np.concatenate([bgr, bgr[:, :, :1]], axis=2))
    assert mode == 'RGBA' and out.shape == (36, 44, 4)
    out, mode = up.enhance((bgr.astype(np.uint16) * 257))
    assert out.dtype == np.uint16 and out.shape == (36, 44, 3)
    with pytest.raises(NotImplementedError):
        up.enhance(bgr, outscale=3)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_rrdbnet_hip_matches_reference_goldens(name):
    from codeformer_amd import lib
    lib.load()
    case = _digests()[name]
    net = _build(case).cuda()
    gold = torch.from_numpy(np.load(os.path.join(GOLD, f'rrdbnet_{name}.npz'))['out'])
    y = net(_input(case['shape'], case['in_seed']).cuda()).cpu()
    tol = 2e-4 if name == 'x2_full' else 2e-5
    err = float((y - gold).abs().max())
    print(f'rrdbnet {name}: max|d| {err:.3e} (out max {float(gold.abs().max()):.3f})')
    assert tuple(y.shape) == tuple(gold.shape) and err <= tol


@pytest.mark.gpu
def test_rrdbnet_hip_sizes_batches_and_blocks():
    """Edge-tile masking and batch handling: odd sizes, sizes on the tile grid, batch 3; single blocks through their own
    forward(); bitwise batch invariance (tile choice never depends on the batch)."""
    from codeformer_amd import lib
    from oracle import rrdbnet_oracle as RO
    lib.load()
    d = _digests()
    net = _build(d['x2_small'])
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    for shape, seed in (((1, 3, 32, 32), 1), ((1, 3, 34, 66), 2), ((3, 3, 50, 18), 3), ((2, 3, 2, 2), 4)):
        x = _input(shape, seed)
        y = net(x.cuda()).cpu()
        ref = RO.rrdbnet_forward(sd, x, 2)
        assert tuple(y.shape) == tuple(ref.shape)
        assert float((y - ref).abs().max()) <= 2e-5, shape
    x = _input((3, 3, 50, 18), 3).cuda()
    assert torch.equal(net(x)[1:2], net(x[1:2]))
    assert torch.equal(net(x), net(x))
    blk = net.body[0]
    f = torch.randn(2, 64, 21, 37, generator=torch.Generator().manual_seed(9)) * 0.5
    assert float((blk(f.cuda()).cpu() - RO.rrdb(sd, 'body.0', f)).abs().max()) <= 2e-5
    assert float((blk.rdb2(f.cuda()).cpu() - RO.dense_block(sd, 'body.0.rdb2', f)).abs().max()) <= 2e-5


@pytest.mark.gpu
def test_pixel_unshuffle_and_strided_conv_ops():
    """cf_pixel_unshuffle_nhwc exactly; cf_conv2d on channel slices (ld_in1 / ld_out) + leaky / axpy / axpy2 epilogues
    against fp64 references of the same operands (2e-5 + 1e-5*|ref|, as for the other instantiations)."""
    from codeformer_amd import lib, ops
    from oracle import rrdbnet_oracle as RO
    import torch.nn.functional as F
    lib.load()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, 12, 20, generator=g)
    for s in (1, 2, 4):
        y = ops.pixel_unshuffle_nhwc(x.cuda(), s).cpu()
        ref = RO.pixel_unshuffle(x, s) if s > 1 else x
        assert y.shape[3] % 16 == 0 and torch.equal(y[..., :ref.shape[1]], ref.permute(0, 2, 3, 1))
        assert not bool(y[..., ref.shape[1]:].any())    # zero channel padding (none for s = 4: 48 channels)
    B, H, W = 2, 19, 35
    xin = torch.randn(B, H, W, 64, generator=g)
    buf = torch.randn(B, H, W, 128, generator=g)
    res = torch.randn(B, H, W, 64, generator=g)
    res2 = torch.randn(B, H, W, 64, generator=g)
    w32 = torch.randn(32, 128, 3, 3, generator=g) * 0.05
    w64 = torch.randn(64, 192, 3, 3, generator=g) * 0.05
    b32, b64 = torch.randn(32, generator=g), torch.randn(64, generator=g)

    def ref_conv(inp, w, b):
        return F.conv2d(inp.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)

    def close(a, r):
        return bool(((a.double() - r).abs() <= 2e-5 + 1e-5 * r.abs()).all())

    dbuf = buf.cuda()
    before = dbuf.clone()
    out = ops.conv2d(xin.cuda(), ops.pack_weight(w32.cuda(), b32.cuda()), x2=dbuf[..., :64], epilogue=ops.EPI_LEAKY,
                     out=dbuf[..., 64:96])
    r = F.leaky_relu(ref_conv(torch.cat([xin, buf[..., :64]], 3), w32, b32), 0.2)
    assert out.data_ptr() == dbuf[..., 64:96].data_ptr() and close(dbuf[..., 64:96].cpu(), r)
    assert torch.equal(dbuf[..., :64], before[..., :64]) and torch.equal(dbuf[..., 96:], before[..., 96:])  # neighbours untouched
    pw = ops.pack_weight(w64.cuda(), b64.cuda())
    c = ref_conv(torch.cat([xin, buf], 3), w64, b64)
    y1 = ops.conv2d(xin.cuda(), pw, x2=buf.cuda(), epilogue=ops.EPI_AXPY, res=res.cuda(), sft_w=0.2)
    assert close(y1.cpu(), c * 0.2 + res.double())
    y2 = ops.conv2d(xin.cuda(), pw, x2=buf.cuda(), epilogue=ops.EPI_AXPY2, res=res.cuda(), sft_scale=res2.cuda(), sft_w=0.2)
    assert close(y2.cpu(), (c * 0.2 + res.double()) * 0.2 + res2.double())
    with pytest.raises(ValueError):
        ops.conv2d(xin.cuda(), pw, x2=buf.cuda()[:, :, ::2])       # not a channel slice
    with pytest.raises(RuntimeError):                               # strided output + GroupNorm statistics: refused by the C ABI
        ops.conv2d(buf.cuda()[..., :64], ops.pack_weight(w64[:, :64].contiguous().cuda(), b64.cuda()), emit_stats=True)


@pytest.mark.gpu
def test_realesrganer_hip_tiled_equals_reference():
    from basicsr.utils.realesrgan_utils import RealESRGANer
    from codeformer_amd import lib
    lib.load()
    d = _digests()
    gold = np.load(os.path.join(GOLD, 'rrdbnet_esrganer.npz'))
    e = d['esrganer']
    img = _input(tuple(e['img_shape']), e['in_seed']).numpy()
    up = RealESRGANer(2, None, model=_build(d['x2_small']), tile=e['tile'], tile_pad=e['tile_pad'], pre_pad=e['pre_pad'],
                      device='cuda')
    up.pre_process(img)
    up.tile_process()
    tiled = up.post_process().cpu()
    assert float((tiled - torch.from_numpy(gold['tiled'])).abs().max()) <= 2e-5
    out, mode = up.enhance((img * 255).round().astype(np.uint8)[:, :, ::-1])
    assert mode == 'RGB' and out.shape == (74, 90, 3)


@pytest.mark.gpu
def test_rrdbnet_hip_fp16_operands():
    """`RRDBNet.half()` = IEEE-half MFMA operands, fp32 accumulate / tensors (the reference's GPU default is fp16 storage).
    Kernel level: against fp64 convolutions of the SAME half-rounded operands (2e-4 + 1e-4*|ref|: only the accumulation order
    differs).  Network level: against the fp32 reference goldens -- gate max|d| <= 2e-4 and mean|d| <= 3e-5 on the 2-block
    nets (outputs up to 0.09; measured 6e-5 / 1e-5), max|d| <= 1e-2 / mean|d| <= 1e-3 on the 23-block net (outputs up to 1.3;
    measured 2.0e-3 / 3.2e-4)."""
    from codeformer_amd import lib, ops
    import torch.nn.functional as F
    lib.load()
    g = torch.Generator().manual_seed(33)
    B, H, W = 2, 21, 40
    xin = torch.randn(B, H, W, 64, generator=g)
    buf = torch.randn(B, H, W, 128, generator=g)
    res = torch.randn(B, H, W, 64, generator=g)
    h = lambda t: t.half().double()   # noqa: E731

    def ref_conv(inp, w, b, up=False):
        x = h(inp).permute(0, 3, 1, 2)
        if up:
            x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        return F.conv2d(x, h(w), b.double(), padding=1).permute(0, 2, 3, 1)

    def close(a, r):
        return bool(((a.double() - r).abs() <= 2e-4 + 1e-4 * r.abs()).all())

    w32, b32 = torch.randn(32, 128, 3, 3, generator=g) * 0.05, torch.randn(32, generator=g)
    dbuf = buf.cuda()
    ops.conv2d(xin.cuda(), ops.pack_weight(w32.cuda(), b32.cuda(), f16=True), x2=dbuf[..., :64], epilogue=ops.EPI_LEAKY,
               out=dbuf[..., 64:96])
    assert close(dbuf[..., 64:96].cpu(), F.leaky_relu(ref_conv(torch.cat([xin, buf[..., :64]], 3), w32, b32), 0.2))
    w64, b64 = torch.randn(64, 192, 3, 3, generator=g) * 0.05, torch.randn(64, generator=g)
    y = ops.conv2d(xin.cuda(), ops.pack_weight(w64.cuda(), b64.cuda(), f16=True), x2=buf.cuda(), epilogue=ops.EPI_AXPY,
                   res=res.cuda(), sft_w=0.2)
    assert close(y.cpu(), ref_conv(torch.cat([xin, buf], 3), w64, b64) * 0.2 + res.double())
    wu, bu = torch.randn(64, 64, 3, 3, generator=g) * 0.05, torch.randn(64, generator=g)
    y = ops.conv2d(xin.cuda(), ops.pack_weight(wu.cuda(), bu.cuda(), f16=True, up2x=True), upsample=True, epilogue=ops.EPI_LEAKY)
    r = F.leaky_relu(ref_conv(xin, wu, bu, up=True), 0.2)
    # folded taps are summed in fp32 and THEN rounded to half (as for bf16): compare at the half-rounding level of the sum
    assert bool(((y.cpu().double() - r).abs() <= 2e-2 + 1e-2 * r.abs()).all())
    d = _digests()
    for name, (tmax, tmean) in (('x2_small', (2e-4, 3e-5)), ('x4_small', (2e-4, 3e-5)), ('x2_full', (1e-2, 1e-3))):
        case = d[name]
        net = _build(case).cuda().half()
        gold = torch.from_numpy(np.load(os.path.join(GOLD, f'rrdbnet_{name}.npz'))['out'])
        y = net(_input(case['shape'], case['in_seed']).cuda()).cpu()
        err = (y - gold).abs()
        print(f'rrdbnet fp16 {name}: max|d| {float(err.max()):.3e} mean|d| {float(err.mean()):.3e} (out max {float(gold.abs().max()):.3f})')
        assert y.dtype == torch.float32 and float(err.max()) <= tmax and float(err.mean()) <= tmean
        assert net.float().precision == 'fp32'
