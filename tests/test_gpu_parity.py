"""-m gpu: the HIP path (through the C ABI) against the CPU oracle / the reference's golden fixtures.

The checks themselves live in tools/gpu_check.py (also runnable stand-alone with a full per-check report):
  basic   transposes, weight packing, LayerNorm, argmax (incl. ties), codebook gather + AdaIN, GroupNorm tables
  conv    every instantiation of the implicit-GEMM kernel (3x3 s1/s2, 1x1, upsample, concat, NCHW in/out,
          prologues, epilogues) against fp64 torch CPU references, tolerance 2e-5 + 1e-5*|ref|
  attn    attention kernel (8x64 and 1x512, mixed leading dimensions) vs fp64 softmax attention, 5e-6
  blocks  ResBlock / AttnBlock / TransformerSALayer / Fuse_sft_block / Down / Upsample / VectorQuantizer modules
          against outputs of the REFERENCE's modules (tests/golden/blocks_seed7.npz), 2e-5
  net     whole CodeFormer.forward against the reference golden: pixels atol 1e-3 (north-star tolerance),
          logits 1e-4, code indices bit-exact; batch-of-4 == batch-of-1 bitwise; run-to-run bitwise
  bf16    the bf16-MFMA conv instantiations against fp64 convs of the SAME bf16-rounded operands (2e-4: only the
          accumulation order differs), and the whole net with precision='bf16' (BASELINE configs 3/5: generator + CFT in
          bf16, encoder on split halves, Transformer/argmax fp32): logits bitwise equal to the default mode, code indices exact, pixels within
          the stated bf16 gate of the fp32 reference (max|d| <= 0.25, mean|d| <= 0.02 on outputs of std 0.5); precision='fp16'
          (IEEE-half operands, same split): same logits / indices conditions, pixel gate max|d| <= 0.04, mean|d| <= 0.003
"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def chk():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from codeformer_amd import lib
    lib.load()   # loud failure if the native library did not travel
    spec = importlib.util.spec_from_file_location('gpu_check', os.path.join(ROOT, 'tools', 'gpu_check.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize('group', ['basic', 'conv', 'attn', 'blocks', 'net', 'bf16'])
def test_group(chk, group):
    chk.RESULTS.clear()
    chk.GROUPS[group]()
    bad = [r for r in chk.RESULTS if not r[1]]
    assert chk.RESULTS and not bad, bad


def test_tensor_boundary_kernels(chk):
    import numpy as np
    import torch
    from codeformer_amd import ops
    from oracle import codeformer_oracle as O
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (2, 64, 48, 3), dtype=np.uint8)
    t = ops.img_u8_to_tensor(torch.from_numpy(img).cuda())
    ref = torch.from_numpy(((img[..., ::-1] / 255.).astype(np.float32).transpose(0, 3, 1, 2) - 0.5) / 0.5)
    assert torch.equal(t.cpu(), ref)
    x = torch.randn(2, 3, 64, 48, generator=torch.Generator().manual_seed(2)) * 1.2
    x[0, :, 0, 0] = torch.tensor([0.5 / 255 * 2 - 1, 1.5 / 255 * 2 - 1, 2.5 / 255 * 2 - 1])
    got = ops.tensor_to_img_u8(x.cuda()).cpu().numpy()
    for b in range(2):
        assert np.array_equal(got[b], O.tensor2img_u8(x[b]))


def test_bundled_ops(chk):
    import torch
    from codeformer_amd import ops
    from oracle import codeformer_oracle as O
    from oracle.synth import seeded_randn
    x, b = seeded_randn((2, 5, 9, 7), 1), seeded_randn((5,), 2)
    assert torch.allclose(ops.fused_bias_act(x.cuda(), b.cuda()).cpu(), O.fused_bias_act(x, b), atol=1e-6)
    k = seeded_randn((4, 4), 3)
    for up, down, pad in ((1, 1, (1, 2)), (2, 1, (2, 1)), (1, 2, (1, 1)), (2, 2, (0, 3)), (1, 1, (-1, 2))):
        got = ops.upfirdn2d(x.cuda(), k.cuda(), up=up, down=down, pad=pad).cpu()
        ref = O.upfirdn2d(x, k, up=up, down=down, pad=pad)
        assert got.shape == ref.shape and torch.allclose(got, ref, atol=2e-6), (up, down, pad)
    # the reference's import surface (basicsr/ops/{fused_act,upfirdn2d}/__init__.py) on the same kernels
    from basicsr.ops.fused_act import FusedLeakyReLU
    from basicsr.ops.upfirdn2d import upfirdn2d
    m = FusedLeakyReLU(5).cuda()
    m.load_state_dict({'bias': b})
    assert torch.allclose(m(x.cuda()).cpu(), O.fused_bias_act(x, b), atol=1e-6)
    k3 = seeded_randn((3, 3), 4)
    assert torch.allclose(upfirdn2d(x.cuda(), k3.cuda(), up=3, down=2, pad=(2, 2)).cpu(),
                          O.upfirdn2d(x, k3, up=3, down=2, pad=(2, 2)), atol=2e-6)
    # the six LDS-tiled configurations (upfirdn2d_kernel.cu:251-291) on planes that span several tiles with ragged edges
    xl = seeded_randn((2, 3, 37, 150), 5)
    for up, down, ks, pads in ((1, 1, (4, 3), ((1, 2), (0, 0), (-2, 3))), (2, 1, (4, 2), ((2, 1), (0, 3))), (1, 2, (4, 2), ((1, 1), (0, 2), (-1, 0)))):
        for kk in ks:
            kern = seeded_randn((kk, kk), 10 + kk)
            for pad in pads:
                got = ops.upfirdn2d(xl.cuda(), kern.cuda(), up=up, down=down, pad=pad).cpu()
                ref = O.upfirdn2d(xl, kern, up=up, down=down, pad=pad)
                assert got.shape == ref.shape and torch.allclose(got, ref, atol=3e-6), (up, down, kk, pad)


def test_fused_bias_act_every_mode_dtype_and_autograd(chk):
    """cf_fused_bias_act_ex: the act*10+grad switch of fused_bias_act_kernel.cu:36-46 in float / half / bf16 against the oracle's
    restatement in the same dtype, and FusedLeakyReLU's first and second derivatives (fused_act.py:25-71) against torch autograd of
    the plain formula."""
    import torch
    from codeformer_amd import ops
    from oracle import codeformer_oracle as O
    from oracle.synth import seeded_randn
    from basicsr.ops.fused_act import FusedLeakyReLU, fused_leaky_relu
    x, b, r = seeded_randn((3, 6, 5, 7), 11), seeded_randn((6,), 12), seeded_randn((3, 6, 5, 7), 13)
    for dt, tol in ((torch.float32, 1e-6), (torch.float16, 2e-3), (torch.bfloat16, 2e-2)):
        for act in (1, 3):
            for grad in (0, 1, 2):
                for bias in (b, None):
                    got = ops.fused_bias_act(x.to(dt).cuda(), None if bias is None else bias.to(dt).cuda(), 0.2, 1.5, ref=r.to(dt).cuda(), act=act, grad=grad)
                    ref = O.fused_bias_act_modes(x.to(dt), None if bias is None else bias.to(dt), r.to(dt), act, grad, 0.2, 1.5)
                    assert got.dtype == dt and torch.allclose(got.float().cpu(), ref.float(), atol=tol, rtol=tol), (dt, act, grad)
    xg = x.clone().cuda().requires_grad_(True)
    m = FusedLeakyReLU(6).cuda()
    m.load_state_dict({'bias': b})
    y = m(xg)
    go = seeded_randn(tuple(y.shape), 14).cuda()
    gx, gb = torch.autograd.grad(y, (xg, m.bias), go, create_graph=True)
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = O.fused_bias_act(xr, br)
    gxr, gbr = torch.autograd.grad(yr, (xr, br), go.cpu())
    assert torch.allclose(y.detach().cpu(), yr.detach(), atol=1e-6)
    assert torch.allclose(gx.detach().cpu(), gxr, atol=1e-6) and torch.allclose(gb.detach().cpu(), gbr, atol=1e-4)
    # second order: d/d(grad_output) of <grad_input, v> is the same gate applied to v
    v = seeded_randn(tuple(y.shape), 15).cuda()
    go2 = go.clone().requires_grad_(True)
    gx2, _ = torch.autograd.grad(fused_leaky_relu(xg, m.bias, 0.2, 2 ** 0.5), (xg, m.bias), go2, create_graph=True)
    (ggo,) = torch.autograd.grad(gx2, go2, v)
    gate = torch.where(yr.detach() > 0, torch.ones_like(yr), torch.full_like(yr, 0.2)) * 2 ** 0.5
    assert torch.allclose(ggo.cpu(), gate * v.cpu(), atol=1e-6)


def test_inpainting_config(chk):
    """Config 5 network shape (codebook 512, 3 fuse levels, w=1, adain=False) against the reference golden."""
    import numpy as np
    import torch
    from oracle.synth import seeded_input
    net = chk.build_net(512, ('32', '64', '128')).cuda()
    g = np.load(os.path.join(ROOT, 'tests/golden/inpaint_seed0_face0.npz'))
    out, logits, _ = net(seeded_input(1).cuda(), w=1, adain=False)
    assert float((logits.cpu() - torch.from_numpy(g['logits'])).abs().max()) <= 1e-4
    assert np.array_equal(net.last_indices.cpu().numpy(), g['idx'])
    assert float((out.cpu()[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs().max()) <= 1e-3
    logits_default = logits
    net.precision = 'fp32'                                                   # (the run above was the default, split-half, mode)
    out, logits, _ = net(seeded_input(1).cuda(), w=1, adain=False)
    assert float((logits.cpu() - torch.from_numpy(g['logits'])).abs().max()) <= 1e-4
    assert float((out.cpu()[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs().max()) <= 1e-3
    # BASELINE config 5 names bf16: 16-bit operands in generator + CFT only (the encoder runs as in the default mode) -> logits bitwise
    # those of the default mode, indices exact, pixels inside the gates of tools/gpu_check.py:g_bf16 (bf16 0.25 max, fp16 0.04 max on outputs of std ~0.5)
    for prec, gate in (('bf16', 0.25), ('fp16', 0.04)):
        net.precision = prec
        out16, logits16, _ = net(seeded_input(1).cuda(), w=1, adain=False)
        assert torch.equal(logits16, logits_default) and np.array_equal(net.last_indices.cpu().numpy(), g['idx'])
        d = (out16 - out).abs()
        print(f'inpainting config {prec}: max|d| {float(d.max()):.4f} mean|d| {float(d.mean()):.5f}')
        assert float(d.max()) <= gate


# ---- size-independent properties at BASELINE config-2 sizes (the CPU oracle is too slow there) -------------------------
@pytest.mark.parametrize('precision', ['f16x2', 'fp32', 'bf16'])
def test_full_size_batch16_is_bitwise_batch_invariant(chk, precision):
    """Config 2 shape (16 faces): every face of the batch equals the same face restored alone / in a 2-rank shard -- in the product's default
    mode, in the IEEE-fp32 mode bench.py's headline runs (whose token GEMMs take another kernel at sixteen faces than at one) and in the bf16
    storage mode of configs 3 / 5."""
    import torch
    from oracle.synth import seeded_input
    net = chk.build_net().cuda()
    net.precision = precision
    x = seeded_input(16).cuda()
    full = net(x, w=0.5, adain=True)
    for i in (0, 7, 15):
        one = net(x[i:i + 1].contiguous(), w=0.5, adain=True)
        assert torch.equal(full[0][i:i + 1], one[0]) and torch.equal(full[1][i:i + 1], one[1])
    half = net(x[8:].contiguous(), w=0.5, adain=True)           # what rank 1 of a 2-GPU run computes
    assert torch.equal(full[0][8:], half[0])
    assert torch.isfinite(full[0]).all()


def test_full_size_conv_linearity_and_layout_round_trip(chk):
    import torch
    from codeformer_amd import ops
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(2, 256, 256, 128, generator=g).cuda(), torch.randn(2, 256, 256, 128, generator=g).cuda()
    pw = ops.pack_weight((torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda(), None)
    lhs = ops.conv2d(1.5 * x - 0.5 * y, pw)
    rhs = 1.5 * ops.conv2d(x, pw) - 0.5 * ops.conv2d(y, pw)
    assert float((lhs - rhs).abs().max()) < 2e-5 * max(1.0, float(rhs.abs().max()))
    assert torch.equal(ops.to_nhwc(ops.to_nchw(x)), x)


def test_shape_errors_are_raised_not_swallowed(chk):
    import torch
    net = chk.build_net().cuda()
    with pytest.raises(ValueError, match='512x512'):
        net(torch.zeros(1, 3, 256, 256, device='cuda'), w=0.5)
    net.precision = 'fp8'
    with pytest.raises(ValueError, match='precision'):
        net(torch.zeros(1, 3, 512, 512, device='cuda'), w=0.5)
    from codeformer_amd import ops
    pw = ops.pack_weight(torch.zeros(64, 64, 3, 3, device='cuda'))
    with pytest.raises(RuntimeError, match='cf_conv2d'):      # stride-2 output 10x10 is off the tile grid: refused, not garbage
        ops.conv2d(torch.zeros(1, 20, 20, 64, device='cuda'), pw, stride=2)
    with pytest.raises(RuntimeError, match='cf_conv2d'):      # off-grid size + GroupNorm statistics: refused
        ops.conv2d(torch.zeros(1, 20, 20, 64, device='cuda'), pw, emit_stats=True)
    assert ops.conv2d(torch.ones(1, 20, 20, 64, device='cuda'), pw).abs().max().item() == 0.0   # stride 1: masked edge tiles


def test_packed_weight_cache_follows_the_parameters(chk):
    """load_state_dict / in-place parameter updates must reach the kernels (the packed copies are a cache)."""
    import torch
    from oracle.synth import seeded_input
    net = chk.build_net().cuda()
    x = seeded_input(1).cuda()
    a = net(x, w=0.5, adain=True)[0].clone()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd2 = dict(sd)
    sd2['generator.blocks.22.conv2.weight'] = sd['generator.blocks.22.conv2.weight'] * 1.5
    net.load_state_dict(sd2)
    b = net(x, w=0.5, adain=True)[0].clone()
    assert not torch.equal(a, b)
    net.load_state_dict(sd)
    assert torch.equal(net(x, w=0.5, adain=True)[0], a)
    bias0 = net.generator.blocks[24].bias.detach().clone()
    with torch.no_grad():
        net.generator.blocks[24].bias.add_(0.25)          # versioned in-place op
    c = net(x, w=0.5, adain=True)[0]
    assert float((c - a - 0.25).abs().max()) < 1e-5
    net.generator.blocks[24].bias.data.copy_(bias0)        # un-versioned edit (.data) -> explicit invalidation
    net.invalidate_packed_weights()
    assert torch.equal(net(x, w=0.5, adain=True)[0], a)


def test_hip_graph_replay_is_bitwise_eager(chk):
    """use_hip_graphs: capture once, replay; identical bits to the eager path, survives a weight reload."""
    import torch
    from oracle.synth import seeded_input
    net = chk.build_net().cuda()
    x = seeded_input(2).cuda()
    assert net.use_hip_graphs == 'auto' and net.graph_max_batch == 4     # the default: replay for batches of at most four faces
    net.use_hip_graphs = False
    eager = [t.clone() for t in net(x, w=0.5, adain=True)]
    assert not net._graphs
    net.use_hip_graphs = 'auto'
    auto = net(x, w=0.5, adain=True)
    assert len(net._graphs) == 1 and all(torch.equal(a, b) for a, b in zip(auto, eager))
    net(seeded_input(8).cuda(), w=0.5, adain=True)
    assert len(net._graphs) == 1                                         # eight faces: eager
    net.use_hip_graphs = True
    for _ in range(2):                                   # first call captures, second replays
        got = net(x, w=0.5, adain=True)
        assert all(torch.equal(a, b) for a, b in zip(got, eager))
    y = seeded_input(2, seed=99).cuda()
    net.use_hip_graphs = False
    eager_y = [t.clone() for t in net(y, w=0.5, adain=True)]
    net.use_hip_graphs = True
    assert all(torch.equal(a, b) for a, b in zip(net(y, w=0.5, adain=True), eager_y))   # same graph, new input
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd['generator.blocks.24.bias'] = sd['generator.blocks.24.bias'] + 1.0
    net.load_state_dict(sd)                                                             # repack -> recapture
    assert float((net(y, w=0.5, adain=True)[0] - eager_y[0] - 1.0).abs().max()) < 1e-5


def test_forward_under_inference_mode_and_replaced_parameters(chk):
    """Advisor items of round 4: (a) torch.inference_mode() -- inference tensors carry no version counter, the range-scale cache must
    not read one; bits equal to the no_grad call, eager and graphed.  (b) A REPLACED Parameter object (`m.bias = nn.Parameter(...)`: not
    an in-place update, not load_state_dict) must reach a default-on graph replay.  (c) `last_indices` of a replay is a fresh tensor."""
    import torch
    from oracle.synth import seeded_input
    net = chk.build_net().cuda()
    x = seeded_input(2).cuda()
    for graphs in (False, 'auto'):
        net.use_hip_graphs = graphs
        ref = [t.clone() for t in net(x, w=0.5, adain=True)]
        with torch.inference_mode():
            got = net(x, w=0.5, adain=True)
            got2 = net(x.clone(), w=0.5, adain=True)     # an inference-tensor input as well
        assert all(torch.equal(a, b) for a, b in zip(got, ref)) and all(torch.equal(a, b) for a, b in zip(got2, ref))
    net.use_hip_graphs = 'auto'
    a = net(x, w=0.5, adain=True)[0].clone()
    idx_a = net.last_indices
    y = seeded_input(2, seed=5).cuda()
    net(y, w=0.5, adain=True)
    assert net.last_indices is not idx_a and idx_a.data_ptr() != net.last_indices.data_ptr()
    net.use_hip_graphs = False
    net(x, w=0.5, adain=True)
    assert torch.equal(net.last_indices, idx_a)           # the earlier call's indices were not overwritten by the later replay
    net.use_hip_graphs = 'auto'
    old = net.generator.blocks[24].bias
    net.generator.blocks[24].bias = torch.nn.Parameter(old.detach() + 0.5, requires_grad=False)   # a new object, version 0
    b = net(x, w=0.5, adain=True)[0]
    assert float((b - a - 0.5).abs().max()) < 1e-5
    from codeformer_amd.archs.vqgan_arch import _Conv3x3
    net.generator.blocks[24] = _Conv3x3(64, 3).cuda().requires_grad_(False)                      # a replaced sub-module
    c = net(x, w=0.5, adain=True)[0]
    net.use_hip_graphs = False
    assert torch.equal(c, net(x, w=0.5, adain=True)[0])


def test_code_only_and_vqautoencoder_module_api(chk):
    """Other callers of the boundary: code_only=True (training stage II) and VQAutoEncoder.forward (scripts/inference_vqgan.py)."""
    import torch
    from oracle.synth import seeded_input
    net = chk.build_net().cuda()
    x = seeded_input(1).cuda()
    full = net(x, w=0.5, adain=True)
    logits, lq = net(x, w=0.5, code_only=True)
    assert torch.equal(logits, full[1]) and torch.equal(lq, full[2])
    # VQAutoEncoder path: encoder -> L2 nearest code -> generator (no fusion); self-consistency + host generator on the same codes
    import codeformer_amd.archs  # noqa: F401
    from codeformer_amd.utils.registry import ARCH_REGISTRY
    torch.manual_seed(3)
    vq = ARCH_REGISTRY.get('VQAutoEncoder')(512, 64, [1, 2, 2, 4, 4, 8], 'nearest', 2, [16], 1024).eval()
    vq_gpu = ARCH_REGISTRY.get('VQAutoEncoder')(512, 64, [1, 2, 2, 4, 4, 8], 'nearest', 2, [16], 1024).eval()
    vq_gpu.load_state_dict(vq.state_dict())
    vq_gpu = vq_gpu.cuda()
    out, loss, stats = vq_gpu(x)
    assert out.shape == (1, 3, 512, 512) and torch.isfinite(out).all() and torch.isfinite(loss)
    idx = stats['min_encoding_indices'].view(-1)
    assert idx.shape == (256,) and int(idx.min()) >= 0 and int(idx.max()) < 1024
    zq = vq_gpu.quantize.get_codebook_feat(idx, [1, 16, 16, 256])
    assert torch.equal(zq, vq.quantize.embedding.weight[idx.cpu()].view(1, 16, 16, 256).permute(0, 3, 1, 2).cuda())
    with torch.no_grad():
        ref = vq.generator(zq.cpu())                        # host (stock torch) generator on the same quantised latent
    assert float((vq_gpu.generator(zq).cpu() - ref).abs().max()) < 1e-3


def test_winograd_conv_kernel(chk):
    """cf_conv2d(winograd=1) -- the F(2x2,3x3) evaluation of the 3x3 stride-1 convolutions -- against fp64 references, for every
    prologue / epilogue / concat combination the network uses and for all its layer widths: tolerance 2e-5 + 1e-5*|ref| (the bound
    of the direct kernel; measured errors are 1-6e-6, below the direct kernel's), epilogue GroupNorm partials relative 1e-5,
    run-to-run and batch bitwise.  Shapes the kernel does not cover are refused by the C ABI, not silently rerouted."""
    import importlib.util
    import torch
    from codeformer_amd import ops
    spec = importlib.util.spec_from_file_location('wino_check', os.path.join(ROOT, 'tools', 'wino_check.py'))
    wc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wc)
    cases = [dict(B=1, H=16, W=16, cin=16, cout=64),
             dict(B=2, H=16, W=32, cin=32, cout=64, seed=1),
             dict(B=2, H=16, W=16, cin=64, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=2),
             dict(B=2, H=32, W=32, cin=128, cout=64, c_split=64, prologue=ops.PRO_LEAKY, epilogue=ops.EPI_SFT, seed=3),
             dict(B=2, H=16, W=16, cin=512, cout=512, prologue=ops.PRO_AFFINE_SWISH, stats=True, seed=4),
             dict(B=1, H=64, W=64, cin=256, cout=256, prologue=ops.PRO_AFFINE, stats=True, seed=5),
             dict(B=1, H=40, W=48, cin=64, cout=64, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=6,
                  direct=False)]   # 40 rows: off the direct kernel's 16-row grid, where it has no statistics epilogue
    for c in cases:
        B, H, W, cin, cout = (c.pop(k) for k in ('B', 'H', 'W', 'cin', 'cout'))
        ed, ew, es, rmax = wc.case(B, H, W, cin, cout, timing=False, **c)
        assert ew <= 2e-5 + 1e-5 * rmax and es <= 1e-5, (B, H, W, cin, cout, c, ew, es)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 32, 48, 128, generator=g).cuda()
    pw = ops.pack_weight((torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda(), torch.randn(128, generator=g).cuda(),
                         bf16=ops.WINOGRAD)
    y = ops.conv2d(x, pw)
    assert torch.equal(y, ops.conv2d(x, pw)) and torch.equal(y[1:2], ops.conv2d(x[1:2].contiguous(), pw))
    with pytest.raises(RuntimeError, match='winograd'):
        ops.conv2d(x[:, :20].contiguous(), pw)       # 20 rows: not a whole number of 8x16 patches
    with pytest.raises(RuntimeError, match='winograd'):
        ops.conv2d(x, pw, stride=2)
    assert ops.winograd_ok(128, 128, 32, 48) and not ops.winograd_ok(128, 128, 20, 48) and not ops.winograd_ok(128, 96, 32, 48)
