"""Host-side pieces around the hot path (CPU): the cv2-rule bilinear resize, EXIF handling of the image reader, the content-hash
gate of the build script and the upsampler flags of the entrypoint on the --has_aligned path."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_resize_bilinear_follows_the_cv2_inter_linear_rule():
    from codeformer_amd.utils.img_util import resize_bilinear
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    assert resize_bilinear(a, (5, 7)) is a
    up = resize_bilinear(a, (10, 14))
    assert up.shape == (14, 10, 3) and np.array_equal(up[0, 0], a[0, 0]) and np.array_equal(up[-1, -1], a[-1, -1])   # clamped borders
    # dst x = 1 -> src 0.25: weights 0.75 / 0.25 on both axes where interior
    want = (a[0, 0].astype(np.int64) * 1536 + a[0, 1].astype(np.int64) * 512)
    assert np.array_equal(up[0, 1], ((want * 2048 + (1 << 21)) >> 22).astype(np.uint8))
    # shrinking takes TWO samples per axis (no antialiasing): 16 -> 8 is the mean of neighbouring pairs, 16 -> 4 skips samples
    g = np.tile(np.arange(0, 256, 16, dtype=np.uint8)[None, :, None], (4, 1, 3))
    assert resize_bilinear(g, (8, 4))[0, :, 0].tolist() == [8, 40, 72, 104, 136, 168, 200, 232]
    assert resize_bilinear(g, (4, 4))[0, :, 0].tolist() == [24, 88, 152, 216]     # src 1.5, 5.5, ...: mean of samples (1,2), (5,6), ...
    assert np.unique(resize_bilinear(np.full((9, 8, 3), 77, np.uint8), (3, 5))).tolist() == [77]


def test_imread_applies_exif_orientation_like_cv2(tmp_path):
    from PIL import Image
    from codeformer_amd.utils.img_util import imread_bgr
    a = np.zeros((4, 6, 3), np.uint8)
    a[0, 0] = (255, 0, 0)
    im = Image.fromarray(a)
    ex = im.getexif()
    ex[0x0112] = 6          # "rotate 90 CW to display"
    im.save(tmp_path / 'r.jpg', exif=ex, quality=100, subsampling=0)
    b = imread_bgr(str(tmp_path / 'r.jpg'))
    assert b.shape == (6, 4, 3) and b[0, -1, 2] > 200 and b[0, 0, 2] < 60     # the red corner moved to the top right; BGR order


def test_build_id_follows_the_sources():
    from codeformer_amd import build, lib
    assert build.built_id() == build.source_hash() and not build.needs_build()
    assert lib.load().cf_build_id().decode() == build.source_hash()


def test_upsampler_flags_are_accepted_on_the_aligned_path(tmp_path):
    """--bg_upsampler realesrgan / --face_upsample: the reference builds the upsampler but never runs it for aligned crops
    (inference_codeformer.py:217-229); without its checkpoint the run goes on and says so."""
    from PIL import Image
    src = tmp_path / 'cropped_faces'
    os.makedirs(src)
    Image.fromarray(np.random.default_rng(2).integers(0, 256, (512, 512, 3), dtype=np.uint8)).save(src / 'a.png')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_codeformer.py'), '--has_aligned', '-i', str(src), '-o',
                        str(tmp_path / 'o'), '--device', 'cpu', '--random_init_seed', '0', '--bg_upsampler', 'realesrgan',
                        '--face_upsample'], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.listdir(tmp_path / 'o' / 'restored_faces') == ['a.png'] and 'upsampler' in r.stdout


def test_operand_range_host_rules():
    """Host-side range rules of the 16-bit-operand kernels (ops.gn_range_ok / exact_code) and the determinism of the range variants."""
    import torch
    from codeformer_amd import ops
    from oracle.synth import range_variant
    assert ops.gn_range_ok(1.2, 0.3, 4 * 512 * 512)            # reference-like gains on the largest layer: fine
    assert not ops.gn_range_ok(20.0, 0.0, 4 * 512 * 512)       # 20 * 1024 > 16376
    assert ops.gn_range_ok(20.0, 0.0, 16 * 16 * 16)            # ... but fine on a 16x16 latent (sqrt(n) = 64)
    assert ops.exact_code(ops.SPLIT) == ops.WINOGRAD and ops.exact_code(ops.SPLIT_DIRECT) == 0 and ops.exact_code(2) == ops.WINOGRAD
    assert ops.exact_code(1) == 1 and ops.exact_code(0) == 0 and ops.exact_code(ops.WINOGRAD) == ops.WINOGRAD
    sd = {'generator.blocks.5.conv2.weight': torch.ones(4, 4, 3, 3), 'generator.blocks.5.conv2.bias': torch.ones(4),
          'generator.blocks.5.norm1.weight': torch.ones(4), 'fuse_convs_dict.32.scale.2.weight': torch.ones(4, 4, 3, 3),
          'fuse_convs_dict.32.scale.2.bias': torch.ones(4), 'fuse_convs_dict.32.encode_enc.conv2.weight': torch.ones(4, 4, 3, 3)}
    big = range_variant(sd, 'big', calib={'fuse_convs_dict.32.scale.2.weight': 8.0})
    assert float(big['generator.blocks.5.conv2.weight'][0, 0, 0, 0]) == 64.0 and float(big['generator.blocks.5.norm1.weight'][0]) == 1.0
    assert float(big['fuse_convs_dict.32.scale.2.weight'][0, 0, 0, 0]) == 0.125 and float(big['fuse_convs_dict.32.scale.2.bias'][0]) == 1.0
    small = range_variant(sd, 'small')
    assert float(small['generator.blocks.5.conv2.bias'][0]) == 1.0 / 4096.0
    h1, h2 = range_variant(sd, 'heavy'), range_variant(sd, 'heavy')
    assert all(torch.equal(h1[k], h2[k]) for k in sd)


def test_kernel_selection_host_rules():
    """Which kernel a layer takes is decided on the host from its per-image shape (never the batch): the split-K counts, the stride-2 and
    1x1 forms of the split-half kernel, the folded upsample from 16x16 up.  No GPU: PackedWeight objects are built by hand."""
    from codeformer_amd import ops
    lin = ops.PackedWeight(None, None, 512, 512, 1, 512, 512)
    lin1024 = ops.PackedWeight(None, None, 512, 1024, 1, 512, 1024)
    # Linear layers: the largest split with at most 128 workgroups in flight; eligibility by the per-image shape only
    assert [ops.splitk_for(lin, 16, 16, 512, b) for b in (1, 2, 4, 8, 16)] == [4, 2, 1, 1, 1]
    assert [ops.splitk_for(lin1024, 16, 16, 1024, b) for b in (1, 2, 4, 16)] == [4, 2, 1, 1]
    assert ops.splitk_for(lin, 64, 64, 512, 1) == 0 and ops.splitk_for(lin, 16, 16, 320, 1) == 0
    # split-half token GEMMs of at most 2^20 outputs and K <= 1024 -- one to four faces, eight for the 512-column layers: the in-workgroup
    # split (ABI v21: same bits, no workspace)
    glin = ops.PackedWeight(None, None, 512, 512, 1, 512, 512, bf16=ops.OPERAND_F16X2)
    glin1024 = ops.PackedWeight(None, None, 512, 1024, 1, 512, 1024, bf16=ops.OPERAND_F16X2)          # 1024 -> 512 (MLP-down)
    gup = ops.PackedWeight(None, None, 1024, 512, 1, 1024, 512, bf16=ops.OPERAND_F16X2)               # 512 -> 1024 (q|k, MLP-up)
    assert [ops.splitk_for(glin, 16, 16, 512, b) for b in (1, 2, 4, 8, 16)] == [ops.SPLITK_IN_WORKGROUP] * 4 + [1]
    assert [ops.splitk_for(glin1024, 16, 16, 1024, b) for b in (1, 4, 8, 16)] == [ops.SPLITK_IN_WORKGROUP] * 3 + [1]
    assert [ops.splitk_for(gup, 16, 16, 512, b) for b in (1, 4, 8)] == [ops.SPLITK_IN_WORKGROUP, ops.SPLITK_IN_WORKGROUP, 1]
    assert ops.splitk_for(ops.PackedWeight(None, None, 512, 2048, 1, 512, 2048, bf16=ops.OPERAND_F16X2), 16, 16, 2048, 1) == 4
    # Winograd latents: at most 256 workgroups; 16 faces -> one workgroup per tile
    wino = ops.PackedWeight(None, None, 512, 512, 9, 512, 512, bf16=ops.OPERAND_F16X2, wino=True)
    assert [ops.splitk_for(wino, 16, 16, 512, b) for b in (1, 2, 4, 8, 16, 32)] == [4, 4, 4, 2, 1, 1]
    assert ops.splitk_for(wino, 32, 32, 512, 1) == 0
    # shapes of the stride-2 and 1x1 forms
    assert ops.split_s2_ok(64, 64, 512, 512) and ops.split_s2_ok(256, 256, 32, 32) and ops.split_s2_ok(16, 64, 16, 32)
    assert not ops.split_s2_ok(24, 64, 64, 64) and not ops.split_s2_ok(64, 96, 64, 64) and not ops.split_s2_ok(64, 64, 24, 32)
    assert ops.split_1x1_ok(128, 64, 512, 512) and ops.split_1x1_ok(512, 256, 64, 64, c_split=256)
    assert not ops.split_1x1_ok(512, 256, 32, 32)              # 1024 pixels: token-sized, stays on the GEMM
    assert not ops.split_1x1_ok(128, 64, 36, 64) and not ops.split_1x1_ok(128, 48, 64, 64) and not ops.split_1x1_ok(96, 64, 64, 64, c_split=48)
    # operand codes: the folded upsample takes the direct split kernel from 16x16 up, plain 3x3 below 32x32 the Winograd form
    assert ops.conv_code(ops.SPLIT, 512, 512, 16, 16, up2x=True) == ops.SPLIT and ops.conv_code(ops.SPLIT, 512, 512, 16, 16) == ops.WSPLIT
    assert ops.conv_code(ops.SPLIT_DIRECT, 512, 512, 16, 16) == 0 and ops.conv_code(ops.SPLIT, 128, 128, 24, 16, up2x=True) == 0
    # a stride-2 / 1x1 weight form is tied to its descriptor
    s2 = ops.PackedWeight(None, None, 64, 64, 9, 64, 64, bf16=ops.OPERAND_F16X2, s2=True)
    c1 = ops.PackedWeight(None, None, 64, 128, 1, 64, 128, bf16=ops.OPERAND_F16X2, conv1=True)
    assert ops.needs_act_scale(s2) == ops.RANGE_SCALE and ops.needs_act_scale(c1) == ops.RANGE_SCALE and not ops.needs_act_scale(lin)


def test_layernorm_bound_decides_the_operands_of_the_transformer_gemms():
    """ln_code: a Linear layer behind a LayerNorm takes split-half operands only while |gamma| sqrt(C - 1) + |beta| (+ |pos|) stays inside
    the IEEE-half range; the bound follows the parameters (in-place edits bump their version)."""
    import torch
    from codeformer_amd import ops
    from codeformer_amd.archs.codeformer_arch import TransformerSALayer, ln_code
    layer = TransformerSALayer(embed_dim=512, nhead=8, dim_mlp=1024)
    pos = torch.zeros(256, 512)
    assert ln_code(layer, layer.norm1, ops.GSPLIT, pos) == ops.GSPLIT and ln_code(layer, layer.norm2, ops.GSPLIT) == ops.GSPLIT
    assert ln_code(layer, layer.norm1, 0, pos) == 0
    with torch.no_grad():
        layer.norm2.weight.mul_(2000.0)                    # 2000 * sqrt(511) > 32768
    assert ln_code(layer, layer.norm2, ops.GSPLIT) == 0 and ln_code(layer, layer.norm1, ops.GSPLIT, pos) == ops.GSPLIT
    pos2 = torch.full((256, 512), 4.0e4)
    assert ln_code(layer, layer.norm1, ops.GSPLIT, pos2) == 0


def test_stride2_space_to_depth_identity():
    """The algebra the stride-2 form of the split-half kernel rests on (cf_split.hip, split_weight_value_s2): Downsample's
    pad(0,1,0,1) + 3x3 stride-2 conv (vqgan_arch.py:117-126) equals a 2x2 stride-1 conv of the space-to-depth view
    X[(p,q,c)][i][j] = x[c][2i+p][2j+q] (zero row / column appended bottom / right) with W'[n][(p,q,c)][ty][tx] = w[n][c][2ty+p][2tx+q]
    where that tap exists and 0 elsewhere -- checked in fp64 on CPU, with the channel order (p, q, c) the gather uses."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(11)
    B, C, Co, H, W = 2, 5, 7, 12, 16
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(Co, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    X = torch.stack([x[:, :, p::2, q::2] for p in (0, 1) for q in (0, 1)], dim=1).reshape(B, 4 * C, H // 2, W // 2)   # channel = (p*2+q)*C + c
    Wp = torch.zeros(Co, 4 * C, 2, 2, dtype=torch.float64)
    zero_blocks = 0
    for ty in (0, 1):
        for tx in (0, 1):
            for p in (0, 1):
                for q in (0, 1):
                    ky, kx = 2 * ty + p, 2 * tx + q
                    if ky > 2 or kx > 2:
                        zero_blocks += 1
                        continue
                    Wp[:, (p * 2 + q) * C:(p * 2 + q + 1) * C, ty, tx] = w[:, :, ky, kx]
    assert zero_blocks == 7                                     # 16 tap x parity blocks, 9 taps
    got = F.conv2d(F.pad(X, (0, 1, 0, 1)), Wp, b)               # 2x2 taps at (i + ty, j + tx); beyond the last row / column: zero
    assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-12


def test_linear_input_bounds_one_layer_behind_a_layernorm():
    """bounded_code: out-proj / MLP-down take split-half operands while max_n(|W| L + |b|) stays inside the half range."""
    import torch
    from codeformer_amd import ops
    from codeformer_amd.archs.codeformer_arch import TransformerSALayer, bounded_code
    layer = TransformerSALayer(embed_dim=512, nhead=8, dim_mlp=1024)
    w, b = layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias
    assert bounded_code(layer, 'o', ops.GSPLIT, layer.norm1, w[1024:], b[1024:]) == ops.GSPLIT
    assert bounded_code(layer, 'd', ops.GSPLIT, layer.norm2, layer.linear1.weight, layer.linear1.bias) == ops.GSPLIT
    assert bounded_code(layer, 'd', 0, layer.norm2, layer.linear1.weight, layer.linear1.bias) == 0
    with torch.no_grad():
        layer.linear1.weight.mul_(500.0)                   # rows of ~1000 * 0.03 * 500 * sqrt(511): far outside
    assert bounded_code(layer, 'd', ops.GSPLIT, layer.norm2, layer.linear1.weight, layer.linear1.bias) == 0
    assert bounded_code(layer, 'o', ops.GSPLIT, layer.norm1, w[1024:], b[1024:]) == ops.GSPLIT


def test_entry_scripts_compile():
    """bench.py / __graft_entry__.py / inference_codeformer.py and the tools are driver-facing scripts no CPU test imports: at least they
    must compile (a quote inside an f-string once slipped through a GPU-less edit)."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, f) for f in ('bench.py', '__graft_entry__.py', 'inference_codeformer.py')] + sorted(glob.glob(os.path.join(root, 'tools', '*.py')))
    assert len(files) > 10
    for f in files:
        compile(open(f).read(), f, 'exec')


def test_param_signature_follows_replaced_parameters_and_modules():
    """The graph-replay key of CodeFormer (advisor item of round 4): a REPLACED Parameter object (version 0 again, possibly the same
    size) and a replaced / added sub-module must change the signature; in-place versioned updates and load_state_dict too; calling it
    twice without a change must not."""
    import torch
    import codeformer_amd.archs  # noqa: F401
    from codeformer_amd.utils.registry import ARCH_REGISTRY
    net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9, connect_list=['32', '64', '128', '256']).eval()
    s0 = net._param_signature()
    assert net._param_signature() == s0
    old = net.generator.blocks[24].bias
    net.generator.blocks[24].bias = torch.nn.Parameter(old.detach().clone(), requires_grad=False)     # a new object with equal content
    s1 = net._param_signature()
    assert s1 != s0
    with torch.no_grad():
        net.generator.blocks[24].bias.add_(1.0)                                                          # versioned in-place update
    s2 = net._param_signature()
    assert s2 != s1
    from codeformer_amd.archs.vqgan_arch import _Conv3x3
    net.generator.blocks[24] = _Conv3x3(64, 3)                                                           # a replaced sub-module
    s3 = net._param_signature()
    assert s3 != s2 and net._param_signature() == s3
    net.extra = torch.nn.Linear(4, 4)                                                                    # an added sub-module
    assert net._param_signature() != s3
    import copy
    import pickle
    copy.deepcopy(net.ft_layers[0])                                                                      # module attributes stay copyable / picklable
    pickle.dumps(net.idx_pred_layer)


def test_version_reads_are_safe_under_inference_mode():
    """ops.tensor_version: inference tensors carry no version counter (reading `._version` raises); version-keyed caches skip them."""
    import torch
    from codeformer_amd import ops
    t = torch.zeros(3)
    assert ops.tensor_version(t) == t._version
    with torch.inference_mode():
        u = torch.zeros(3)
        assert ops.tensor_version(u) is None
    assert ops.tensor_version(u) is None and u.is_inference()


def test_bf16_gate_is_a_stated_multiple_of_the_intrinsic_cost(golden_dir):
    """Where the bf16 pixel gate (0.196 / 0.0152) comes from: the CPU oracle with bf16-rounded operands in the convolutions the 'bf16' mode
    puts on bf16 MFMA AND (round 6) bf16 storage of the generator / fusion activations of more than 1024 pixels, against the reference's fp32
    output at w = 0.7 -- the cost of that arithmetic for any implementation.  The gate must stay between 1.2x and 2x that cost
    (tools/bf16_gate_derivation.py; profiles/r06_bf16_gate_derivation.txt has the numbers: the last 'vs the same golden' line is the storage mode)."""
    import re
    txt = open(os.path.join(ROOT, 'profiles', 'r06_bf16_gate_derivation.txt')).read()
    mx, mean = (float(v) for v in re.findall(r'vs the same golden: max ([0-9.]+) mean ([0-9.]+)', txt)[-1])
    assert 'bf16 storage' in txt
    assert 1.2 * mx <= 0.196 <= 2.0 * mx and 1.2 * mean <= 0.0152 <= 2.0 * mean, (mx, mean)
    src = open(os.path.join(ROOT, 'tests', 'test_gpu_real_images.py')).read()
    assert "'bf16': (0.196, 0.0152)" in src


def test_bench_line_stays_under_two_kilobytes():
    """The driver keeps a 2 KB tail of bench.py's stdout and truncates long strings (round-5 review: the fp32 figure was cut out of its record).
    The line is assembled from `compact_roofline` + flat scalars; with every field filled by values of realistic width it must stay below 2000
    characters, every string below 128, and the committed line of the round must carry the contract's fields with the IEEE-fp32 headline."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = {'kind': 'conv3x3_wino43', 'kernel': 'x' * 300, 'bound': 'mfma', 'achieved': 123.45, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': 0.5351, 'traffic': 1186352743,
            'traffic_recorded': 1186352743, 'traffic_source': 'profiles/r06_pmc_bench_fp32.json', 'traffic_per_alg_bytes': 1.214, 'avg_launch_ms': 0.6357, 'launches_per_step': 54,
            'ms_per_step': 34.33, 'alg_bytes_per_launch': 977596568, 'frac_hbm_peak_alg_bytes': 0.1922, 'effective_tflops': 336.68, 'frac_algorithmic': 2.1404,
            'executed_tflops': 84.17, 'frac_executed': 0.5351, 'other_kernels': {'a': {'kernel': 'y' * 500}}, 'achieved_is': 'z' * 200}
    r = bench.compact_roofline(full)
    assert 'other_kernels' not in r and len(r['kernel']) <= 110 and all(not isinstance(v, (dict, list)) for v in r.values())
    line = json.load(open(os.path.join(ROOT, 'profiles', 'r06_bench.json')))
    assert line['dtype'] == 'f32' and 'IEEE fp32' in line['config']['workload'] and line['vs_baseline'] is None and line['unit'] == 'faces/s'
    for k in ('f16x2_faces_per_s', 'f16x2_ms_per_step', 'config3_rank_faces_per_s', 'config3_rank_ms_per_step'):
        assert k in line and line['config'][k] == line[k]            # flat scalars, the throughput figures mirrored where the driver's parsed record keeps them
    for k in ('f16x2_max_abs_pixel_diff', 'f16x2_code_indices_equal', 'config3_max_abs_pixel_diff', 'config3_mean_abs_pixel_diff', 'config3_code_indices_equal'):
        assert k in line
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(line['roofline']) and {'value', 'unit', 'cores', 'kind', 'sample'} <= set(line['cpu_baseline'])
    line['roofline'] = r
    text = json.dumps(line)
    assert len(text) < 2000, len(text)

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(t) for t in strings(line)) < 128


def test_no_timing_or_ablation_scaffolds_in_the_product_sources():
    """Round 6 moved every timing / ablation scaffold out of codeformer_amd/csrc (tools/experiments/*.patch restore them for an experiment build):
    the macro names must not come back, and the stripping tool must find nothing to do."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('strip_mod', os.path.join(ROOT, 'tools', 'strip_experiment_macros.py'))
    st = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(st)
    csrc = os.path.join(ROOT, 'codeformer_amd', 'csrc')
    for f in sorted(os.listdir(csrc)):
        src = open(os.path.join(csrc, f)).read()
        code = '\n'.join(l.split('//')[0] for l in src.split('\n'))
        for m in st.MACROS:
            assert m not in code, (f, m)
        if f in st.FILES:
            assert st.strip(src) == src, f
