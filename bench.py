"""bench.py -- aligned 512x512 faces/sec of the MI355X-native CodeFormer path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N=1: plain python; N>1: launched by torchrun, one rank per GPU)

A "step" is one CodeFormer.forward(x, w=0.5, adain=True) over one batch of 16 synthetic faces per GPU (config 2 of
BASELINE.json: seeded rand(16,3,512,512)*2-1, seed-0 random-init weights unless weights/CodeFormer/codeformer.pth exists)
followed, for N>1, by the single gather of the restored faces to rank 0 (left in flight while the next step computes; all K
gathers are joined inside the timed region).  Inputs are resident in HBM before the timed region; host PNG decode/encode is
outside the path and outside the timed region.  Weak scaling: 16 faces per GPU.

The headline (`value`, `ms_per_step`, `dtype`, `roofline`) is the IEEE-fp32 evaluation -- BASELINE config 2 to the letter
(--precision fp32, the default since round 6): every product of every convolution / Linear / attention on fp32 MFMA operands, fp32
tensors, fp32 accumulation (Winograd F(2x2,3x3) / F(4x4,3x3) where eligible: the same function in another summation order than
ATen's).  At N=1 the default run also gates and times two secondary legs and reports them as SCALARS (top level and, mirrored,
inside `config`, so that no consumer has to dig through nested objects):
  f16x2_*         the same step with precision='f16x2' (the module's product default): fp32 tensors / accumulation, the 3x3 / image-sized
                  1x1 / LayerNorm-bounded Linear products on split operands (fp32 = hi + lo IEEE halves, three f16 MFMAs per product:
                  22-bit operands -- NARROWER than IEEE fp32, which is why it is not the headline) behind the same golden gate;
  config3_rank_*  one rank's share of BASELINE config 3 (batch 16 per GPU, w = 0.7, precision 'bf16': bf16 STORAGE of the generator /
                  fusion-block activations + bf16 MFMA operands, fp32 accumulate; encoder / Transformer / argmax as in 'f16x2') behind
                  its own gate (indices exact, logits 1e-4, pixels within the derived bf16 gate of the reference golden at w = 0.7).
The whole JSON line stays under 2 KB; the long-form records (every kernel class of every leg, gate derivations, sample descriptions)
go to stderr with --details.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline:     the kernel class with the largest summed duration of a step, timed live per launch with events on the launch stream:
                `achieved` / `frac` against the dense peak of the MFMA type it issues (fp32: 157.3 TFLOP/s, f16: 2500; MI355X_MICROARCH.md).
                For a Winograd kernel on un-split operands (fewer multiplies than the direct form) achieved = EXECUTED MFMA FLOPs /
                duration (a hardware fraction, <= 1) and the algorithmic rate is `effective_tflops`.  `traffic_recorded` (= `traffic`) =
                HBM bytes per launch from the committed rocprofv3 PMC passes of the same precision mode (profiles/*_pmc_bench_<mode>.json;
                builder-box evidence, quoted only when that file's build id equals the loaded library's, null otherwise).
  cpu_baseline: the CPU oracle (oracle/codeformer_oracle.py -- kind "port": the restatement pinned to the reference's outputs, not the
                reference's own files, which do not exist on the GPU box; torch CPU fp32, up to 64 host threads) timed on a bounded
                sample (batch-1 forwards for ~10-30 s) on rank 0 at N=1.
  parity:       the golden-face gate that runs BEFORE the timed region: pixels / logits / code indices of the
                seeded face against the reference's committed output; the run aborts if it fails.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: dense f16 / bf16 MFMA (v_mfma_f32_32x32x16_f16); the sparse 5 PF figure is not used
HBM_PEAK_GBS = 8000.0
GFLOP_PER_FACE = 809.77            # BASELINE.md section 3 (restoration, w>0, 4 fuse levels): the REFERENCE algorithm's FLOPs
# The five Upsample blocks (nearest x2 + 3x3, 125.6 GFLOP/face in the reference's formulation) run as four 2x2 sub-pixel
# convolutions with folded taps: 4/9 of those MACs.  Hardware-utilisation figures use the EXECUTED count.
FUSED_MIN_GB_PER_FACE = 4.155


def build_net(device):
    import codeformer_amd.archs  # noqa: F401
    from codeformer_amd.utils.registry import ARCH_REGISTRY
    torch.manual_seed(0)
    net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9,
                                          connect_list=['32', '64', '128', '256']).eval()
    ckpt = os.path.join(ROOT, 'weights', 'CodeFormer', 'codeformer.pth')
    weights = 'seed-0 random-init'
    if os.path.exists(ckpt):
        net.load_state_dict(torch.load(ckpt, map_location='cpu')['params_ema'])
        weights = 'codeformer.pth'
    sd_cpu = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net.to(device), sd_cpu, weights


def recorded_traffic(prefixes=('wsplit_kernel',), precision='f16x2'):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_bench.json,
    produced by tools/pmc_bench.sh on this same bench step; bench.py cannot profile itself -- the file is stamped with the
    library's build id and ignored (traffic = null) when the kernels have changed since).  Per the MI355X guide:
    FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads, so
    bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, launch-weighted over the kernels whose name (without the
    `void (anonymous namespace)::` decoration) starts with one of `prefixes`."""
    import glob
    from codeformer_amd import lib
    # one file per precision mode: *_pmc_bench_<precision>.json (tools/pmc_bench.sh <tag> <precision>: the passes of bench.py --precision
    # <precision>); rounds 1-5 wrote the f16x2 step (then the default) to *_pmc_bench.json
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'*_pmc_bench_{precision}.json')))
    if precision == 'f16x2' and not files:
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_bench.json')))
    if not files:
        return None
    d = json.load(open(files[-1]))
    # Counters are only as good as the kernels they were taken from: the file carries the build id (source hash) of the library
    # that was profiled (tools/pmc_bench.sh); if the loaded library differs, nothing is quoted -- a stale number is worse than null.
    have, want = d.get('_meta', {}).get('cf_build_id'), lib.load().cf_build_id().decode()
    if have != want:
        return None
    n = tot = 0
    for k, v in d.items():
        if k == '_meta':
            continue
        name = k.replace('void ', '').replace('(anonymous namespace)::', '')
        if name.startswith(tuple(prefixes)) and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            ln = v['FETCH_SIZE']['launches']
            n += ln
            tot += ln * (2.0 * v['FETCH_SIZE']['mean'] + v['WRITE_SIZE']['mean']) * 1024.0
    return {'bytes_per_launch': round(tot / n), 'source': os.path.relpath(files[-1], ROOT)} if n else None


def wf43_names(f32):
    """Name prefixes of the instantiations wf43_kernel<PRO, EPI, NW, KS, F32, OVL> of one operand type (F32 is the fifth argument: no common prefix)."""
    return tuple(f'wf43_kernel<{p}, {e}, {nw}, {ks}, {f32},' for p in range(4) for e in range(3) for nw, ks in ((8, 16), (16, 16), (16, 32)))


def roofline_leg(net, x, w):
    """Per-launch event timing of every implicit-GEMM launch of one forward; returns the roofline object + a table."""
    from codeformer_amd import ops
    for _ in range(2):
        net(x, w=w, adain=True)
    torch.cuda.synchronize()
    reps = 3
    agg = {}
    shapes = {}
    for _ in range(reps):
        ops.PROFILE = []
        net(x, w=w, adain=True)
        torch.cuda.synchronize()
        rec, ops.PROFILE = ops.PROFILE, None
        for kind, flops, nbytes, e0, e1, shape in rec:
            a = agg.setdefault(kind, [0.0, 0.0, 0.0, 0])
            a[0] += flops
            a[1] += nbytes
            a[2] += e0.elapsed_time(e1) * 1e-3
            a[3] += 1
            sh = shapes.setdefault((kind,) + tuple(shape), [0.0, 0.0, 0])
            sh[0] += flops
            sh[1] += e0.elapsed_time(e1) * 1e-3
            sh[2] += 1
    table = {k: {'launches_per_forward': v[3] // reps, 'gflop_per_forward': v[0] / reps / 1e9,
                 'ms_per_forward': v[2] / reps * 1e3, 'tflops': v[0] / v[2] / 1e12, 'alg_gbs': v[1] / v[2] / 1e9}
             for k, v in agg.items()}
    table['by_shape (kind,B,H,W,Cin,Cout): launches, ms_total, TFLOP/s'] = {
        str(k): [v[2] // reps, round(v[1] / reps * 1e3, 3), round(v[0] / v[1] / 1e12, 1)]
        for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])}
    # kind -> (kernel, MFMA FLOPs executed per algorithmic FLOP booked by ops.conv2d, dense MFMA peak of the type it issues)
    KINDS = {
        'conv3x3_f16x2': ('split_conv_kernel<9,...> (3x3 s1, fp32 operands as hi+lo halves: 3 f16 MFMAs per product, fp32 accumulate)', 3.0, F16_MFMA_PEAK_TFLOPS, ('split_conv_kernel<9',)),
        'conv_up2x_f16x2': ('split_conv_kernel<4,...> (nearest-x2 + 3x3 folded to 2x2 sub-pixel taps, split halves)', 3.0, F16_MFMA_PEAK_TFLOPS, ('split_conv_kernel<4, 2, 2, false', 'split_conv_kernel<4, 1, 2, false')),
        'conv3x3_wino': ('winograd_kernel<.,false> (3x3 s1 as Winograd F(2x2,3x3), fp32 MFMA)', 4.0 / 9.0, FP32_MFMA_PEAK_TFLOPS, ('winograd_kernel<false, false', 'winograd_kernel<true, false')),
        'conv3x3_wino_f16x2_8w': ('wsplit_kernel (3x3 s1 as Winograd F(2x2,3x3), 128 channels per 8-wave workgroup; U and V as hi+lo halves: 3 f16 '
                                  'MFMAs per transform-domain product, fp32 accumulate)', 3.0 * 4.0 / 9.0, F16_MFMA_PEAK_TFLOPS, ('wsplit_kernel',)),
        'conv3x3_wino_f16_8w': ('wsplit_kernel<., F16> (Winograd F(2x2,3x3), single IEEE-half operands, one MFMA per product)', 4.0 / 9.0, F16_MFMA_PEAK_TFLOPS, ('wsplit_kernel',)),
        'conv3x3_wino_bf16_8w': ('wsplit_kernel<., BF16> (Winograd F(2x2,3x3), single bf16 operands, one MFMA per product)', 4.0 / 9.0, F16_MFMA_PEAK_TFLOPS, ('wsplit_kernel',)),
        'conv3x3_wino_f16x2': ('winograd_kernel<.,true> (the same on the four-wave 64-channel kernel: 64-channel layers and the 16x16 latents)',
                               3.0 * 4.0 / 9.0, F16_MFMA_PEAK_TFLOPS, ('winograd_kernel<false, true', 'winograd_kernel<true, true')),
        'conv3x3_wino43_f16x2': ('wf43_kernel (3x3 s1 as Winograd F(4x4,3x3) on 16x16 patches, generator / fusion layers only: 64 output channels per 8-wave workgroup (two per CU) '
                                 'or 128 per 16-wave workgroup; U and V as hi+lo halves: 36 transform-domain products per 16 outputs x 3 f16 MFMAs)', 3.0 * 2.25 / 9.0, F16_MFMA_PEAK_TFLOPS, wf43_names('false')),
        'conv3x3_wino43': ('wf43_kernel<..., true> (the same kernel with IEEE-fp32 operands: 36 transform-domain products per 16 outputs on v_mfma_f32_16x16x4_f32; '
                           "precision 'fp32')", 2.25 / 9.0, FP32_MFMA_PEAK_TFLOPS, wf43_names('true')),
        'conv3x3': ('igemm_kernel<9,1,...> (direct 3x3 s1 implicit GEMM, fp32 MFMA)', 1.0, FP32_MFMA_PEAK_TFLOPS, ('igemm_kernel<9, 1',)),
        'conv3x3_io': ('conv3x3_few_cin / conv3x3_few_cout (3->64 and 64->3 at 512x512 on the vector ALU: write- / read-bound)', 0.0, HBM_PEAK_GBS, ('conv3x3_few_c',)),
        'conv_up2x': ('igemm_kernel<4,1,...> (folded nearest-x2 + 3x3, fp32 MFMA)', 1.0, FP32_MFMA_PEAK_TFLOPS, ('igemm_kernel<4, 1',)),
        'conv3x3_s2': ('igemm_kernel<9,2,...> (3x3 stride 2, fp32 MFMA)', 1.0, FP32_MFMA_PEAK_TFLOPS, ('igemm_kernel<9, 2',)),
        'conv3x3_s2_f16x2': ('split_conv_kernel<4,...> (3x3 stride 2 as a 2x2 convolution of the space-to-depth input, split halves: 16 of which 9 tap blocks are non-zero)', 3.0 * 16.0 / 9.0, F16_MFMA_PEAK_TFLOPS, ('split_conv_kernel<4, 2, 2, true', 'split_conv_kernel<4, 1, 2, true')),
        'gemm1x1': ('igemm_kernel<1,1,...> (1x1 conv / Linear, fp32 MFMA)', 1.0, FP32_MFMA_PEAK_TFLOPS, ('igemm_kernel<1, 1',)),
        'conv1x1_stream_f16x2': ('split_conv_kernel<1,...> (1x1 skip convolutions on images of more than 1024 pixels, split halves: read-once / write-once streaming)', 0.0, HBM_PEAK_GBS, ('split_conv_kernel<1',)),
        'gemm1x1_f16x2': ('gemm_split_tile_kernel (the parameter-bounded Linear layers of the Transformer on split halves: A and pre-split B fragments straight from L2, 3 f16 MFMAs per product)', 3.0, F16_MFMA_PEAK_TFLOPS, ('gemm_split_tile_kernel', 'gemm_split_kernel', 'gemm_split_chunk_kernel')),
    }

    def entry(kind):
        """`achieved` / `frac` (= `frac_algorithmic`): the convolution's ALGORITHMIC FLOPs (2*taps*Cin*Cout per output pixel, SURVEY
        8(d)) of these launches over their summed durations, against the dense peak of the MFMA type the kernel issues.
        `executed_tflops` / `frac_executed`: the FLOPs the MFMA pipe really executes for them (Winograd F(2,3) 4/9 and F(4,3) 1/4 of the
        multiplies, times 3 for split operands) -- what the matrix pipe is busy with, not what the layer needed."""
        name, ratio, peak, pmc = KINDS.get(kind, (kind, 1.0, F16_MFMA_PEAK_TFLOPS if kind.endswith(('_f16', '_bf16', '_f16x2')) else FP32_MFMA_PEAK_TFLOPS, ('igemm_kernel<9, 1',)))
        c = agg[kind]
        tr = recorded_traffic(pmc, net.precision)
        common = {'kind': kind, 'avg_launch_ms': round(c[2] / c[3] * 1e3, 4), 'launches_per_step': c[3] // reps,
                  'algorithmic_gflop_per_step': round(c[0] / reps / 1e9, 1), 'ms_per_step': round(c[2] / reps * 1e3, 2),
                  'alg_bytes_per_launch': round(c[1] / c[3]), 'frac_hbm_peak_alg_bytes': round(c[1] / c[2] / 1e9 / HBM_PEAK_GBS, 4),
                  'traffic_recorded': tr['bytes_per_launch'] if tr else None, 'traffic': tr['bytes_per_launch'] if tr else None,
                  'traffic_source': tr['source'] if tr else None,    # (recorded on the builder's box by tools/pmc_bench.sh; quoted because its build id is the loaded library's)
                  'traffic_per_alg_bytes': round(tr['bytes_per_launch'] / (c[1] / c[3]), 3) if tr else None}
        if ratio == 0.0:   # no MFMA: algorithmic bytes / duration against the HBM peak
            gbs = c[1] / c[2] / 1e9
            return {'bound': 'hbm', 'kernel': name, 'achieved': round(gbs, 1), 'peak': peak, 'unit': 'GB/s', 'frac': round(gbs / peak, 4), **common}
        alg = c[0] / c[2] / 1e12
        # `frac` is a fraction of a hardware peak, so it must be <= 1 by construction: a Winograd kernel on un-split operands executes
        # FEWER MFMA FLOPs than the direct convolution it evaluates (ratio < 1) and its algorithmic rate can exceed the pipe's peak --
        # there `achieved` / `frac` are the EXECUTED rate and the algorithmic one is reported as `effective_tflops` / `frac_algorithmic`
        # (an "effective" figure, not a roofline fraction).  Where the kernel executes at least the algorithmic FLOPs (ratio >= 1: split
        # operands) `achieved` / `frac` stay the contract's algorithmic figure, which is then <= the executed one <= 1.
        lead = alg * min(ratio, 1.0)
        return {'bound': 'mfma', 'kernel': name, 'achieved': round(lead, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(lead / peak, 4),
                'achieved_is': 'executed MFMA FLOPs / duration (Winograd: fewer multiplies than the direct form)' if ratio < 1.0 else 'algorithmic FLOPs / duration',
                'effective_tflops': round(alg, 2), 'frac_algorithmic': round(alg / peak, 4), 'executed_tflops': round(alg * ratio, 2), 'frac_executed': round(alg * ratio / peak, 4),
                'executed_per_algorithmic_flop': round(ratio, 4), **common}

    order = sorted(agg, key=lambda k: -agg[k][2])
    roof = entry(order[0])                                   # the dominant kernel = the kind with the largest summed duration
    roof['other_kernels'] = {k: entry(k) for k in order[1:]}
    # executed fp32-MFMA-equivalent work of one step (Winograd at its 4/9; a split-half product counted once) + attention (2.01 GF / face)
    exec_flops = sum(v[0] * (0.25 if k.startswith('conv3x3_wino43') else 4.0 / 9.0 if k.startswith('conv3x3_wino') else 1.0) for k, v in agg.items()) / reps + 2.01e9 * x.shape[0]
    roof['executed_gflop_per_face_whole_path'] = round(exec_flops / x.shape[0] / 1e9, 2)
    return roof, table


def parity_gate(net, sd_cpu, weights, w):
    """nothing is timed before the path has reproduced the reference.  Seed-0 weights: the committed golden
    of the REFERENCE's forward on the seeded face (tests/golden/restoration_seed0_face0.npz: pixels 1e-3, logits 1e-4, code indices
    exact); other weights (a real checkpoint): the CPU oracle on the same face."""
    import numpy as np
    from oracle.synth import seeded_input
    x = seeded_input(1)
    out, logits, _ = net(x.to(next(net.parameters()).device), w=w, adain=True)
    torch.cuda.synchronize()
    idx = net.last_indices.cpu().view(-1)
    gold = os.path.join(ROOT, 'tests', 'golden', 'restoration_seed0_face0.npz')
    if weights == 'seed-0 random-init' and w == 0.5 and os.path.exists(gold):
        g = np.load(gold)
        ref_out, ref_logits, ref_idx, src = torch.from_numpy(g['out']), torch.from_numpy(g['logits']), torch.from_numpy(g['idx']).view(-1), 'reference golden (tests/golden/restoration_seed0_face0.npz)'
    else:
        from oracle import codeformer_oracle as O
        ref_out, ref_logits, _, ref_idx = O.codeformer_forward(x, sd_cpu, w=w, adain_flag=True, return_idx=True)
        ref_idx, src = ref_idx.view(-1), 'CPU oracle (oracle/codeformer_oracle.py)'
    res = {'against': src, 'max_abs_pixel_diff': float((out.cpu() - ref_out).abs().max()), 'max_abs_logit_diff': float((logits.cpu() - ref_logits).abs().max()),
           'code_indices_equal': bool(torch.equal(idx, ref_idx)), 'tolerances': 'pixels 1e-3, logits 1e-4, code indices exact'}
    if not (res['max_abs_pixel_diff'] <= 1e-3 and res['max_abs_logit_diff'] <= 1e-4 and res['code_indices_equal']):
        raise SystemExit(f'bench.py: parity gate FAILED, nothing timed: {json.dumps(res)}')
    return res


def config3_gate(net, weights):
    """Gate of the bf16 / w = 0.7 leg (tests/test_gpu_real_images.py:test_config3_fidelity_weight_through_the_network, tools/gpu_check.py:g_bf16):
    the seeded face at w = 0.7 with net.precision = 'bf16' against the REFERENCE's committed outputs -- code indices exact, logits 1e-4
    (encoder and Transformer do not run on bf16), pixels within the stated bf16 gate (max 0.196, mean 0.0152 on outputs of std ~0.5).
    Where the gate comes from (tools/bf16_gate_derivation.py, profiles/r06_bf16_gate_derivation.txt): the CPU oracle with the operands of the
    same 58 convolutions rounded to bf16 AND the 46 generator / fusion activations of more than 1024 pixels stored as bf16 (rounded once where
    they are written, fp32 accumulation) differs from the reference's fp32 output on this face by max 0.1307 / mean 0.01215 -- the cost of
    that arithmetic for ANY implementation with these weights (operands alone, the mode of rounds 2-5: 0.1094 / 0.01064).  The gate is 1.5x /
    1.25x that intrinsic cost."""
    import numpy as np
    from oracle.synth import seeded_input
    g7 = os.path.join(ROOT, 'tests', 'golden', 'restoration_seed0_face0_w0.7.npz')
    g5 = os.path.join(ROOT, 'tests', 'golden', 'restoration_seed0_face0.npz')
    if weights != 'seed-0 random-init' or not (os.path.exists(g7) and os.path.exists(g5)):
        return {'against': None, 'note': 'no committed reference output for these weights at w=0.7: leg timed without a gate'}
    g, g0 = np.load(g7), np.load(g5)
    out, logits, _ = net(seeded_input(1).to(next(net.parameters()).device), w=0.7, adain=True)
    torch.cuda.synchronize()
    d = (out.cpu()[:, :, ::4, ::4] - torch.from_numpy(g['out_sub'])).abs()
    storage = bool(getattr(net, 'bf16_storage', False))
    res = {'against': 'reference golden (tests/golden/restoration_seed0_face0_w0.7.npz, every 4th pixel; logits / indices of restoration_seed0_face0.npz)',
           'bf16_storage': storage,
           'max_abs_pixel_diff': float(d.max()), 'mean_abs_pixel_diff': float(d.mean()),
           'max_abs_logit_diff': float((logits.cpu() - torch.from_numpy(g0['logits'])).abs().max()),
           'code_indices_equal': bool(np.array_equal(net.last_indices.cpu().numpy().reshape(-1), g0['idx'].reshape(-1))),
           'tolerances': 'pixels max 0.196 / mean 0.0152 (bf16 operands + bf16 storage), logits 1e-4, code indices exact',
           'gate_derivation': 'CPU oracle with bf16-rounded operands in the same 58 convolutions and bf16 storage of the 46 activations of more than 1024 pixels vs the '
                              'reference fp32 output: max 0.1307 / mean 0.01215 (intrinsic cost, profiles/r06_bf16_gate_derivation.txt); gate = 1.5x / 1.25x of it'}
    res['x_intrinsic_bf16_cost'] = [round(res['max_abs_pixel_diff'] / (0.1307 if storage else 0.1094), 3), round(res['mean_abs_pixel_diff'] / (0.01215 if storage else 0.01064), 3)]
    if not (res['max_abs_pixel_diff'] <= 0.196 and res['mean_abs_pixel_diff'] <= 0.0152 and res['max_abs_logit_diff'] <= 1e-4 and res['code_indices_equal']):
        raise SystemExit(f'bench.py: config-3 (bf16, w=0.7) gate FAILED, leg not timed: {json.dumps(res)}')
    return res


def cpu_baseline_leg(sd_cpu, w, budget_s=25.0):
    """CPU oracle, batch-1 forwards, bounded: at most ~budget_s of timed work (>= 1 forward) after one warm-up.
    Threads = the CPUs this process may run on, capped at 64 (torch's intra-op pool stops scaling well before that on
    conv workloads and oversubscribed boxes get dramatically slower); the count used is reported in `cores`."""
    from oracle import codeformer_oracle as O
    from oracle.synth import seeded_input
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 64))
    torch.set_num_threads(cores)
    x = seeded_input(1)
    t0 = time.perf_counter()
    O.codeformer_forward(x, sd_cpu, w=w, adain_flag=True)  # warm-up (first call pays oneDNN primitive creation)
    warm = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 + warm * 0.8 < budget_s and n < 16):
        O.codeformer_forward(x, sd_cpu, w=w, adain_flag=True)
        n += 1
    dt = time.perf_counter() - t0
    return {'value': round(n / dt, 4), 'unit': 'faces/s', 'cores': cores, 'kind': 'port',
            'sample_short': f'{n} batch-1 oracle forwards, torch CPU fp32, {cores} threads, w={w}',
            'sample': f'{n} batch-1 forward(s) of the CPU oracle (torch {torch.__version__} CPU fp32, {cores} threads of {avail} '
                      f'available) on the seeded 512x512 input, w={w}, adain=True, after 1 warm-up ({warm:.1f} s)'}


SHORT_KERNEL = {   # <= 110 characters: the driver's record truncates longer strings
    'conv3x3_wino43': 'wf43_kernel<..,F32>: 3x3 s1 as Winograd F(4x4,3x3), IEEE-fp32 operands on v_mfma_f32_16x16x4_f32',
    'conv3x3_wino43_f16x2': 'wf43_kernel: 3x3 s1 as Winograd F(4x4,3x3), hi+lo f16 operands (3 MFMAs per product), fp32 accumulate',
    'conv3x3_wino': 'winograd_kernel<.,false>: 3x3 s1 as Winograd F(2x2,3x3), fp32 MFMA',
    'conv3x3_wino_bf16_8w': 'wsplit_kernel<.,BF16>: Winograd F(2x2,3x3), bf16 operands, fp32 accumulate',
    'conv_up2x': 'igemm_kernel<4,1>: nearest-x2 + 3x3 folded to 2x2 sub-pixel taps, fp32 MFMA',
}


def compact_roofline(r):
    """The line's `roofline`: scalars and short strings only (the full record, `other_kernels` included, goes to --details)."""
    keep = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_recorded', 'traffic_source', 'traffic_per_alg_bytes', 'avg_launch_ms',
            'launches_per_step', 'ms_per_step', 'alg_bytes_per_launch', 'effective_tflops')
    # (executed_tflops / frac_executed of the long form equal achieved / frac for the Winograd classes: not repeated in the line)
    out = {'kernel': SHORT_KERNEL.get(r.get('kind'), r['kernel'])[:110]}
    out.update({k: r[k] for k in keep if k in r})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-per-gpu', type=int, default=16)
    ap.add_argument('--w', type=float, default=0.5)
    ap.add_argument('--precision', choices=['fp32', 'f16x2', 'bf16', 'fp16'], default='fp32',
                    help="arithmetic of the headline: fp32 (default) = IEEE-fp32 MFMA operands everywhere, BASELINE config 2 to the letter; "
                         "f16x2 = fp32 operands split into hi+lo IEEE halves (22-bit operands, the module's product default); bf16 / fp16 = 16-bit "
                         "operands (bf16: also 16-bit storage) in generator + CFT (BASELINE configs 3/5), encoder on split halves")
    ap.add_argument('--no-parity-gate', action='store_true', help='skip the golden-face check that precedes the timed region')
    ap.add_argument('--no-f16x2-leg', '--no-exact-leg', dest='no_f16x2_leg', action='store_true', help="skip the secondary precision='f16x2' timing of the default run")
    ap.add_argument('--no-config3-leg', action='store_true', help='skip the bf16 / w=0.7 timing (one rank of BASELINE config 3) of the default run')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--details', action='store_true', help='print the long-form record (every kernel class of every leg, gates, samples) to stderr')
    args = ap.parse_args()

    import torch.distributed as dist
    from codeformer_amd import lib, parallel
    from oracle.synth import seeded_input
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (there is no CPU execution path to benchmark)'
    lib.load()
    rank, world, dev = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus}')

    net, sd_cpu, weights = build_net(dev)
    net.precision = args.precision
    B = args.batch_per_gpu
    x = seeded_input(B, seed=1234 + rank).to(dev)     # weak scaling: every rank restores its own 16 faces
    total = B * world
    details = {}

    # One gather is kept in flight: step i's restored faces travel to rank 0 (RCCL, its own stream) while step i+1 computes.
    # Every gather of the timed steps is joined before the closing synchronize, so the K steps are complete inside the bracket.
    parity = None
    if rank == 0 and not args.no_parity_gate and args.precision in ('fp32', 'f16x2'):
        parity = parity_gate(net, sd_cpu, weights, args.w)     # raises before anything is timed if the path does not match the reference
    pending = [None]

    def step():
        handle, _ = parallel.restore_sharded_async(net, x, total, w=args.w, adain=True, dst=0)
        prev, pending[0] = pending[0], handle
        return prev.wait() if prev is not None else None

    def drain():
        faces = pending[0].wait() if pending[0] is not None else None
        pending[0] = None
        return faces

    for _ in range(args.warmup):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        faces_per_s = args.steps * total / dt
        config2 = args.w == 0.5 and B == 16
        workload = {'fp32': 'BASELINE config 2 (IEEE fp32 arithmetic)', 'f16x2': 'BASELINE config 2 shapes, split-half f16 operands (22-bit, NOT IEEE fp32)',
                    'bf16': 'BASELINE config 3 rank share (bf16 storage + operands in generator/CFT)'}.get(args.precision, 'custom') if config2 or args.precision == 'bf16' else 'custom'
        line = {
            'metric': 'aligned 512x512 faces/sec (whole node) at w=0.5', 'value': round(faces_per_s, 2), 'unit': 'faces/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'f16x2': 'f32 tensors/accumulate, f16 hi+lo operands (22-bit)', 'bf16': 'bf16 storage+operands / f32 accumulate (generator+CFT)',
                      'fp16': 'f16 operands / f32 accumulate (generator+CFT)'}[args.precision],
            'data': 'synthetic',
            'config': {'workload': f'{workload}: {B} faces/GPU, w={args.w}, adain, {weights} weights', 'global_batch': total,
                       'parallelism': f'faces sharded x{world}, one gather to rank 0' if world > 1 else 'single GPU', 'precision': args.precision},
            'whole_path': {'effective_tflops_reference_flop_count': round(faces_per_s * GFLOP_PER_FACE / 1e3, 2),
                           'frac_hbm_peak_fused_min_bytes': round(faces_per_s * FUSED_MIN_GB_PER_FACE / (HBM_PEAK_GBS * world), 4)},
        }
        if parity is not None:
            line['parity'] = {k: parity[k] for k in ('max_abs_pixel_diff', 'max_abs_logit_diff', 'code_indices_equal')}
            line['parity']['against'] = 'reference golden' if 'golden' in parity['against'] else 'CPU oracle'
            details['parity'] = parity
        if not args.no_roofline:
            roof, table = roofline_leg(net, x, args.w)
            ex = roof.pop('executed_gflop_per_face_whole_path')      # products really evaluated (folded upsample taps, Winograd 4/9 / 1/4)
            details['roofline'] = roof
            details['per_class_table'] = table
            line['roofline'] = compact_roofline(roof)
            line['whole_path']['evaluated_gflop_per_face'] = ex
            if args.precision == 'fp32':                             # every product on the fp32 pipe: a whole-path fraction makes sense
                line['whole_path']['executed_tflops_fp32'] = round(faces_per_s * ex / 1e3, 2)
                line['whole_path']['frac_fp32_mfma_peak'] = round(faces_per_s * ex / 1e3 / (FP32_MFMA_PEAK_TFLOPS * world), 4)
            details['whole_path'] = dict(line['whole_path'])
            line['whole_path'] = {k: v for k, v in line['whole_path'].items() if k in ('frac_fp32_mfma_peak', 'frac_hbm_peak_fused_min_bytes')}

        def timed(w):
            for _ in range(args.warmup):
                net(x, w=w, adain=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                net(x, w=w, adain=True)
            torch.cuda.synchronize()
            return time.perf_counter() - t1

        def leg_roofline(w):
            r, _ = roofline_leg(net, x, w)
            r.pop('executed_gflop_per_face_whole_path')
            return r

        def mirror(key, val):   # secondary-leg scalars at top level; the four throughput figures also inside `config` (the driver's parsed record keeps that object whole)
            line[key] = val
            if key.endswith(('_faces_per_s', '_ms_per_step')):
                line['config'][key] = val

        secondary = world == 1 and args.precision == 'fp32' and config2
        if secondary and not args.no_f16x2_leg:
            y_exact = net(x, w=args.w, adain=True)
            net.precision = 'f16x2'
            gate2 = None if args.no_parity_gate else parity_gate(net, sd_cpu, weights, args.w)   # the same gate as the headline, before this leg is timed
            y_split = net(x, w=args.w, adain=True)
            dt1 = timed(args.w)
            mirror('f16x2_faces_per_s', round(args.steps * total / dt1, 2))
            mirror('f16x2_ms_per_step', round(dt1 / args.steps * 1e3, 3))
            if gate2 is not None:
                mirror('f16x2_max_abs_pixel_diff', gate2['max_abs_pixel_diff'])
                mirror('f16x2_max_abs_logit_diff', gate2['max_abs_logit_diff'])
                mirror('f16x2_code_indices_equal', gate2['code_indices_equal'])
            details['f16x2'] = {'what': "the same step with precision='f16x2' (split-half operands: 22-bit, narrower than IEEE fp32)", 'parity': gate2,
                                'max_abs_pixel_diff_vs_fp32_leg': float((y_split[0] - y_exact[0]).abs().max()),
                                'max_abs_logit_diff_vs_fp32_leg': float((y_split[1] - y_exact[1]).abs().max()),
                                'code_indices_equal_vs_fp32_leg': bool(torch.equal(y_split[1].argmax(-1), y_exact[1].argmax(-1)))}
            if args.details and not args.no_roofline:
                details['f16x2']['roofline'] = leg_roofline(args.w)
            net.precision = args.precision
        if secondary and not args.no_config3_leg:
            net.precision = 'bf16'
            gate = config3_gate(net, weights)      # raises before anything is timed
            dt3 = timed(0.7)
            mirror('config3_rank_faces_per_s', round(args.steps * total / dt3, 2))
            mirror('config3_rank_ms_per_step', round(dt3 / args.steps * 1e3, 3))
            for k_src, k_dst in (('max_abs_pixel_diff', 'config3_max_abs_pixel_diff'), ('mean_abs_pixel_diff', 'config3_mean_abs_pixel_diff'),
                                 ('max_abs_logit_diff', 'config3_max_abs_logit_diff'), ('code_indices_equal', 'config3_code_indices_equal')):
                if k_src in gate:
                    mirror(k_dst, gate[k_src] if isinstance(gate[k_src], bool) else float(f'{gate[k_src]:.4g}'))
            details['config3_rank'] = {'what': "one rank's share of BASELINE config 3 (batch 16 per GPU, w=0.7, precision=bf16: bf16 storage of the generator / fusion "
                                               'activations from 32x32 up + bf16 MFMA operands, f32 accumulate; encoder and Transformer as in f16x2)', 'gate': gate}
            if args.details and not args.no_roofline:
                details['config3_rank']['roofline'] = leg_roofline(0.7)
            net.precision = args.precision
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline_leg(sd_cpu, args.w)
            details['cpu_baseline'] = dict(cb)
            cb['sample'] = cb['sample_short']
            line['cpu_baseline'] = cb
            del cb['sample_short'], details['cpu_baseline']['sample_short']
        for k in ('max_abs_pixel_diff', 'max_abs_logit_diff'):   # 4 significant digits keep the line short
            if 'parity' in line:
                line['parity'][k] = float(f"{line['parity'][k]:.4g}")
        for k in ('f16x2_max_abs_pixel_diff', 'f16x2_max_abs_logit_diff'):
            if k in line:
                mirror(k, float(f'{line[k]:.4g}'))
        out = json.dumps(line)
        if len(out) > 2040:   # the driver keeps a 2 KB tail of stdout: drop the mirrors inside `config` before anything is cut blindly
            for k in [k for k in line['config'] if k.startswith(('f16x2_', 'config3_'))]:
                del line['config'][k]
            out = json.dumps(line)
        if args.details:
            print(json.dumps(details, indent=1), file=sys.stderr)
        print(out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
