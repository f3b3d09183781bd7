"""Interleaved A/B timing of libcodeformer_hip variants (gpurun_ablate/lib_*.so) on the split-half conv kernel.
All variants are loaded in one process through ctypes; rounds alternate variants so clock drift hits them equally."""
import ctypes, glob, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codeformer_amd import lib as L, ops

libs = {}
for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_ablate', 'lib_*.so'))):
    l = ctypes.CDLL(f)
    l.cf_conv2d.restype = ctypes.c_int
    l.cf_conv2d.argtypes = [ctypes.POINTER(L.ConvDesc), ctypes.c_void_p]
    libs[os.path.basename(f)[4:-3]] = l
B = int(os.environ.get('AB_BATCH', 16))
KIND = os.environ.get('AB_KIND', 'split')   # split | wino (fp32 Winograd) | wsplit (split-half Winograd) | wf16 / wbf16 (single-operand Winograd)
WINO = KIND in ('wino', 'wsplit', 'wf16', 'wbf16')
shapes = [(128, 128, 256, 1, 0), (64, 64, 512, 1, 0), (256, 256, 64, 1, 0), (128, 128, 256, 0, 0), (128, 128, 256, 0, 1)]
if os.environ.get('AB_SHAPES'):   # "cin,cout,H,swish,up;..."
    shapes = [tuple(int(v) for v in item.split(',')) for item in os.environ['AB_SHAPES'].split(';')]
for cin, cout, H, swish, up in shapes:
    x = torch.randn(B, H, H, cin, device='cuda')
    Ho = 2 * H if up else H
    pw = ops.pack_weight(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, torch.randn(cout, device='cuda'), bf16={'wino': ops.WINOGRAD, 'wsplit': ops.WSPLIT, 'wf16': ops.WF16, 'wbf16': ops.WBF16}.get(KIND, ops.SPLIT), up2x=bool(up))
    sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
    res = torch.randn(B, Ho, Ho, cout, device='cuda')
    out = torch.empty(B, Ho, Ho, cout, device='cuda')
    stats = torch.empty(B * 32 * 4 * (Ho // 8) * (Ho // 8) * 2, dtype=torch.float64, device='cuda')
    d = L.ConvDesc(in0=x.data_ptr(), c0=cin, batch=B, hin=H, win=H, hout=Ho, wout=Ho, cout=cout, cout_pad=pw.cout_pad, taps=9,
                   stride=1, upsample=up, prologue=2 if swish else 0, epilogue=1 if swish else 0, pro_scale=sc.data_ptr(),
                   pro_shift=sh.data_ptr(), weight=pw.w.data_ptr(), bias=pw.bias.data_ptr(), res=res.data_ptr(), out=out.data_ptr(),
                   bf16_mfma={'wino': 0, 'wf16': 2, 'wbf16': 1}.get(KIND, ops.OPERAND_F16X2), winograd=int(WINO), acc_scale=1.0 / pw.scale, stats_out=stats.data_ptr(),
                   stats_cpg=cout // 32)
    st = torch.cuda.current_stream().cuda_stream
    times = {k: [] for k in libs}
    for k, l in libs.items():
        for _ in range(3):
            assert l.cf_conv2d(ctypes.byref(d), st) == 0, k
    torch.cuda.synchronize()
    if os.environ.get('AB_COMPARE'):   # outputs and statistics partials of every variant against the first one, bitwise
        ref = None
        for k, l in libs.items():
            out.fill_(float('nan')); stats.zero_()
            assert l.cf_conv2d(ctypes.byref(d), st) == 0, k
            torch.cuda.synchronize()
            cur = (out.clone(), stats.clone())
            if ref is None:
                ref = cur
            else:
                print(f'   compare {k}: out equal {torch.equal(cur[0], ref[0])} nan {int(torch.isnan(cur[0]).sum())} stats equal {torch.equal(cur[1], ref[1])}', flush=True)
    for rnd in range(5):
        for k, l in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                l.cf_conv2d(ctypes.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * B * Ho * Ho * cout * cin * 9
    print(f'--- {cin}->{cout} @{H}{" up2x" if up else ""} {"swish+res+stats" if swish else "plain+stats"}')
    for k, v in times.items():
        v = sorted(v)
        print(f'   {k:14s} median {v[len(v)//2]:.3f} ms  min {v[0]:.3f} ms   {fl / v[len(v)//2] / 1e9:6.1f} TF-equiv (median)', flush=True)
