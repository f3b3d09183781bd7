"""Interleaved timing of Winograd-kernel variants (gpurun_ablate/lib_*.so) on representative shapes."""
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
B = 16
shapes = [(128, 128, 256), (64, 64, 512), (256, 256, 64), (512, 512, 16)]
if os.environ.get('AB_SHAPES'):
    shapes = [tuple(int(v) for v in item.split(',')) for item in os.environ['AB_SHAPES'].split(';')]
for cin, cout, H in shapes:
    x = torch.randn(B, H, H, cin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    bias = torch.randn(cout, device='cuda')
    pw = ops.pack_weight(w, bias, bf16=ops.WINOGRAD)
    pd = ops.pack_weight(w, bias)
    sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
    res = torch.randn(B, H, H, cout, device='cuda')
    out = torch.empty(B, H, H, cout, device='cuda')
    def desc(p, wino):
        return L.ConvDesc(in0=x.data_ptr(), c0=cin, batch=B, hin=H, win=H, hout=H, wout=H, cout=cout, cout_pad=p.cout_pad, taps=9,
                          stride=1, prologue=2, epilogue=1, pro_scale=sc.data_ptr(), pro_shift=sh.data_ptr(), weight=p.w.data_ptr(),
                          bias=p.bias.data_ptr(), res=res.data_ptr(), out=out.data_ptr(), winograd=wino)
    dw, dd = desc(pw, 1), desc(pd, 0)
    st = torch.cuda.current_stream().cuda_stream
    runs = [(k, l, dw) for k, l in libs.items()] + [('direct', next(iter(libs.values())), dd)]
    times = {k: [] for k, _, _ in runs}
    for k, l, d in runs:
        for _ in range(3):
            assert l.cf_conv2d(ctypes.byref(d), st) == 0
    torch.cuda.synchronize()
    for rnd in range(5):
        for k, l, d in runs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                l.cf_conv2d(ctypes.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * B * H * H * cout * cin * 9
    print(f'--- 3x3 {cin}->{cout} @{H} swish+res')
    for k, v in times.items():
        v = sorted(v)
        print(f'   {k:12s} median {v[len(v)//2]:.3f} ms   {fl / v[len(v)//2] / 1e9:6.1f} TF-equiv')
