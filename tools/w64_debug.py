"""Where do outputs / statistics of two library builds differ?  (gpurun_ablate/lib_*.so, split-half Winograd, 64 output channels)"""
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
B, cin, cout, H = int(os.environ.get('DBG_B', 2)), int(os.environ.get('DBG_CIN', 64)), 64, int(os.environ.get('DBG_H', 128))
swish = int(os.environ.get('DBG_SWISH', 0))
torch.manual_seed(1)
x = torch.randn(B, H, H, cin, device='cuda')
pw = ops.pack_weight(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, torch.randn(cout, device='cuda'), bf16=ops.WSPLIT)
sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
res = torch.randn(B, H, H, cout, device='cuda')
nparts = 4 * (H // 8) * (H // 16)
outs = {}
for k, l in libs.items():
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    stats = torch.zeros(B, 32, nparts, 2, dtype=torch.float64, device='cuda')
    d = L.ConvDesc(in0=x.data_ptr(), c0=cin, batch=B, hin=H, win=H, hout=H, wout=H, cout=cout, cout_pad=64, taps=9, stride=1, prologue=2 if swish else 0,
                   epilogue=1 if swish else 0, pro_scale=sc.data_ptr(), pro_shift=sh.data_ptr(), weight=pw.w.data_ptr(), bias=pw.bias.data_ptr(),
                   res=res.data_ptr(), out=out.data_ptr(), bf16_mfma=ops.OPERAND_F16X2, winograd=1, acc_scale=1.0 / pw.scale,
                   stats_out=stats.data_ptr(), stats_cpg=2)
    assert l.cf_conv2d(ctypes.byref(d), torch.cuda.current_stream().cuda_stream) == 0, k
    torch.cuda.synchronize()
    outs[k] = (out.cpu(), stats.cpu())
names = list(outs)
a, b = outs[names[0]], outs[names[1]]
do = (a[0] - b[0]).abs()
print('out: nan', int(torch.isnan(b[0]).sum()), 'max diff', float(do.nan_to_num(1e9).max()), 'n diff', int((do > 0).sum()), 'of', do.numel())
if int((do > 0).sum()):
    idx = (do > 0).nonzero()
    print(' first diffs (b,y,x,c):', idx[:8].tolist(), ' y range', int(idx[:, 1].min()), int(idx[:, 1].max()), ' x range', int(idx[:, 2].min()), int(idx[:, 2].max()))
ds = (a[1] - b[1]).abs()
print('stats: n diff', int((ds > 0).sum()), 'of', ds.numel(), 'max', float(ds.max()))
if int((ds > 0).sum()):
    idx = (ds > 0).nonzero()
    print(' first (b,group,part,which):', idx[:12].tolist())
    i = tuple(idx[0].tolist())
    print(' values', float(a[1][i]), float(b[1][i]), ' zero entries in b:', int((b[1] == 0).sum()), 'in a:', int((a[1] == 0).sum()))
    print(' groups differing:', sorted(set(idx[:, 1].tolist()))[:40], ' parts%4:', sorted(set((idx[:, 2] % 4).tolist())))
