"""Per-stage shader-cycle sums of one workgroup of the F(4x4,3x3) kernel (library built with -DF4_TIMING=1 from a scratch copy of csrc/ with
tools/experiments/ablation_and_timing_macros.patch applied -- the stamps are not in the product sources since round 6; CF_LIB_PATH).
usage: CF_LIB_PATH=gpurun_ablate/lib_timing.so python tools/f43_timing.py [cin cout H] [fp32]     (fp32: the IEEE-fp32-operand form)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops, lib as L
FP32 = 'fp32' in sys.argv
sys.argv = [a for a in sys.argv if a != 'fp32']
cin, cout, H = (int(v) for v in (sys.argv[1:4] + ['64', '64', '512'][len(sys.argv) - 1:]))
B = 16
x = torch.randn(B, H, H, cin, device='cuda')
pw = ops.pack_weight(torch.randn(cout, cin, 3, 3, device='cuda') * 0.05, torch.randn(cout, device='cuda'), bf16=ops.WF43F if FP32 else ops.WF43)
sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
res = torch.randn(B, H, H, cout, device='cuda')
import time
for _ in range(3):
    y = ops.conv2d(x, pw, prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=res, emit_stats=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.conv2d(x, pw, prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=res, emit_stats=True)
e1.record()
torch.cuda.synchronize()
print(f'launch (instrumented build): {e0.elapsed_time(e1) / 10:.3f} ms' + ('  [one workgroup per CU]' if os.environ.get('CF_F43_ONE_WG') else ''))
raw = ctypes.CDLL(L.LIB_PATH)
buf = (ctypes.c_ulonglong * 256)()
assert raw.cf_debug_f4_timing(buf) == 0
names = ['fill', 'T work', 'T barrier', 'M work', 'M barrier', 'epi load+stage', 'epi barriers', 'epi compute+store']
n = int(buf[9])
if int(buf[10]):
    print(f'shader clock of the stamped workgroup: {int(buf[8]) / (int(buf[10]) * 10e-9) / 1e9:.3f} GHz ({int(buf[8])} cycles in {int(buf[10]) * 10e-3:.1f} us)')
print(f'{"fp32 operands " if FP32 else ""}{cin}->{cout} @ {H}x{H} x {B}: {n} slabs; shader cycles summed over the patch (per slab in brackets for the slab stages)')
print('wave  ' + '  '.join(f'{s:>17s}' for s in names) + '   total')
for w in range(16 if cout % 128 == 0 and not os.environ.get("CF_F43_NARROW") else 8):
    v = [int(buf[w * 16 + k]) for k in range(8)]
    print(f'  {w}:  ' + '  '.join((f'{t:9d} [{t // n:5d}]' if 1 <= k <= 4 else f'{t:17d}') for k, t in enumerate(v)) + f'   {int(buf[w * 16 + 8])}')
