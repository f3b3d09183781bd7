"""First (3->64 from NCHW) and last (64->3 to NCHW) convolution of the network: time + bitwise digest (A/B of library builds: CF_LIB_PATH)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
B = int(os.environ.get('B', 16))
torch.manual_seed(0)
x = torch.rand(B, 3, 512, 512, device='cuda') * 2 - 1
w1, b1 = torch.randn(64, 3, 3, 3, device='cuda') * 0.2, torch.randn(64, device='cuda') * 0.1
y = torch.randn(B, 512, 512, 64, device='cuda')
w2, b2 = torch.randn(3, 64, 3, 3, device='cuda') * 0.05, torch.randn(3, device='cuda') * 0.1
sc, sh = torch.rand(B, 64, device='cuda') + 0.5, torch.randn(B, 64, device='cuda') * 0.1
pw1, pw2 = ops.pack_weight(w1, b1), ops.pack_weight(w2, b2)
def t(fn):
    for _ in range(3): r = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): r = fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    return sorted(ts)[2], r
t1, o1 = t(lambda: ops.conv2d(x, pw1, in_nchw=True, emit_stats=True))
t2, o2 = t(lambda: ops.conv2d(y, pw2, out_nchw=True, prologue=ops.PRO_AFFINE, scale=sc, shift=sh))
dig = lambda z: hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()[:12]
print(f'first conv {t1:.3f} ms ({o1.numel() * 4 / t1 / 1e9:.2f} TB/s written) out {dig(o1)} stats {dig(o1._cf_stats.part)} | last conv {t2:.3f} ms ({y.numel() * 4 / t2 / 1e9:.2f} TB/s read) out {dig(o2)}')
