"""Split count sweep of the split-K layers (16x16 latents on the four-wave Winograd kernel, the Transformer's Linear layers) per batch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
torch.manual_seed(0)
SUMS = []

def bits(t):
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum())

def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return sorted(ts)[2] * 1e3

for B in (1, 2, 4, 8, 16, 32):
    c, H = 512, 16
    x = torch.randn(B, H, H, c, device='cuda')
    pw = ops.pack_weight(torch.randn(c, c, 3, 3, device='cuda') * 0.05, torch.randn(c, device='cuda'), bf16=ops.WSPLIT)
    sc, sh = torch.rand(B, c, device='cuda') + 0.5, torch.randn(B, c, device='cuda') * 0.1
    r = torch.randn(B, H, H, c, device='cuda')
    line = [f'auto(sk{ops.splitk_for(pw, H, H, c, B)})']
    for sk in (1, 2, 4):
        kw = dict(prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=r, emit_stats=True, split_k=sk)
        line.append(f'sk{sk}: {timeit(lambda: ops.conv2d(x, pw, **kw)):6.1f}')
    print(f'B={B:2d} wino 512@16^2  ' + '  '.join(line), flush=True)
    for K, N in ((512, 512), (512, 1024), (1024, 512), (512, 1536)):
        for code, name in ((0, 'fp32'),):
            pl = ops.pack_weight(torch.randn(N, K, 1, 1, device='cuda') * 0.05, torch.randn(N, device='cuda'), bf16=code)
            xt = torch.randn(B, 16, 16, K, device='cuda')
            line = [f'auto(sk{ops.splitk_for(pl, 16, 16, K, B)})']
            for sk in (1, 2, 4, 8):
                if (K // 128) % sk:
                    continue
                line.append(f'sk{sk}: {timeit(lambda: ops.conv2d(xt, pl, split_k=sk)):6.1f}')
                SUMS.append(bits(ops.conv2d(xt, pl, split_k=sk)))
            print(f'B={B:2d} linear {K}->{N} {name}  ' + '  '.join(line), flush=True)
print('checksum of all linear outputs:', hex(sum(SUMS) & (2**64 - 1)))
