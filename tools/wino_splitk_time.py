import sys, torch
sys.path.insert(0, '/root/repo')
from codeformer_amd import ops
def t_ms(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, H, cin, cout) in ((16, 16, 512, 512), (16, 32, 256, 256), (1, 16, 512, 512), (1, 32, 256, 256), (4, 16, 512, 512)):
    x = torch.randn(B, H, H, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.02
    sc, sh = torch.rand(B, cin, device='cuda') + 0.5, torch.randn(B, cin, device='cuda') * 0.1
    res = torch.randn(B, H, H, cout, device='cuda')
    pw = ops.pack_weight(w, None, bf16=ops.WINOGRAD)
    kw = dict(prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=res, emit_stats=True)
    row = []
    for sk in (0, 1, 2, 4):
        if sk and (cin // 128) % sk: continue
        row.append(f'split_k={sk}: {t_ms(lambda: ops.conv2d(x, pw, split_k=sk, **kw)) * 1e3:.1f} us')
    print(B, H, cin, cout, ' | '.join(row), flush=True)
