"""Split-K Winograd probe: the 256-channel 32x32 / 64x64 layers and the 512-channel latents on the eight-wave kernel (split_k=0) and on
the four-wave split-K kernel with 1..8 workgroups per tile, at B = 1 and B = 16 (swish prologue + residual + statistics)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
for B in (1, 2, 16):
    for c, H in ((256, 32), (256, 64), (128, 128), (512, 16)):
        x = torch.randn(B, H, H, c, device='cuda')
        w = torch.randn(c, c, 3, 3, device='cuda') * 0.05
        b = torch.randn(c, device='cuda')
        sc, sh = torch.rand(B, c, device='cuda') + 0.5, torch.randn(B, c, device='cuda') * 0.1
        r = torch.randn(B, H, H, c, device='cuda')
        pw = ops.pack_weight(w, b, bf16=ops.WSPLIT)
        line = []
        for sk in (0, 1, 2, 4, 8):
            if sk and (c // 128) % sk:
                continue
            kw = dict(prologue=ops.PRO_AFFINE_SWISH, scale=sc, shift=sh, epilogue=ops.EPI_RESIDUAL, res=r, emit_stats=True, split_k=sk)
            try:
                for _ in range(3):
                    y = ops.conv2d(x, pw, **kw)
            except Exception as e:
                line.append(f'sk{sk}: {str(e)[:60]}')
                continue
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    y = ops.conv2d(x, pw, **kw)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            line.append(f'sk{sk}: {sorted(ts)[2] * 1e3:6.1f} us')
        print(f'B={B:2d} {c}ch @{H}^2: ' + '  '.join(line), flush=True)
