"""Average socket power and shader clock (rocm-smi samples) while one layer runs in a loop -- the evidence behind "power-limited" (DESIGN.md section 3):
the 128 -> 128 @256x256 x16 F(4,3) launch in the form CF_F43_WIDE selects, next to a memory-only kernel (a 1 GB device copy) for scale.
GPU box only.   usage: CF_F43_WIDE=k32|k16|ovl python tools/power_probe.py [fp32]      (fp32: the IEEE-fp32-operand form, plus the direct fp32 kernels of the
same step: folded upsample, 1x1 skip, token GEMM)"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util  # noqa: E402

from codeformer_amd import ops  # noqa: E402

spec = importlib.util.spec_from_file_location('f43_check', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'f43_check.py'))
fc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fc)


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            pw = next((float(v) for k, v in card.items() if 'Power' in k and 'W' in k), None)
            sclk = next((v for k, v in card.items() if k.startswith('sclk')), None)
            out.append((pw, sclk))
        except Exception as e:  # noqa: BLE001
            out.append((None, str(e)[:60]))
        time.sleep(0.05)


def run(name, fn, seconds=4.0):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    pws = [p for p, _ in out if p is not None]
    clk = [c for _, c in out if c]
    print(f'{name:34s} {dt / n * 1e3:7.3f} ms per launch   power avg {sum(pws) / max(len(pws), 1):6.0f} W (max {max(pws, default=0):.0f}, {len(pws)} samples)   sclk samples {clk[len(clk) // 2:len(clk) // 2 + 3]}', flush=True)


mode = os.environ.get('CF_F43_WIDE', 'default')
c = dict(B=16, H=256, W=256, cin=128, cout=128, prologue=ops.PRO_AFFINE_SWISH, epilogue=ops.EPI_RESIDUAL, stats=True, seed=9)
x1, x2, w, b, kw, _ = fc.make(ref=False, **c)
FP32 = 'fp32' in sys.argv
pw = ops.pack_weight(w, b, bf16=ops.WF43F if FP32 else ops.WF43)
if FP32:
    kw.pop('act', None)
run(f'F(4,3){" fp32 operands" if FP32 else ""} 128->128 @256^2 x16 [{mode}]', lambda: ops.conv2d(x1, pw, x2=x2, **kw))
if FP32:
    c64 = dict(c, H=512, W=512, cin=64, cout=64)
    y1, y2, w64, b64, kw64, _ = fc.make(ref=False, **c64)
    kw64.pop('act', None)
    pw64 = ops.pack_weight(w64, b64, bf16=ops.WF43F)
    run('F(4,3) fp32 operands 64->64 @512^2 x16', lambda: ops.conv2d(y1, pw64, **kw64))
    del y1, y2
    xu = torch.randn(16, 256, 256, 128, device='cuda')
    pwu = ops.pack_weight(w, b, up2x=True)
    run('folded upsample fp32 128ch 256->512 x16', lambda: ops.conv2d(xu, pwu, upsample=True, emit_stats=True))
    del xu
    xt = torch.randn(16, 16, 16, 512, device='cuda')
    pwt = ops.pack_weight(torch.randn(1024, 512, device='cuda') * 0.05, torch.randn(1024, device='cuda'))
    run('token GEMM fp32 4096 x 512 -> 1024', lambda: ops.conv2d(xt, pwt))
src = torch.empty(256 << 20, dtype=torch.float32, device='cuda')
dst = torch.empty_like(src)
run('device copy 1 GiB', lambda: dst.copy_(src))
