#!/bin/bash
# Sample the shader clock / power while the big conv shape runs back-to-back (is the fp32-MFMA kernel clock-limited?)
python - <<'PY' &
import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from codeformer_amd import ops
x = torch.randn(16, 256, 256, 128, device='cuda'); w = torch.randn(128, 128, 3, 3, device='cuda') * 0.05
pw = ops.pack_weight(w, torch.randn(128, device='cuda'))
t0 = time.time()
n = 0
while time.time() - t0 < 14:
    for _ in range(50): ops.conv2d(x, pw)
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print(f'conv 128->128@256 B16: {dt/n*1e3:.3f} ms  {2*16*256*256*128*128*9/(dt/n)/1e12:.1f} TFLOP/s sustained over {dt:.0f}s')
PY
sleep 6
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4; sleep 1.5; done
wait
