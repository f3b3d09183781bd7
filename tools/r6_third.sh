#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== bf16 storage tests"; timeout 900 python -m pytest tests/test_gpu_bf16_storage.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25
S="--no-cpu-baseline --no-f16x2-leg --no-config3-leg --no-roofline --steps 10 --warmup 3"
echo "== bf16 storage bench"; timeout 300 python bench.py --precision bf16 --w 0.7 $S 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "== bf16 operands-only bench"; CODEFORMER_HIP_BF16_STORAGE=0 timeout 300 python bench.py --precision bf16 --w 0.7 $S 2>&1 | grep -v amdgpu.ids | cut -c1-260
echo "== details (storage)"; timeout 300 python bench.py --precision bf16 --w 0.7 --no-cpu-baseline --details --steps 5 --warmup 2 2> gpurun_out/r6_bf16_storage_details.txt | cut -c1-100
python - <<'PY'
import json,re
t=open('gpurun_out/r6_bf16_storage_details.txt').read()
i=t.index('{')
d=json.loads(t[i:])
r=d['roofline']
print(r['kind'], r['ms_per_step'], r['avg_launch_ms'])
for k,v in r['other_kernels'].items(): print(k, v['ms_per_step'], v['launches_per_step'], v['avg_launch_ms'], v.get('frac_hbm_peak_alg_bytes'))
PY
