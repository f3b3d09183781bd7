#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== test"; timeout 600 python -m pytest tests/test_gpu_split.py -x -q -s -k "upsampling_gather or f43" 2>&1 | grep -v amdgpu.ids | tail -12
echo "== fp32 goldens"; timeout 900 python -m pytest tests/test_gpu_real_images.py tests/test_gpu_parity.py -x -q -k "fp32 or net or config3 or margin or batch16" 2>&1 | tail -4
S="--no-cpu-baseline --no-f16x2-leg --no-config3-leg --no-roofline --steps 10 --warmup 3"
echo "== fp32 bench, F43 upsample on"; timeout 300 python bench.py $S 2>/dev/null | cut -c1-260
echo "== fp32 bench, F43 upsample off"; CODEFORMER_HIP_F43_UPSAMPLE=0 timeout 300 python bench.py $S 2>/dev/null | cut -c1-260
echo "== on again"; timeout 300 python bench.py $S 2>/dev/null | cut -c1-200
echo "== details"; timeout 300 python bench.py --no-cpu-baseline --no-f16x2-leg --no-config3-leg --details --steps 5 --warmup 2 2> gpurun_out/r6_fp32_details.txt | cut -c1-100
python - <<'PY'
import json
t=open('gpurun_out/r6_fp32_details.txt').read()
d=json.loads(t[t.index('{'):])
r=d['roofline']
print(r['kind'], r['ms_per_step'], r['launches_per_step'], r['frac'])
for k,v in r['other_kernels'].items(): print(k, v['ms_per_step'], v['launches_per_step'], v['avg_launch_ms'], v.get('frac'))
bs=[v for k,v in d['per_class_table'].items() if k.startswith('by_shape')][0]
for k,v in list(bs.items())[:16]: print(k,v)
PY
