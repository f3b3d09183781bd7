#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== net tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_images.py tests/test_gpu_range.py tests/test_gpu_bf16_storage.py -x -q 2>&1 | tail -4
echo "== latency"; timeout 300 python tools/latency.py f16x2 2>&1 | grep -v amdgpu.ids | grep auto | tee gpurun_out/r6_latency_b.txt
echo "== one-face timeline"; bash tools/b1_timeline.sh 1 r6b | head -6
