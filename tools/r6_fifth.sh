#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_split.py -x -q -k "two_token or fused_finalize or process_level or token_gemm" 2>&1 | grep -v amdgpu.ids | tail -15
echo "== net tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_images.py tests/test_gpu_range.py -x -q 2>&1 | tail -4
echo "== latency"; timeout 300 python tools/latency.py f16x2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_latency_a.txt
echo "== latency, fusions off"; CODEFORMER_HIP_FINALIZE_FUSED=0 CODEFORMER_HIP_QKV_ONE_LAUNCH=0 timeout 300 python tools/latency.py f16x2 2>&1 | grep -v amdgpu.ids | grep auto | tee -a gpurun_out/r6_latency_a.txt
echo "== one-face timeline"; bash tools/b1_timeline.sh 1 r6a | head -14
