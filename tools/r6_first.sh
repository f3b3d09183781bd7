#!/bin/bash
# round 6, first box call: fp32 token-tile GEMM probe + tests, the new bench line, A/B of the overlapped 16-wave F(4,3) form with fp32 operands
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== gemm probe"; timeout 300 python tools/gemm_f32_tile_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_gemm_f32_tile_probe.txt
echo "== gemm probe, old kernel"; CF_GEMM_F32_TILE=0 timeout 300 python tools/gemm_f32_tile_probe.py 2>&1 | grep -v amdgpu.ids | grep "B=16" | tee -a gpurun_out/r6_gemm_f32_tile_probe.txt
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_split.py -q -x -k "fp32_token or splitk_gemm" 2>&1 | tail -3
echo "== bench (default)"; timeout 900 python bench.py --details > gpurun_out/r6_bench_a.json 2> gpurun_out/r6_bench_a_details.txt; echo "rc=$?"; wc -c gpurun_out/r6_bench_a.json; cat gpurun_out/r6_bench_a.json
S="--no-cpu-baseline --no-f16x2-leg --no-config3-leg --no-roofline --steps 10 --warmup 3"
for m in k32 ovl k16; do echo "== fp32 CF_F43_WIDE=$m"; CF_F43_WIDE=$m timeout 300 python bench.py $S 2>/dev/null | cut -c1-200; done
echo "== fp32 ovl + prio 2"; CF_LIB_PATH=$PWD/gpurun_ablate/lib_ovlprio.so CF_F43_WIDE=ovl timeout 300 python bench.py $S 2>/dev/null | cut -c1-200
echo "== fp32 k32 again"; timeout 300 python bench.py $S 2>/dev/null | cut -c1-200
echo "== fp32, old token GEMM"; CF_GEMM_F32_TILE=0 timeout 300 python bench.py $S 2>/dev/null | cut -c1-200
