#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in k32 ovl; do
for shape in "128 128 256" "64 64 512"; do
  echo "=== CF_F43_WIDE=$m fp32 $shape"
  CF_F43_WIDE=$m CF_LIB_PATH=$PWD/gpurun_ablate/lib_timing.so timeout 200 python tools/f43_timing.py $shape fp32 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/r6_f43_fp32_stage_timing.txt 2>&1
cat gpurun_out/r6_f43_fp32_stage_timing.txt | cut -c1-220
timeout 200 python tools/power_probe.py --help 2>&1 | head -5
