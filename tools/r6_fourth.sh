#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in k32 ovl; do CF_F43_WIDE=$m timeout 300 python tools/power_probe.py fp32 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6_power_probe_fp32.txt
