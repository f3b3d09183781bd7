#!/bin/bash
# (needs tools/experiments/ablation_and_timing_macros.patch applied to a scratch copy of csrc/: the scaffolds left the product sources in round 6)
# Stage ablations of the F(4x4,3x3) kernel: gpurun_ablate/lib_ab*.so built by tools/split_ab.sh with -DF4_ABLATE=n (1 no MFMAs, 2 no
# transform, 4 no prologue + store, 8 no epilogue, 16 no weight fetch).  Usage (GPU box, repo root): bash tools/f43_ablate.sh <tag>
tag=${1:-1}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== full"; timeout 200 python tools/f43_check.py time 2>&1 | grep -v amdgpu.ids | sed -e 's/F(2,3).*//' | tee gpurun_out/f43_ablate_$tag.log
for n in 1 2 4 8 16 23; do
  f=gpurun_ablate/lib_ab$n.so; [ -f "$f" ] || continue
  echo "== F4_ABLATE=$n" | tee -a gpurun_out/f43_ablate_$tag.log
  CF_LIB_PATH=$f timeout 200 python tools/f43_check.py time 2>&1 | grep -v amdgpu.ids | sed -e 's/F(2,3).*//' | tee -a gpurun_out/f43_ablate_$tag.log
done
