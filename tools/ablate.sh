#!/bin/bash
# Build timing-only ablation variants of the conv kernel (never shipped) and time one shape with each.
# usage (repo root, here): bash tools/ablate.sh build ; (GPU box): bash tools/ablate.sh run [cin cout H]
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p gpurun_ablate
  for k in 0 1 3 4 5; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCF_ABLATE=$k -Iinclude -Icodeformer_amd/csrc \
      -o gpurun_ablate/lib_ablate$k.so codeformer_amd/csrc/cf_igemm.hip codeformer_amd/csrc/cf_norm.hip \
      codeformer_amd/csrc/cf_attention.hip codeformer_amd/csrc/cf_misc.hip &
  done; wait; ls -la gpurun_ablate
else
  shift
  for k in 0 1 3 4 5; do
    echo "== ablate $k (0 base, 1 no epilogue, 3 no per-step barrier, 4 no weight fetch/store, 5 no LDS fragment reads)"
    CF_LIB_PATH=$PWD/gpurun_ablate/lib_ablate$k.so python tools/conv_bench.py 16 brief
  done
fi
