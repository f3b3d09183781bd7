#!/bin/bash
# Build kernel-tuning variants (compile-time knobs of cf_igemm.hip) and A/B them in ONE process, interleaved.
# usage: bash tools/ab_variants.sh build "name:-Dflag ..." ... ; (GPU box) bash tools/ab_variants.sh run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift; rm -rf gpurun_ablate; mkdir -p gpurun_ablate
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -Iinclude -Icodeformer_amd/csrc \
      -o gpurun_ablate/lib_$name.so codeformer_amd/csrc/cf_igemm.hip codeformer_amd/csrc/cf_winograd.hip codeformer_amd/csrc/cf_norm.hip \
      codeformer_amd/csrc/cf_attention.hip codeformer_amd/csrc/cf_misc.hip &
  done; wait; ls gpurun_ablate
else
  python tools/ab_variants.py
fi
