#!/bin/bash
# Build timing variants of cf_split.hip (compile-time SP_* knobs) and A/B them in ONE process, interleaved.
# usage: bash tools/split_ab.sh build "name:-Dflag ..." ... ; (GPU box) python tools/split_ab.py
cd "$(dirname "$0")/.."
shift; rm -rf gpurun_ablate; mkdir -p gpurun_ablate
# AB_EXTRA_SRC: extra sources, e.g. tools/experiments/cf_w64.hip together with -DCF_W64_EXPERIMENT=1
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -Iinclude -Icodeformer_amd/csrc \
    -o gpurun_ablate/lib_$name.so codeformer_amd/csrc/cf_igemm.hip codeformer_amd/csrc/cf_winograd.hip codeformer_amd/csrc/cf_split.hip codeformer_amd/csrc/cf_wsplit.hip codeformer_amd/csrc/cf_wf43.hip codeformer_amd/csrc/cf_gemm_split.hip \
    codeformer_amd/csrc/cf_norm.hip codeformer_amd/csrc/cf_attention.hip codeformer_amd/csrc/cf_misc.hip codeformer_amd/csrc/cf_paste.hip $AB_EXTRA_SRC &
done; wait; ls gpurun_ablate
