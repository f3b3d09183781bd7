#!/bin/bash
# F(4x4,3x3) kernel: accuracy + per-layer time, stage ablations (gpurun_ablate/lib_ab*.so from tools/split_ab.sh with -DF4_ABLATE=n),
# then the GPU test suite and the bench with / without the F(4,3) layers.  Usage (GPU box, repo root): bash tools/f43_round.sh <tag>
tag=${1:-1}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== f43 check"; timeout 600 python tools/f43_check.py check big > gpurun_out/f43_check_$tag.log 2>&1; echo "rc=$?"; cat gpurun_out/f43_check_$tag.log | tail -20
echo "== f43 time"; timeout 300 python tools/f43_check.py time > gpurun_out/f43_time_$tag.log 2>&1; echo "rc=$?"; cat gpurun_out/f43_time_$tag.log
for f in gpurun_ablate/lib_ab*.so; do
  [ -f "$f" ] || continue
  echo "== $f"; CF_LIB_PATH=$f timeout 200 python tools/f43_check.py time 2>&1 | sed -e 's/F(2,3).*//' | tee -a gpurun_out/f43_ablate_$tag.log
done
if [ "$2" != "nosuite" ]; then
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_f43_$tag.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/pytest_gpu_f43_$tag.log
echo "== bench F43 default"; timeout 400 python bench.py --details --no-cpu-baseline --no-exact-leg --no-config3-leg > gpurun_out/bench_f43_$tag.json 2> gpurun_out/bench_details_f43_$tag.txt; echo "rc=$?"; cat gpurun_out/bench_f43_$tag.json
echo "== bench F43=0"; CODEFORMER_HIP_F43=0 timeout 400 python bench.py --no-cpu-baseline --no-exact-leg --no-config3-leg --no-roofline > gpurun_out/bench_f43off_$tag.json 2>/dev/null; echo "rc=$?"; cat gpurun_out/bench_f43off_$tag.json
fi
