#!/bin/bash
# Round-end sequence on the GPU box: parity tests, smoke, the default bench line (+ per-class table), rocprofv3 kernel-trace summary, PMC
# passes of the same step (stamped with the library's build id), one-face latency sweep.  Usage (repo root): bash tools/final_round.sh <tag>
tag=${1:-r04}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_gpu_$tag.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py --details > gpurun_out/bench_$tag.json 2> gpurun_out/bench_details_$tag.txt; echo "rc=$?"; cut -c1-600 gpurun_out/bench_$tag.json
echo "== rocprofv3 kernel trace"; rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-exact-leg --no-config3-leg --no-parity-gate > gpurun_out/rocprof_run_$tag.log 2>&1; echo "rc=$?"
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/rocprof_kernel_stats_$tag.csv && head -8 "$f" | cut -c1-160
echo "== pmc"; bash tools/pmc_bench.sh $tag > gpurun_out/pmc_bench_$tag.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pmc_bench_$tag.log | cut -c1-300
echo "== latency"; timeout 300 python tools/latency.py f16x2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/latency_$tag.txt
