#!/bin/bash
# Standard on-box sequence: parity tests, smoke, bench (+per-class table), rocprofv3 kernel-trace summary of the bench.
# Usage (GPU box, repo root): bash tools/gpu_round.sh <tag>
tag=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_gpu_$tag.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench"; timeout 600 python bench.py --details > gpurun_out/bench_$tag.json 2> gpurun_out/bench_details_$tag.txt; echo "rc=$?"; cat gpurun_out/bench_$tag.json
echo "== rocprofv3 kernel trace of bench.py (no cpu baseline)"
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-exact-leg --no-config3-leg --no-parity-gate > gpurun_out/rocprof_run_$tag.log 2>&1; echo "rc=$?"
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/rocprof_kernel_stats_$tag.csv && head -12 "$f"
ls /tmp/prof -R | head -20
