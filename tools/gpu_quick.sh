#!/bin/bash
# Quick perf probe: bench with the per-shape table (no CPU baseline) + rocprofv3 kernel-trace stats as csv.
tag=${1:-q}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python bench.py --details --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_details_$tag.txt; echo "bench rc=$?"; cat gpurun_out/bench_$tag.json
rm -rf /tmp/prof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/rocprof_run_$tag.log 2>&1; echo "rocprof rc=$?"
find /tmp/prof -type f | head; f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/rocprof_kernel_stats_$tag.csv && head -15 "$f"
