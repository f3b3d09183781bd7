#!/bin/bash
# Kernel-trace summary of the batch-N call (default N=1): where a small-batch forward spends its time.
n=${1:-1}; tag=${2:-b$n}
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof_b && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python bench.py --batch-per-gpu $n --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-exact-leg --no-config3-leg > gpurun_out/rocprof_run_$tag.log 2>&1; echo "rc=$?"
tail -1 gpurun_out/rocprof_run_$tag.log | cut -c1-200
f=$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/rocprof_kernel_stats_$tag.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/rocprof_kernel_stats_$tag.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms per forward', tot/23/1e6)
for r in rows[:22]:
    print(f"{int(r['TotalDurationNs'])/23/1e3:9.1f} us/fwd  {int(r['Calls'])//23:4d} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
