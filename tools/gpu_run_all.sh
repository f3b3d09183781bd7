#!/bin/bash
# Run every gpu_check group in its own process (a faulting kernel cannot take the others down).
# Usage (on the GPU box, from the repo root): bash tools/gpu_run_all.sh [group ...]
mkdir -p gpurun_out
groups=${@:-basic conv attn blocks net}
rc=0
for g in $groups; do
  timeout 600 python tools/gpu_check.py $g > gpurun_out/check_$g.log 2>&1
  s=$?
  echo "== $g exit $s"; grep -E "FAIL|SUMMARY|faces/s|indices|Error|error" gpurun_out/check_$g.log | head -40
  [ $s -ne 0 ] && rc=1
done
exit $rc
