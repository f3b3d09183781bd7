#!/bin/bash
# PMC passes (each its own run; --kernel-trace only, as the pool requires) on the dominant conv shape.
tag=${1:-pmc}; shift
args=${@:-128 128 256 16 4}
mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc$i -o p -- python tools/conv_single.py $args > gpurun_out/pmc_${tag}_run$i.log 2>&1
  echo "set $i rc=$?"
  f=$(find /tmp/pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'igemm' in r['Kernel_Name']:
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f'   {c:36s} n={len(v)} mean={sum(v)/len(v):.4g} last={v[-1]:.4g}')
PY
  [ -n "$f" ] && cp "$f" gpurun_out/pmc_${tag}_set$i.csv
done
