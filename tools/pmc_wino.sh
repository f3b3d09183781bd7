#!/bin/bash
# Extra PMC passes for the Winograd kernel: VALU / LDS / MFMA instruction activity, LDS bank conflicts and the MFMA-VALU
# co-execution counter (each counter set in its own run, kernel-trace only).
mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_]*\(VALU\|LDS\|MFMA\)[A-Z_]*" | sort -u | tr '\n' ' ' > gpurun_out/pmc_avail_sq.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pw$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pw$i -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-exact-leg --no-config3-leg > gpurun_out/pmcwino_run$i.log 2>&1
  echo "set $i rc=$?"
done
python - <<'PY'
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('/tmp/pw*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'winograd_kernel' in n or 'igemm_kernel<4' in n:
            agg[n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open('gpurun_out/pmc_wino.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
PY
