export TMPDIR=/tmp; mkdir -p gpurun_out
for cfg in "1 128 3 128" "1 64 3 128" "1 128 2 128" "0 128 3 128" "1 128 4 128" "1 64 5 64"; do set -- $cfg; echo "== swish=$1 cin=$2 B=$3 H=$4"; DBG_SWISH=$1 DBG_CIN=$2 DBG_B=$3 DBG_H=$4 timeout 60 python tools/w64_debug.py 2>&1 | grep -v amdgpu.ids | head -2; done
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pw$i
  CONV_KIND=wsplit timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pw$i -o p -- python tools/split_one.py 64 64 512 1 0 3 > gpurun_out/pmc_w64_run$i.log 2>&1
  echo "set $i rc=$?"
done
python - <<'PY'
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('/tmp/pw*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'w64_kernel' in n or 'winograd_kernel' in n:
            agg[n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open('gpurun_out/pmc_w64.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
PY
