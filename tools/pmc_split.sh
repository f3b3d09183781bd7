#!/bin/bash
# PMC passes over the split-half conv kernel on one layer shape (each counter set in its own run, kernel-trace only).
# usage (GPU box): bash tools/pmc_split.sh <tag> cin cout H swish up
tag=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/ps$i
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/ps$i -o p -- python tools/split_one.py "$@" > gpurun_out/pmcsplit_${tag}_run$i.log 2>&1
  echo "set $i rc=$?"
done
python - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob('/tmp/ps*/**/*counter_collection.csv', recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'split_conv' in name or 'winograd' in name or 'igemm' in name or 'wsplit' in name or 'wf43_kernel' in name:
            key = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:64]
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            if (f, r['Dispatch_Id']) not in seen:
                seen.add((f, r['Dispatch_Id']))
                dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {}
for k, d in agg.items():
    out[k] = {c: sum(v) / len(v) for c, v in d.items()}
    out[k]['mean_duration_ns'] = sum(dur[k]) / len(dur[k])
json.dump(out, open(f'gpurun_out/pmc_split_{tag}.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
