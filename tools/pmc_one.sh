#!/bin/bash
# PMC passes with explicit counter sets over tools/split_one.py (kernel-trace only): bash tools/pmc_one.sh <tag> "<set1>" "<set2>" ... -- <split_one args>
tag=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); rm -rf /tmp/po$i
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/po$i -o p -- python tools/split_one.py "$@" > gpurun_out/pmcone_${tag}_run$i.log 2>&1
  echo "set $i rc=$?"
done
python - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('/tmp/po*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'split_conv' in name or 'winograd' in name or 'igemm' in name:
            key = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:64]
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(f'gpurun_out/pmc_one_{tag}.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
