#!/bin/bash
# PMC passes over one bench step (each counter set in its own run, --kernel-trace only): per-kernel averages -> JSON.
# Usage: bash tools/pmc_bench.sh <tag> [precision]   precision fp32 (default, bench.py's headline) / f16x2 / bf16
#        -> gpurun_out/pmc_bench_<tag>_<precision>.json (copy to profiles/<tag>_pmc_bench_<precision>.json: bench.py quotes
#        `traffic_recorded` of the leg with that precision from it, when the build id matches)
tag=${1:-r01}
prec=${2:-fp32}
suffix="_$prec"
wflag=""; [ "$prec" = "bf16" ] && wflag="--w 0.7"
mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pb$i; [ $i = 1 ] && rm -rf /tmp/pb[0-9]*
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pb$i -o p -- python bench.py --steps 1 --warmup 1 --precision $prec $wflag --no-cpu-baseline --no-roofline --no-exact-leg --no-config3-leg --no-parity-gate > gpurun_out/pmcbench_${tag}${suffix}_run$i.log 2>&1
  echo "set $i rc=$?"
done
python - "$tag$suffix" "$prec $wflag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob('/tmp/pb*/**/*counter_collection.csv', recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'igemm' in name or 'gemm_split' in name or 'act_scale' in name or 'split_conv' in name or 'winograd' in name or 'wsplit' in name or 'wf43_kernel' in name or 'few_c' in name or 'gn_' in name or 'attn' in name or 'layernorm' in name or 'argmax' in name or 'gather' in name:
            key = name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:64]
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            if (f, r['Dispatch_Id']) not in seen and r['Counter_Name'] in ('GRBM_GUI_ACTIVE', 'FETCH_SIZE'):
                seen.add((f, r['Dispatch_Id']))
                dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {}
for k, d in agg.items():
    out[k] = {c: {'launches': len(v), 'mean': sum(v) / len(v), 'sum': sum(v)} for c, v in d.items()}
    if dur[k]:
        out[k]['mean_duration_ns'] = sum(dur[k]) / len(dur[k])
# stamp the library the counters were taken from: bench.py only quotes `traffic` from a file whose build id equals the loaded library's
import ctypes, os
lib = ctypes.CDLL(os.path.join('codeformer_amd', 'libcodeformer_hip.so'))
lib.cf_build_id.restype = ctypes.c_char_p
out['_meta'] = {'cf_build_id': lib.cf_build_id().decode(), 'command': f'bench.py --steps 1 --warmup 1 --precision {sys.argv[2].strip()} --no-cpu-baseline --no-roofline --no-exact-leg --no-config3-leg --no-parity-gate'}
json.dump(out, open(f'gpurun_out/pmc_bench_{tag}.json', 'w'), indent=1, sort_keys=True)
for k, d in out.items():
    if k == '_meta':
        continue
    print(k, {c: (round(v['mean'], 1) if isinstance(v, dict) else round(v)) for c, v in d.items()})
PY
