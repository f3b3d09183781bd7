#!/bin/bash
# A/B of the three forms of the 16-wave F(4,3) workgroup in one box call: accuracy check per form, digests (k16 == ovl), launch times.
tag=${1:-ovl}
mkdir -p gpurun_out
for m in k32 k16 ovl; do
  echo "=== CF_F43_WIDE=$m"
  CF_F43_WIDE=$m timeout 300 python tools/f43_check.py check 2>&1 | grep -v amdgpu.ids | grep -E "FAIL|16->|cat|128|256" | head -20
  CF_F43_WIDE=$m timeout 300 python tools/f43_ovl_ab.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/f43_ovl_ab_$tag.txt 2>&1
python - <<PY
import re
txt = open('gpurun_out/f43_ovl_ab_$tag.txt').read()
sec = {m: s for m, s in zip(('k32', 'k16', 'ovl'), re.split(r'=== CF_F43_WIDE=\w+\n', txt)[1:])}
dig = {m: re.findall(r'digest (.*)', s) for m, s in sec.items()}
print('k16 == ovl digests:', dig['k16'] == dig['ovl'] and len(dig['ovl']) > 0, '| FAIL lines:', txt.count('FAIL'))
for m, s in sec.items():
    for l in re.findall(r'time .*', s):
        print(l)
PY
