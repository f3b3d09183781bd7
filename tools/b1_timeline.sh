#!/bin/bash
# One-face call, graph replay: kernel timeline (rocprofv3 --kernel-trace) of N replays -> busy time, gaps and per-class sums per forward.
# usage (GPU box, repo root): bash tools/b1_timeline.sh [batch] [tag]
b=${1:-1}; tag=${2:-b$b}
mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/b1_run.py <<PY
import sys, torch
sys.path.insert(0, '.')
import codeformer_amd.archs
from codeformer_amd.utils.registry import ARCH_REGISTRY
from oracle.synth import seeded_input
torch.manual_seed(0)
net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9, connect_list=['32', '64', '128', '256']).eval().cuda()
x = seeded_input($b).cuda()
for _ in range(5):
    net(x, w=0.5, adain=True)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(20):
    net(x, w=0.5, adain=True)
torch.cuda.synchronize()
print('ms per call', (time.perf_counter() - t) / 20 * 1e3)
PY
rm -rf /tmp/b1prof && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1prof -o t -- python /tmp/b1_run.py 2>&1 | grep "ms per call"
f=$(find /tmp/b1prof -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee gpurun_out/b1_timeline_$tag.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 20 forwards = the last 20 repetitions of the per-forward launch pattern: find the period from the tail
names = [r['Kernel_Name'] for r in rows]
last = names[-1]
idx = [i for i, n in enumerate(names) if n == last]
# period = distance between the last two occurrences that repeats
per = None
for k in range(len(idx) - 2, -1, -1):
    p = idx[-1] - idx[k]
    if p > 50 and names[-p:] == names[-2 * p:-p]:
        per = p
        break
print('launches per forward:', per)
R = 10
tail = rows[-R * per:]
span = (int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])) / R / 1e3
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tail) / R / 1e3
gaps = [max(0, int(tail[i]['Start_Timestamp']) - int(tail[i - 1]['End_Timestamp'])) for i in range(1, len(tail))]
print(f'per forward: span {span:.1f} us, kernel busy {busy:.1f} us, gaps {sum(gaps) / R / 1e3:.1f} us (avg {sum(gaps) / len(gaps) / 1e3:.2f} us per boundary)')
agg = collections.defaultdict(lambda: [0, 0, 0])
for i, r in enumerate(tail):
    k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:70]
    a = agg[k]
    a[0] += 1
    a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if i:
        a[2] += max(0, int(r['Start_Timestamp']) - int(tail[i - 1]['End_Timestamp']))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{a[1] / R / 1e3:9.1f} us/fwd  {a[0] / R:6.1f} launches  avg {a[1] / a[0] / 1e3:7.1f} us  gap before {a[2] / a[0] / 1e3:5.2f} us  {k}')
PY
