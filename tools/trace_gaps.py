"""Kernel-trace helper: per kernel name, launches, average duration and the average idle gap BEFORE and AFTER it on the stream."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print('columns:', [c for c in rows[0].keys()])
agg = collections.defaultdict(lambda: [0, 0, 0, 0, set()])
for i in range(1, len(rows) - 1):
    r = rows[i]
    k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:58]
    a = agg[k]
    a[0] += 1
    a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a[2] += max(0, int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']))
    a[3] += max(0, int(rows[i + 1]['Start_Timestamp']) - int(r['End_Timestamp']))
    a[4].add(r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '?')))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if a[0] >= 20:
        print(f'{a[0]:6d} x  dur {a[1] / a[0] / 1e3:8.1f} us  gap before {a[2] / a[0] / 1e3:6.1f}  after {a[3] / a[0] / 1e3:6.1f}  lds {sorted(a[4])[:3]}  {k}')
