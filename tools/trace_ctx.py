"""Kernel-trace helper: average duration of the attention launches of a bench run, of their successor, and the gaps around them."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
att, nxt, gap_before, gap_after, prv = [], [], [], [], []
for i, r in enumerate(rows[:-1]):
    if 'attn64' in r['Kernel_Name'] or 'attn_kernel<64>' in r['Kernel_Name']:
        att.append(dur(r)); nxt.append(dur(rows[i + 1])); prv.append(dur(rows[i - 1]))
        gap_before.append(int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']))
        gap_after.append(int(rows[i + 1]['Start_Timestamp']) - int(r['End_Timestamp']))
m = lambda v: sum(v) / max(len(v), 1) / 1e3
print(f'{len(att)} attention launches: avg {m(att):.1f} us, previous kernel {m(prv):.1f} us, next kernel {m(nxt):.1f} us, gap before {m(gap_before):.1f} us, gap after {m(gap_after):.1f} us')
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r['Kernel_Name'][:60]; tot[k][0] += 1; tot[k][1] += dur(r)
print('busy total ms', sum(v[1] for v in tot.values()) / 1e6, 'span ms', (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6)
