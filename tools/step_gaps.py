"""Device idle time inside one training step from a rocprofv3 kernel trace (all queues merged): the largest gaps with the kernels
around them, and the idle total -- run on a plain and on a --force-reducer trace to see where the reducer's host path leaves the
device waiting.   usage: python tools/step_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
mn = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-60:], r.get('Queue_Id', '?')) for r in rows))
steps, cur = [], []
for e in ev:
    cur.append(e)
    if e[2].endswith('sgd_kernel'):
        steps.append(cur)
        cur = []
tot = []
for st in steps[2:]:
    t0, idle, end = st[0][0], 0, st[0][1]
    for s, e, n, q in st[1:]:
        if s > end:
            idle += s - end
        end = max(end, e)
    tot.append(((end - t0) / 1e6, idle / 1e6))
print('# per step (wall ms, idle ms): ' + ' '.join(f'({w:.2f}, {i:.2f})' for w, i in tot))
print(f'# mean wall {sum(w for w, _ in tot) / len(tot):.3f} ms, mean idle {sum(i for _, i in tot) / len(tot):.3f} ms')
st = steps[-2]
t0, end, prev = st[0][0], st[0][1], st[0]
gaps = []
for s, e, n, q in st[1:]:
    if s > end:
        gaps.append((s - end, (end - t0) / 1e3, prev[2], n))
    if e > end:
        end, prev = e, (s, e, n, q)
gaps.sort(reverse=True)
print(f'# step {len(steps) - 2}: {len(gaps)} gaps, {sum(g[0] for g in gaps) / 1e3:.0f} us idle; gaps >= {mn} us:')
for g, at, a, b in sorted([x for x in gaps if x[0] >= mn * 1e3], key=lambda x: x[1]):
    print(f'  at {at:9.1f} us  idle {g / 1e3:7.1f} us   after {a}  ->  before {b}')
