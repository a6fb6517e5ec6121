"""GPU idle time inside a training step: union of the kernel intervals of a rocprofv3 kernel trace vs the wall time between the
first and last kernel of each step (steps are delimited by the sgd_kernel launch).  Prints the largest gaps with the kernels
on either side.   usage: python gap_profile.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
steps, cur = [], []
for e in ev:
    cur.append(e)
    if e[2].startswith('sgd_kernel'):
        steps.append(cur); cur = []
for si, st in enumerate(steps[-6:-1]):
    t0, t1 = st[0][0], max(e[1] for e in st)
    busy, end, gaps = 0, t0, []
    prev = None
    for s, e, n in st:
        if s > end:
            gaps.append((s - end, prev, n)); busy += e - s; end = e
        elif e > end:
            busy += e - end; end = e
        if e >= end: prev = n
    print(f'step {si}: wall {(t1 - t0) / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(t1 - t0 - busy) / 1e6:.3f} ms  launches {len(st)}')
    if si == 2:
        for g, a, b in sorted(gaps, key=lambda x: -x[0])[:14]:
            print(f'    gap {g / 1e3:7.1f} us   after {str(a)[:48]:48s} before {b[:48]}')
