"""The end of a training step on the device, from a rocprofv3 kernel trace of bench.py (plain or --force-reducer): every kernel of
the last 1.5 ms before sgd_kernel of the second-to-last step with its queue, start, duration and the idle gap in front of it --
where the reducer's host path shows up as device idle time.
usage: python tools/tail_gap.py <kernel_trace.csv> [window_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows), key=lambda e: e[0])
steps, cur = [], []
for e in ev:
    cur.append(e)
    if e[2].startswith('sgd_kernel'):
        steps.append(cur)
        cur = []
gaps = []
for st in steps[3:-1]:
    sgd = st[-1]
    last_bwd = max((e for e in st if e[2].startswith(('conv_wgrad', 'conv_tap', 'fold_unpack', 'narrow_head', 'roi_align'))), key=lambda e: e[1])
    busy_end = max(e[1] for e in st[:-1] if not e[2].startswith(('sumsq', 'sgd')))
    gaps.append(((sgd[0] - last_bwd[1]) / 1e3, (sgd[0] - busy_end) / 1e3, (sgd[1] - st[0][0]) / 1e6))
print(f'# {len(gaps)} steps: gap last backward kernel -> sgd_kernel start (us): ' + ' '.join(f'{g[0]:.0f}' for g in gaps))
print(f'# mean {sum(g[0] for g in gaps) / len(gaps):.0f} us; step wall (first kernel -> sgd end) mean {sum(g[2] for g in gaps) / len(gaps):.3f} ms')
st = steps[-2]
sgd = st[-1]
t_end = sgd[1]
prev_end = None
for s, e, n, q in st:
    if s < sgd[0] - win * 1e3:
        prev_end = max(prev_end or e, e)
        continue
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f'  t-{(sgd[0] - s) / 1e3:8.1f} us  q{q:>2s}  dur {(e - s) / 1e3:7.1f} us  idle before {max(gap, 0.0):6.1f} us  {n.split("(")[0][:60]}')
    prev_end = max(prev_end or e, e)
