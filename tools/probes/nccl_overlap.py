"""From a rocprofv3 kernel trace of `bench.py --force-reducer` (one-rank RCCL group, reducer active): every RCCL kernel of the last
full step with its queue, start / duration relative to the step, and the compute kernels of OTHER queues that ran during it.
usage: python tools/probes/nccl_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows),
            key=lambda e: e[0])
steps, cur = [], []
for e in ev:
    cur.append(e)
    if e[2].startswith('sgd_kernel'):
        steps.append(cur); cur = []
st = steps[-2]
t0, t1 = st[0][0], max(e[1] for e in st)
nccl = [e for e in st if 'nccl' in e[2].lower() or 'rccl' in e[2].lower()]
print(f'step wall {(t1 - t0) / 1e6:.3f} ms, {len(st)} launches, {len(nccl)} RCCL kernels; queues in the step: '
      f'{sorted(collections.Counter(e[3] for e in st).items())}')
sgd = [e for e in st if e[2].startswith('sgd_kernel')][0]
last_bwd = max((e for e in st if e[2].startswith(('conv_wgrad', 'conv_tap', 'fold_unpack'))), key=lambda e: e[1])
print(f'last conv / unpack kernel ends at {(last_bwd[1] - t0) / 1e3:.1f} us, sgd_kernel starts at {(sgd[0] - t0) / 1e3:.1f} us')
tot_n = tot_o = 0
for s, e, n, q, sid in nccl:
    over = collections.Counter()
    for s2, e2, n2, q2, _ in st:
        if q2 == q or 'nccl' in n2.lower():
            continue
        o = min(e, e2) - max(s, s2)
        if o > 0:
            over[n2.split('(')[0][:48]] += o
    cov = sum(over.values())
    tot_n += e - s; tot_o += min(cov, e - s)
    top = ', '.join(f'{k} {v / 1e3:.0f}us' for k, v in over.most_common(3))
    print(f'  queue {q} stream {sid}  start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  {n.split("(")[0][:40]:40s} | beside: {top or "-"}')
print(f'RCCL kernel time {tot_n / 1e3:.1f} us per step, of which {tot_o / 1e3:.1f} us with compute kernels of other queues in flight')
