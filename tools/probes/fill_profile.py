"""How much of a step runs with the GPU under-filled?  From a rocprofv3 kernel trace: at every instant sum the workgroups of
the kernels in flight (grid / workgroup size, each kernel capped at the 256 CUs); time with fewer than 128 workgroups in
flight is 'thin'.  Prints the thin time per step and the kernels that own it.   usage: python fill_profile.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def nwg(r):
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    w = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    return max(1, g // max(1, w))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], nwg(r)) for r in rows), key=lambda e: e[0])
steps, cur = [], []
for e in ev:
    cur.append(e)
    if e[2].startswith('sgd_kernel'):
        steps.append(cur); cur = []
st = steps[-3]
t0, t1 = st[0][0], max(e[1] for e in st)
pts = []
for i, (s, e, n, w) in enumerate(st):
    pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
live = set()
thin = collections.Counter(); idle = 0; thin_total = 0
last = t0
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        fill = sum(min(256, st[j][3]) for j in live)
        if not live:
            idle += dt
        elif fill < 128:
            thin_total += dt
            for j in live:
                thin[st[j][2][:70]] += dt / len(live)
    last = t
    if d == 1: live.add(i)
    else: live.discard(i)
print(f'step wall {(t1 - t0) / 1e6:.3f} ms   idle {idle / 1e6:.3f} ms   thin (<128 workgroups in flight) {thin_total / 1e6:.3f} ms   launches {len(st)}')
for n, v in thin.most_common(28):
    print(f'   {v / 1e3:8.1f} us  {n}')

# ---- contiguous under-filled windows (idle or thin, gaps of filled time < 5 us bridged): where the serial sections are
segs, cur_s, cur_e, names = [], None, None, []
last = t0
live = set()
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        fill = sum(min(256, st[j][3]) for j in live)
        if fill < 128:
            if cur_s is None or last - cur_e > 5000:
                if cur_s is not None:
                    segs.append((cur_s, cur_e, names))
                cur_s, names = last, []
            cur_e = t
            for j in live:
                if st[j][2][:40] not in names:
                    names.append(st[j][2][:40])
    last = t
    if d == 1: live.add(i)
    else: live.discard(i)
if cur_s is not None:
    segs.append((cur_s, cur_e, names))
print(f'under-filled windows: {len(segs)}, total {sum(e - s for s, e, _ in segs) / 1e6:.3f} ms')
for s_, e_, nm in sorted(segs, key=lambda x: -(x[1] - x[0]))[:14]:
    print(f'   at {(s_ - t0) / 1e6:6.2f} ms  {(e_ - s_) / 1e3:7.1f} us  {len(nm):2d} kernels: ' + ' | '.join(nm[:7]))

# ---- launch by launch inside the largest windows
for s_, e_, nm in sorted(segs, key=lambda x: -(x[1] - x[0]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 0]:
    print(f'--- window at {(s_ - t0) / 1e6:.2f} ms, {(e_ - s_) / 1e3:.0f} us')
    prev_end = s_
    for a, b, n, w in st:
        if b >= s_ and a <= e_:
            print(f'    +{(a - s_) / 1e3:7.1f} us  gap {(a - prev_end) / 1e3:6.1f}  dur {(b - a) / 1e3:6.1f} us  wg {w:6d}  {n[:80]}')
            prev_end = max(prev_end, b)
