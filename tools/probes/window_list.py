"""Launch-by-launch listing of one step's window between the first launch of kernel A and the first launch of kernel B after it
(rocprofv3 kernel trace).  usage: python window_list.py <kernel_trace.csv> <substring A> <substring B>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
A, B = sys.argv[2], sys.argv[3]
def nwg(r):
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    w = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    return max(1, g // max(1, w))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], nwg(r), r.get('Queue_Id', '?')) for r in rows),
            key=lambda e: e[0])
steps, cur = [], []
for e in ev:
    cur.append(e)
    if e[2].startswith('sgd_kernel'):
        steps.append(cur); cur = []
st = steps[-3]
i0 = next(i for i, e in enumerate(st) if A in e[2])
i1 = next(i for i, e in enumerate(st) if i > i0 and B in e[2])
t0 = st[i0][0]
end = t0
print(f'window {A} -> {B}: {(st[i1][0] - t0) / 1e3:.1f} us, {i1 - i0} launches')
for s, e, n, w, q in st[i0:i1 + 1]:
    print(f'  +{(s - t0) / 1e3:8.1f} us  gap {(s - end) / 1e3:6.1f}  dur {(e - s) / 1e3:6.1f}  wg {w:6d}  q {q:>3s}  {n[:90]}')
    end = max(end, e)
if len(sys.argv) > 4:      # extra: the N launches that follow the window
    for s, e, n, w, q in st[i1 + 1:i1 + 1 + int(sys.argv[4])]:
        print(f'  +{(s - t0) / 1e3:8.1f} us  gap {(s - end) / 1e3:6.1f}  dur {(e - s) / 1e3:6.1f}  wg {w:6d}  q {q:>3s}  {n[:90]}')
        end = max(end, e)
if len(sys.argv) > 5:      # extra: the last N launches of the step
    print('--- step tail')
    tail = st[-int(sys.argv[5]):]
    end = tail[0][0]
    for s, e, n, w, q in tail:
        print(f'  +{(s - t0) / 1e3:8.1f} us  gap {(s - end) / 1e3:6.1f}  dur {(e - s) / 1e3:6.1f}  wg {w:6d}  q {q:>3s}  {n[:90]}')
        end = max(end, e)
