#!/bin/bash
# HBM traffic per launch of the conv kernels on the micro-benchmark shapes (run on the GPU box): FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes (never combined with the trace domains), grouped by (kernel, grid).  usage: tools/pmc_traffic_shapes.sh <fwd|dgrad|wgrad>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/pmc_shapes
export TMPDIR=/tmp; cd /tmp
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python $R/tools/bench_conv.py $1 > $OUT/$c.txt 2>&1 || true
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv' not in r['Kernel_Name']:
                continue
            k = (r['Kernel_Name'][:40], r.get('Grid_Size'))
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
for k, v in agg.items():
    f = v.get('FETCH_SIZE', 0) * 1024 * 2 / max(1, len(cnt[k]['FETCH_SIZE']))      # gfx950: FETCH_SIZE counts 2 KiB... (guide: x2 correction)
    w = v.get('WRITE_SIZE', 0) * 1024 / max(1, len(cnt[k]['WRITE_SIZE']))
    print(f'{k[0]:40s} grid={k[1]:>9s} n={len(cnt[k]["FETCH_SIZE"]):3d}  fetch {f / 1e6:8.1f} MB  write {w / 1e6:8.1f} MB')
PY
