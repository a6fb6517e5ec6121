#!/bin/bash
# SQ stall breakdown of the conv kernels on the micro-benchmark shapes (run on the GPU box from /tmp with TMPDIR=/tmp).
# usage: tools/pmc_conv.sh <fwd|dgrad|wgrad> <out_dir>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${2:-$R/gpurun_out/pmc_conv}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/a -- python $R/tools/bench_conv.py $1 > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/b -- python $R/tools/bench_conv.py $1 > /dev/null 2>&1 || true
python - <<PY
import csv, glob, collections
for tag in ('a','b'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for f in glob.glob('$OUT/%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k=(r['Kernel_Name'][:44], r.get('Grid_Size'), r.get('Workgroup_Size'))
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES', kv[1].get('SQ_BUSY_CU_CYCLES',0)))[:14]:
        n=len(cnt[k])
        print(tag, k, 'n=%d'%n, {c: round(x/n/1e6,2) for c,x in v.items()})
PY
