#!/bin/bash
# eager vs hipGraph-replayed backbone + neck (run ON the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3g
mkdir -p "$OUT"
cd "$ROOT"
for g in "" graph; do
  python tools/probes/host_tail.py 1024 loft_foa_r50_fpn_2x_bonai.py $g 2>&1 | grep "^size" | tail -3
done | tee "$OUT/host_tail_r50.txt"
ms() { python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('graph_features'))"; }
for i in 1 2; do
  echo "sat eager  $(python bench.py --no-cpu-baseline --no-roofline --no-light 2>/dev/null | ms)"
  echo "sat graph  $(python bench.py --no-cpu-baseline --no-roofline --no-light --graph 2>/dev/null | ms)"
  echo "light eager $(python bench.py --no-cpu-baseline --no-roofline --no-saturate 2>/dev/null | ms)"
  echo "light graph $(python bench.py --no-cpu-baseline --no-roofline --no-saturate --graph 2>/dev/null | ms)"
done | tee "$OUT/bench_ab.txt"
C5=loft_foa_hrnetv2p_w32_2x_bonai.py
for g in "" graph; do
  python tools/probes/host_tail.py 1024 $C5 $g 2>&1 | grep "^size" | tail -3
done | tee "$OUT/host_tail_hrnet.txt"
echo "cfg5 eager $(python bench.py --config $C5 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee "$OUT/bench_c5.txt"
echo "cfg5 graph $(python bench.py --config $C5 --no-cpu-baseline --no-roofline --graph 2>&1 | tail -1 | ms)" | tee -a "$OUT/bench_c5.txt"
C4=loft_foa_r50_fpn_mdconv_c3-c5_2x_bonai.py
echo "cfg4 eager $(python bench.py --config $C4 --no-cpu-baseline --no-roofline 2>/dev/null | ms)" | tee "$OUT/bench_c4.txt"
echo "cfg4 graph $(python bench.py --config $C4 --no-cpu-baseline --no-roofline --graph 2>&1 | tail -1 | ms)" | tee -a "$OUT/bench_c4.txt"
