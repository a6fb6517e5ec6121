#!/bin/bash
# The committed bench line + per-shape table, measured on a box that has run nothing else (see tools/refresh_profiles.sh):
#   gpurun -- 'bash tools/refresh_bench_line.sh'   ->  gpurun_out/refresh/bench_line.json, bench_shapes.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
cd "$ROOT"
LOFT_DUMP_SHAPES=1 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
grep '^#' "$OUT/bench.err" > "$OUT/bench_shapes.txt"
python - <<'PY'
import json
d = json.load(open('gpurun_out/refresh/bench_line.json'))
f = d['value_fp32_parity']
print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'stale', d['roofline']['stale_profiles_not_quoted'], 'fp32', f['value'],
      'forced', d['comm_forced_1rank']['ms_per_step'])
PY
