#!/bin/bash
# round-3 measurement pass (run ON the GPU box): bench line, per-shape table, aten call sites, host tail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3m
mkdir -p "$OUT"
cd "$ROOT"
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
LOFT_DUMP_SHAPES=1 python bench.py --no-cpu-baseline --no-light > "$OUT/bench_sat.json" 2> "$OUT/shapes_sat.txt"
timeout 300 python tools/probes/aten_sites.py > "$OUT/aten_sites.txt" 2>&1
timeout 300 python tools/probes/host_tail.py 1024 > "$OUT/host_tail.txt" 2>&1
cat "$OUT/bench_line.json"
