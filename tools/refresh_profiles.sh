#!/bin/bash
# Refresh the measured artifacts kept under profiles/ (run ON the GPU box:  gpurun -- 'bash tools/refresh_profiles.sh').
# Counters and traces are separate rocprofv3 runs (--pmc only with --kernel-trace); raw traces stay in /tmp, only summaries
# are copied to gpurun_out/refresh/ (then by hand into profiles/).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python "$ROOT/tools/pmc_collect.py" > /dev/null 2>&1
cp "$ROOT/gpurun_out/pmc/summary.json" "$OUT/pmc_traffic.json"
cp "$OUT/pmc_traffic.json" "$ROOT/profiles/round3_pmc_traffic.json"      # bench.py quotes traffic / MFMA utilisation from it
python "$ROOT/bench.py" > "$OUT/bench_line.json" 2> "$OUT/bench.err"
for mode in default serial; do
    rm -rf /tmp/prof_$mode
    if [ $mode = serial ]; then export LOFT_NO_SIDE_STREAM=1; else unset LOFT_NO_SIDE_STREAM; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python "$ROOT/bench.py" --no-cpu-baseline --no-light \
        > "$OUT/bench_under_rocprof_$mode.json" 2> /dev/null
    f=$(find /tmp/prof_$mode -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$mode.csv"
done
unset LOFT_NO_SIDE_STREAM
rm -rf "$ROOT/gpurun_out/pmc"
ls -la "$OUT"
cat "$OUT/bench_line.json"
