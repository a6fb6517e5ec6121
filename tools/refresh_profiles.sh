#!/bin/bash
# Refresh the measured artifacts kept under profiles/ (run ON the GPU box:  gpurun -- 'bash tools/refresh_profiles.sh').
# Counters and traces are separate rocprofv3 runs (--pmc only with --kernel-trace); raw traces stay in /tmp, only summaries
# are copied to gpurun_out/refresh/ (then by hand into profiles/roundN_*).  Every file that bench.py quotes carries the source
# hash of the tree it was measured on (bonai_amd.build.source_hash); bench.py refuses a file whose hash is not the running tree's.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
HASH=$(cd "$ROOT" && python -c "from bonai_amd.build import source_hash; print(source_hash())")
python "$ROOT/tools/pmc_collect.py" > /dev/null 2>&1
cp "$ROOT/gpurun_out/pmc/summary.json" "$OUT/pmc_traffic.json"
cp "$OUT/pmc_traffic.json" "$ROOT/profiles/round6_pmc_traffic.json"      # so that the bench run below quotes it
W=5; K=20
for mode in serial default; do
    rm -rf /tmp/prof_$mode
    if [ $mode = serial ]; then export LOFT_NO_SIDE_STREAM=1; else unset LOFT_NO_SIDE_STREAM; fi
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python "$ROOT/bench.py" --no-cpu-baseline --no-light \
        --no-fp32 --no-forced-comm --steps $K --warmup $W > "$OUT/bench_under_rocprof_$mode.json" 2> /dev/null
    f=$(find /tmp/prof_$mode -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$mode.csv"
    # steps the summary covers: W warm-up + K timed + 2 instrumented (bench.py's roofline leg)
    echo "{\"source_hash\": \"$HASH\", \"steps_profiled\": $((W + K + 2)), \"command\": \"bench.py --no-cpu-baseline --no-light --no-fp32 --no-forced-comm --steps $K --warmup $W\", \"mode\": \"$mode\"}" > "$OUT/kernel_stats_$mode.meta.json"
done
unset LOFT_NO_SIDE_STREAM
cp "$OUT/kernel_stats_serial.csv" "$ROOT/profiles/round6_bench_kernel_stats_serial.csv"
cp "$OUT/kernel_stats_serial.meta.json" "$ROOT/profiles/round6_bench_kernel_stats_serial.meta.json"
# The bench line itself is NOT measured here but by tools/refresh_bench_line.sh in a gpurun call of its own (a box that has run
# nothing else), after pmc_traffic.json / kernel_stats_* have been copied into profiles/ (bench.py quotes them by source hash).
rm -rf "$ROOT/gpurun_out/pmc"
ls -la "$OUT"
