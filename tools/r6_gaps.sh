#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for mode in plain forced; do
  rm -rf /tmp/tg_$mode
  extra=""; [ $mode = forced ] && extra="--force-reducer"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg_$mode -- python "$ROOT/bench.py" --no-cpu-baseline --no-roofline --no-light --no-fp32 \
      --no-forced-comm --steps 8 --warmup 3 $extra > /dev/null 2>&1
  f=$(find /tmp/tg_$mode -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python "$ROOT/tools/step_gaps.py" "$f" 12 > "$OUT/step_gaps_$mode.txt"
  cat "$OUT/step_gaps_$mode.txt" | cut -c1-200
done
