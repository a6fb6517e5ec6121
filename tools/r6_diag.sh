#!/bin/bash
# round-6 diagnosis on one box: (1) tail of the step, plain vs forced reducer; (2) PMC of the stream kernel on the FOA shape
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for mode in plain forced; do
  rm -rf /tmp/tg_$mode
  extra=""; [ $mode = forced ] && extra="--force-reducer"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tg_$mode -- python "$ROOT/bench.py" --no-cpu-baseline --no-roofline --no-light --no-fp32 \
      --no-forced-comm --steps 8 --warmup 3 $extra > /dev/null 2>&1
  f=$(find /tmp/tg_$mode -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python "$ROOT/tools/tail_gap.py" "$f" 1200 > "$OUT/tail_gap_$mode.txt"
  head -3 "$OUT/tail_gap_$mode.txt"
done

python "$ROOT/tools/pmc_stream.py" foa stream0 stream8 > "$OUT/pmc_stream_stdout.txt" 2>&1
tail -60 "$OUT/pmc_stream_stdout.txt"
