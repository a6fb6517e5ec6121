#!/bin/bash
# Same-box A/B of the reducer's host path on the forced one-rank RCCL leg:
#   plain step | forced, every collective through the side stream (LOFT_REDUCER_SIDE_STREAM_ONLY=1) | forced, shipped
#   bash tools/ab_reducer.sh [rounds]   ->  gpurun_out/r6/ab_reducer.txt
N=${1:-2}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/ab_reducer.txt
: > $OUT
run() {   # tag, env...
  tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm --steps 15 --warmup 5 $EXTRA 2> gpurun_out/r6/ab_reducer_err.txt \
    | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], (d.get('comm') or {}).get('exposed_ms'))" >> $OUT || tail -5 gpurun_out/r6/ab_reducer_err.txt >> $OUT
}
for ((i = 0; i < N; i++)); do
  EXTRA="" run plain LOFT_X=1
  EXTRA="--force-reducer" run forced_side_stream_only LOFT_REDUCER_SIDE_STREAM_ONLY=1
  EXTRA="--force-reducer" run forced_shipped LOFT_X=1
done
cat $OUT
