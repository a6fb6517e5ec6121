#!/bin/bash
# Same-box A/B of the reducer's host path on the forced one-rank RCCL leg:
#   plain / forced step, each with the batched unpack launches on the main stream (LOFT_NO_UNPACK_STREAM=1: round 5) and as shipped
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
  EXTRA="--force-reducer" run forced_shipped LOFT_X=1
  EXTRA="--force-reducer" run forced_dryrun_no_collective LOFT_REDUCER_DRYRUN=1
  EXTRA="--force-reducer" run group_alive_reducer_off LOFT_BENCH_INIT_ONLY=1
done
cat $OUT
