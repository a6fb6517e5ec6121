#!/bin/bash
# same-box A/B of an environment setting on the bench's timed loop:  bash tools/ab_env.sh LOFT_BENCH_SLOTS=dense [rounds] [bench args]
KV=$1; N=${2:-2}; shift; shift
cd "$(dirname "$0")/.."
for ((i = 0; i < N; i++)); do
  a=$(python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $KV python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a   $KV $b"
done
