#!/bin/bash
# same-box A/B of one debug switch on the bench's timed loop:  bash tools/ab.sh LOFT_NO_ROI_SORT [rounds] [extra bench args]
# prints ms/step of alternating runs (default first), e.g. "default 37.61  LOFT_NO_ROI_SORT 37.93"
SW=$1; N=${2:-2}; shift; shift
cd "$(dirname "$0")/.."
for ((i = 0; i < N; i++)); do
  a=$(python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $SW=1 python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a   $SW $b"
done
