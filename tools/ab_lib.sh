#!/bin/bash
# same-box A/B of two builds of the library on the bench's timed loop:  bash tools/ab_lib.sh <other libloft_hip.so> [rounds] [bench args]
LIB=$1; N=${2:-3}; shift; shift
cd "$(dirname "$0")/.."
for ((i = 0; i < N; i++)); do
  a=$(python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(LOFT_HIP_LIB=$LIB python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "tree $a   $LIB $b"
done
