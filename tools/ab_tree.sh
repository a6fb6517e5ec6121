#!/bin/bash
# same-box A/B of two source trees on the bench's timed loop:  bash tools/ab_tree.sh _ab_prev [rounds] [bench args]
# (the other tree = `git archive <rev> bonai_amd bench.py configs oracle include | tar -x -C _ab_prev` + the built .so files)
T=$1; N=${2:-3}; shift; shift
cd "$(dirname "$0")/.."
for ((i = 0; i < N; i++)); do
  a=$(python $T/bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(python bench.py --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$T $a   tree $b"
done
