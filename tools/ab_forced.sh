#!/bin/bash
# same-box A/B of two trees on the forced one-rank RCCL leg:  bash tools/ab_forced.sh _ab_prev [rounds]
T=$1; N=${2:-2}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
for ((i = 0; i < N; i++)); do
  for t in $T .; do
    python $t/bench.py --force-reducer --no-cpu-baseline --no-roofline --no-light --no-fp32 --steps 10 --warmup 3 2> gpurun_out/r5/forced_err.txt \
      | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced $t', d['ms_per_step'], d.get('comm'))" || tail -5 gpurun_out/r5/forced_err.txt
  done
done
