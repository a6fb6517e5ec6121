#!/bin/bash
# same-box A/B of an environment switch on the forced one-rank RCCL leg:  bash tools/ab_forced_env.sh LOFT_NO_LEAF_SINK=1 [rounds]
KV=$1; N=${2:-3}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5
run() { env "$@" python bench.py --force-reducer --no-cpu-baseline --no-roofline --no-light --no-fp32 --steps 10 --warmup 3 2> gpurun_out/r5/forced_err.txt \
      | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || tail -5 gpurun_out/r5/forced_err.txt; }
for ((i = 0; i < N; i++)); do
  echo "default $(run X=1)   $KV $(run $KV)"
done
