#!/bin/bash
# the fp32 parity leg INSIDE a full bench run (after the bf16 loops, same process) under environment switches:
#   bash tools/ab_fp32_leg.sh X=1 LOFT_NO_LEAF_SINK=1 LOFT_NO_HEAD_FUSION=1
cd "$(dirname "$0")/.."
for kv in "$@"; do
  env $kv python bench.py --no-cpu-baseline --no-roofline --no-forced-comm 2>/dev/null | grep '^{' \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['value_fp32_parity']; print('$kv', 'bf16', d['ms_per_step'], 'fp32', f['ms_per_step'], 'bf16planes', f['planes_bf16']['ms_per_step'], 'mixed', f['mixed']['neck']['ms_per_step'], f['mixed']['heads']['ms_per_step'])"
done
