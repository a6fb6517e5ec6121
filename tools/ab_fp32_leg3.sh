#!/bin/bash
cd "$(dirname "$0")/.."
p() { grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['value_fp32_parity']; print('$1', 'bf16', d['ms_per_step'], 'fp32', f['ms_per_step'])"; }
python bench.py --no-forced-comm 2>/dev/null | p cpu_baseline_on
python bench.py --no-cpu-baseline 2>/dev/null | p forced_on
python bench.py --no-cpu-baseline --no-forced-comm 2>/dev/null | p both_off
