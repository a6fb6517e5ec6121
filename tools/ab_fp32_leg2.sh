#!/bin/bash
# the fp32 parity leg inside bench runs with / without the roofline leg and the shape dump in front of it
cd "$(dirname "$0")/.."
p() { grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['value_fp32_parity']; print('$1', 'bf16', d['ms_per_step'], 'fp32', f['ms_per_step'])"; }
python bench.py --no-cpu-baseline --no-forced-comm 2>/dev/null | p roofline
LOFT_DUMP_SHAPES=1 python bench.py --no-cpu-baseline --no-forced-comm 2>/dev/null | p roofline+dump
python bench.py --no-cpu-baseline --no-forced-comm --no-roofline 2>/dev/null | p noroofline
