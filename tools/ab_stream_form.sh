#!/bin/bash
# Same-box A/B of a whole training step under the four forms of the two-stage stream schedule (include/loft_hip.h
# loft_conv_stream_form; all bit-identical):  bash tools/ab_stream_form.sh "0 3 0 3 1 2" -> gpurun_out/r6/ab_stream_form.txt
FORMS=${1:-"0 3 0 3"}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/ab_stream_form.txt
: > $OUT
for f in $FORMS; do
  python bench.py --stream-form $f --no-cpu-baseline --no-roofline --no-light --no-fp32 --no-forced-comm --steps 20 --warmup 5 2> gpurun_out/r6/ab_stream_form_err.txt \
    | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('form $f', d['ms_per_step'], d['value'])" >> $OUT || tail -5 gpurun_out/r6/ab_stream_form_err.txt >> $OUT
done
cat $OUT
