#!/bin/bash
# rocprofv3 --stats of the bench's fp32-parity-mode loop at the trained-RPN load (the fp32 kernels have their own names: planes
# template instances, split / absmax passes, *_f32 kernels) -> gpurun_out/r6/fp32_kernel_stats.csv + a per-step summary
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_f32
LOFT_BENCH_F32_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f32 -- python "$ROOT/bench.py" --no-cpu-baseline \
    --no-light --no-forced-comm --no-roofline --steps 2 --warmup 1 > "$OUT/fp32_profile_bench.json" 2> /dev/null
f=$(find /tmp/prof_f32 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/fp32_kernel_stats.csv"
python - <<PY
import csv
def f(x):
    try: return float(x)
    except: return None
rows=[r for r in csv.DictReader(open('$OUT/fp32_kernel_stats.csv')) if f(r['TotalDurationNs']) is not None and f(r['Calls']) is not None]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f'# all kernels of the run (3 bf16 steps + 10 fp32 steps: 5 warm-up + 5 timed): {tot/1e6:.1f} ms')
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}% calls {int(float(r['Calls'])):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
