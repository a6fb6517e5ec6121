#!/bin/bash
# tests/stress/trainer_race.sh [runs-per-switch] [parallel] -> gpurun_out/trainer_race.txt (table) + trainer_race_raw.txt
N=${1:-30}; P=${2:-3}
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
RAW=gpurun_out/trainer_race_raw.txt; : > $RAW
for sw in RACE_OLD_NN - LOFT_NO_WGRAD_STREAM LOFT_NO_ZERO_POOL LOFT_NO_PREPACK LOFT_NO_UNPACK_QUEUE LOFT_NO_SIDE_STREAM LOFT_NO_GRAD_SINK; do
  for ((i = 0; i < N; i += P)); do
    for ((j = 0; j < P && i + j < N; j++)); do
      ( if [ "$sw" != "-" ]; then export $sw=1; fi
        timeout 300 python tests/stress/trainer_race.py 3 256 $(( (i + j) % 3 )) 2>&1 | grep RESULT || echo "RESULT CRASH switches=$sw" ) >> $RAW &
    done
    wait
  done
done
python - <<'PY' | tee gpurun_out/trainer_race.txt
import collections, re
rows = collections.OrderedDict()
for l in open('gpurun_out/trainer_race_raw.txt'):
    m = re.match(r'RESULT (\S+) switches=(\S+)(?: worst=(\S+) rel=(\S+) nbad=(\d+))?', l)
    if not m:
        continue
    r = rows.setdefault(m.group(2), dict(n=0, bad=0, crash=0, worst=0.0, who=''))
    r['n'] += 1
    r['bad'] += m.group(1) == 'BAD'
    r['crash'] += m.group(1) == 'CRASH'
    if m.group(4) and float(m.group(4)) > r['worst']:
        r['worst'], r['who'] = float(m.group(4)), m.group(3)
print(f"{'switch':28s} runs  bad  crash  worst rel. error (parameter)")
for k, r in rows.items():
    print(f"{k:28s} {r['n']:4d} {r['bad']:4d} {r['crash']:5d}  {r['worst']:.3e} ({r['who']})")
PY
