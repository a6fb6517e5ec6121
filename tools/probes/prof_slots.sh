export TMPDIR=/tmp; cd /tmp
for mode in slots atomics; do
rm -rf /tmp/ps_$mode
if [ $mode = atomics ]; then PRE="from bonai_amd import kernels as K; K.WGRAD_SLOTS = False"; else PRE=""; fi
LOFT_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$mode -- python -c "$PRE
import sys, runpy
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline', '--no-saturate', '--steps', '10']
runpy.run_path('$GRAFT_REPO_ROOT/bench.py', run_name='__main__')" > /dev/null 2>&1
f=$(find /tmp/ps_$mode -name '*kernel_stats.csv' | head -1)
echo "== $mode"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r['Name'] for k in ('wgrad', 'unpack', 'fill', 'Fill')):
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
print('total', sum(float(r['TotalDurationNs']) for r in rows) / 1e6)
PY
done
