# same-box A/B: split-K slots (plain stores, summed by the batched unpack) against fp32 atomics
for i in 1 2; do
python - <<'PY'
import subprocess, sys, json, os
for name, code in (('atomics', 'from bonai_amd import kernels as K; K.WGRAD_SLOTS = False'), ('slots  ', '')):
    src = code + '''
import sys, runpy
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline', '--no-saturate']
runpy.run_path('bench.py', run_name='__main__')
'''
    r = subprocess.run([sys.executable, '-c', src], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    print(name, json.loads(line[0])['ms_per_step'] if line else r.stderr[-500:])
PY
done
