"""Same-box A/B of the bench step: python tools/probes/ab_generic.py NAME1 'python code' NAME2 'python code' ... (each code string
runs before bench.py in a fresh process; two rounds, interleaved)."""
import json, subprocess, sys
pairs = list(zip(sys.argv[1::2], sys.argv[2::2]))
extra = ['--no-cpu-baseline', '--no-roofline', '--no-saturate']
for rnd in range(2):
    for name, code in pairs:
        src = code + f'''
import sys, runpy
sys.argv = ['bench.py'] + {extra!r}
runpy.run_path('bench.py', run_name='__main__')
'''
        r = subprocess.run([sys.executable, '-c', src], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        print(f'{name:16s}', json.loads(line[0])['ms_per_step'] if line else r.stderr[-800:], flush=True)
