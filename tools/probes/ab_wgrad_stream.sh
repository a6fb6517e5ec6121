mkdir -p gpurun_out/r2
for i in 1 2; do
LOFT_NO_WGRAD_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-saturate 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('no-wgrad-stream',d['ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-saturate 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('wgrad-stream   ',d['ms_per_step'])"
done
