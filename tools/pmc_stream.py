"""Hardware counters of the 256 x 256 stream kernel on ONE shape, variant by variant (what the K loop waits for).
Run ON the GPU box:  cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/pmc_stream.py [shape] [variants...]
Every pass is its own rocprofv3 run (--pmc with --kernel-trace only).  Writes gpurun_out/r6/pmc_stream_<shape>.txt."""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PASSES = [          # (counters that exist in this rocprofv3's gfx950 list; four per pass; every pass under its own `timeout`)
    ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY'],
    ['SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_VALU'],
    ['SQ_WAIT_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_ADDR_CONFLICT'],
    ['SQ_INST_LEVEL_VMEM', 'SQ_INSTS_VMEM_RD', 'SQ_INST_LEVEL_LDS', 'SQ_INSTS_LDS'],
    ['SQ_VMEM_TA_ADDR_FIFO_FULL', 'SQ_VMEM_TA_CMD_FIFO_FULL', 'SQ_LDS_CMD_FIFO_FULL', 'SQ_LDS_DATA_FIFO_FULL'],
    ['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC'],
    ['TCP_PENDING_STALL_CYCLES_sum', 'TCP_TCC_READ_REQ_LATENCY_sum', 'TCP_TCC_READ_REQ_sum', 'TA_TA_BUSY_sum'],
    ['TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_EA0_RDREQ_sum', 'TCC_REQ_sum'],
]
PASS_TIMEOUT = 150


def run(shape, var, counters, tag):
    d = f'/tmp/pmcs_{tag}'
    shutil.rmtree(d, ignore_errors=True)
    cmd = ['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', d, '--', sys.executable,
                                               os.path.join(ROOT, 'tools', 'stream_shape.py'), shape, str(var)]
    try:
        r = subprocess.run(['timeout', '-k', '10', str(PASS_TIMEOUT)] + cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd='/tmp',
                           text=True, timeout=PASS_TIMEOUT + 30)
    except subprocess.TimeoutExpired:
        return {}, 'timed out'
    acc, n = {}, {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if 'conv_tap_pipe_kernel' not in row.get('Kernel_Name', '') and 'conv_tap_w4' not in row.get('Kernel_Name', ''):
                continue
            c = row['Counter_Name']
            acc[c] = acc.get(c, 0.0) + float(row['Counter_Value'])
            n[c] = n.get(c, 0) + 1
    if not acc:
        return {}, r.stdout[-600:]
    return {c: acc[c] / n[c] for c in acc}, ''


def main():
    import torch  # noqa -- (fail early if the box has no torch)
    sys.path.insert(0, ROOT)
    from bonai_amd import kernels as K
    shape = sys.argv[1] if len(sys.argv) > 1 else 'foa'
    names = sys.argv[2:] or ['stream0', 'stream8']
    variants = {'stream0': K.CONV_STREAM256, 'stream8': K.CONV_STREAM256 | (8 << 12), 'stream10': K.CONV_STREAM256 | (10 << 12),
                'stream12': K.CONV_STREAM256 | (12 << 12), 'ring32': K.CONV_RING32, 'w4': K.CONV_W4, 'lockstep': K.CONV_T256_FAST}
    for k in names:
        if k not in variants:
            variants[k] = int(k, 0)
    out = os.path.join(ROOT, 'gpurun_out', 'r6')
    os.makedirs(out, exist_ok=True)
    lines = [f'# rocprofv3 --pmc, per launch averages, shape {shape} (tools/pmc_stream.py); SQ_* cycle counters are per-SIMD sums in quad-cycles where the guide says so']
    res = {v: {} for v in names}
    for pi, counters in enumerate(PASSES):
        for v in names:
            got, err = run(shape, variants[v], counters, f'{v}_{pi}')
            if not got:
                lines.append(f'# pass {pi} {v}: no rows ({err.strip()[-200:]!r})')
            res[v].update(got)
            with open(os.path.join(out, f'pmc_stream_{shape}.partial.txt'), 'a') as fh:       # (kept if a later pass hangs)
                fh.write(f'{v} pass {pi}: {got or err}\n')
    allc = []
    for counters in PASSES:
        allc += [c for c in counters if any(c in res[v] for v in names)]
    lines.append(f'{"counter":40s}' + ''.join(f'{v:>18s}' for v in names))
    for c in allc:
        lines.append(f'{c:40s}' + ''.join(f'{res[v].get(c, float("nan")):18.4g}' for v in names))
    open(os.path.join(out, f'pmc_stream_{shape}.txt'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
