"""Collect per-kernel hardware counters for one bench step with rocprofv3 and aggregate them per kernel family.

Run ON the GPU box (gpurun):  cd /tmp && export TMPDIR=/tmp && python $GRAFT_REPO_ROOT/tools/pmc_collect.py
Passes (separate runs, --pmc with --kernel-trace only, as the pool requires):
  1. FETCH_SIZE   2. WRITE_SIZE   3. SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE   4. TCC_HIT / TCC_MISS / TCC_EA0_RDREQ (+ _DRAM)
Writes gpurun_out/pmc/summary.json: per family launches, HBM bytes per launch (FETCH_SIZE x2 per
/opt/skills/guides/MI355X_MICROARCH.md: gfx950 tallies 128-byte requests at 64 B; counter unit KiB) and
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs; 32 cycles per 32x32x16 bf16 MFMA) /
((GRBM_GUI_ACTIVE / 8 XCDs) x 256 CUs x 4 SIMDs) -- checked against FLOPs / 2.5 PFLOP/s from the event timings (0.238 vs 0.24)."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'gpurun_out', 'pmc')
# family -> kernel-name substrings.  'conv_tap' / 'conv_wgrad' are the two C-ABI entry points (loft_conv_tap_bf16_v /
# loft_conv_wgrad_bf16_v); each dispatches to several templates of a lock-step and a software-pipelined kernel.
FAMILIES = {'conv_tap': ['conv_tap_kernel', 'conv_tap_pipe_kernel', 'conv64_patch_kernel', 'bneck_tail_kernel', 'bneck_pair_kernel'],
            'conv_wgrad': ['conv_wgrad_kernel', 'conv_wgrad_stream_kernel', 'conv_wgrad_ring_kernel', 'conv_wgrad64_kernel']}
FAMILIES.update({k: [k] for k in [
    'conv_tap_pipe_kernel', 'conv_wgrad_stream_kernel', 'conv_wgrad_ring_kernel', 'conv64_patch_kernel',
    'roi_align_fwd_kernel', 'roi_align_fwd_sep_kernel', 'roi_align_fwd_sep8_kernel', 'roi_align_bwd_tile_kernel', 'roi_align_bwd_mfma_kernel',
    'narrow_head_bwd_kernel', 'random_sample_kernel', 'fold_pack_multi_kernel', 'fold_unpack_bwd_multi_kernel',
    'mdcn_sample_fwd_kernel', 'mdcn_sample_bwd_bin_kernel', 'mdcn_window_gather_kernel', 'nms_scan_kernel',
    'fuse_sum_relu_kernel', 'stem_mfma_kernel', 'bneck_tail_kernel', 'bneck_pair_kernel']})
# per template instance of the weight-gradient stream kernel (0 generic taps, 1 RoI maps, 2 dense 1x1 / FC, 3 stride-1 same-size taps)
EXACT = {f'conv_wgrad_stream_kernel<{i}>': f'conv_wgrad_stream_kernel<{i}>' for i in range(4)}
EXACT.update({'conv_tap_pipe_kernel<1, 0, 4, 2>': 'conv_tap_pipe_kernel<1, 0, 4, 2, false, false, false'})      # (the two-stage 256 x 256 stream schedule, either form)
PASSES = [('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE']), ('mfma', ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'GRBM_GUI_ACTIVE']),
          # round 6: where a kernel's reads are served -- L2 hits / misses and the misses' fabric requests, all
          # and those addressed to DRAM.  (What the Infinity Cache absorbs behind the fabric interface has no counter in this list.)
          ('l2', ['TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_DRAM_sum'])]


def run_pass(tag, counters, extra):
    d = os.path.join(OUT, tag)
    cmd = ['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', d, '--', sys.executable,
                                               os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '1', '--no-cpu-baseline',
                                               '--no-roofline', '--no-light', '--no-fp32', '--no-forced-comm'] + extra
    try:
        subprocess.run(['timeout', '-k', '10', '420'] + cmd, check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT, timeout=460)
    except subprocess.TimeoutExpired:
        pass
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
        os.remove(f)                    # raw per-dispatch rows are large; only the aggregate is kept
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        os.remove(f)
    return rows


def main():
    extra = sys.argv[1:]
    os.makedirs(OUT, exist_ok=True)
    agg = {}
    for tag, counters in PASSES:
        for r in run_pass(tag, counters, extra):
            name = r.get('Kernel_Name', '')
            fams = [f for f, subs in FAMILIES.items() if any(sub + '<' in name or sub + '(' in name or name.endswith(sub) for sub in subs)]
            fams += [f for f, sub in EXACT.items() if sub in name]
            for fam in fams:
                a = agg.setdefault(fam, {})
                c = r.get('Counter_Name')
                a.setdefault(c, [0.0, set()])
                a[c][0] += float(r.get('Counter_Value', 0.0))
                a[c][1].add(r.get('Dispatch_Id'))
    sys.path.insert(0, ROOT)
    from bonai_amd.build import source_hash
    out = {'_how': __doc__.strip().split('\n\n')[0], '_args': extra, '_source_hash': source_hash()}
    for fam, a in agg.items():
        n = max((len(v[1]) for v in a.values()), default=0)
        e = dict(launches=n)
        if 'FETCH_SIZE' in a:
            e['fetch_bytes_corrected_per_launch'] = a['FETCH_SIZE'][0] * 1024 * 2 / max(1, len(a['FETCH_SIZE'][1]))
        if 'WRITE_SIZE' in a:
            e['write_bytes_per_launch'] = a['WRITE_SIZE'][0] * 1024 / max(1, len(a['WRITE_SIZE'][1]))
        if 'fetch_bytes_corrected_per_launch' in e and 'write_bytes_per_launch' in e:
            e['hbm_bytes_per_launch'] = e['fetch_bytes_corrected_per_launch'] + e['write_bytes_per_launch']
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and 'GRBM_GUI_ACTIVE' in a and a['GRBM_GUI_ACTIVE'][0] > 0:
            e['mfma_busy_cycles'] = a['SQ_VALU_MFMA_BUSY_CYCLES'][0]
            e['gui_active_cycles'] = a['GRBM_GUI_ACTIVE'][0]
            e['mfma_util'] = a['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (a['GRBM_GUI_ACTIVE'][0] / 8 * 256 * 4)   # GUI_ACTIVE is summed over the 8 XCDs
        if 'SQ_BUSY_CU_CYCLES' in a:
            e['sq_busy_cu_cycles'] = a['SQ_BUSY_CU_CYCLES'][0]
        if 'TCC_HIT_sum' in a and 'TCC_MISS_sum' in a:
            nl = max(1, len(a['TCC_HIT_sum'][1]))
            hit, miss = a['TCC_HIT_sum'][0] / nl, a['TCC_MISS_sum'][0] / nl
            e['l2_hits_per_launch'], e['l2_misses_per_launch'] = hit, miss
            e['l2_hit_rate'] = hit / max(1.0, hit + miss)
            if 'TCC_EA0_RDREQ_sum' in a:
                e['ea_rdreq_per_launch'] = a['TCC_EA0_RDREQ_sum'][0] / nl
                e['ea_rdreq_dram_per_launch'] = a.get('TCC_EA0_RDREQ_DRAM_sum', [0.0])[0] / nl
                # (128-byte L2 lines; requests past the L2 are tallied at 64 B: see the FETCH_SIZE correction above)
                e['l2_request_bytes_per_launch'] = (hit + miss) * 128.0
        out[fam] = e
    json.dump(out, open(os.path.join(OUT, 'summary.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
