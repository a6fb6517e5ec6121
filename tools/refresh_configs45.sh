#!/bin/bash
# Bench lines + rocprofv3 kernel stats of BASELINE configs 4 (DCNv2) and 5 (HRNet-W32) -- parity-test cases, timed for reference.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for c in config4_dcn:loft_foa_r50_fpn_mdconv_c3-c5_2x_bonai.py config5_hrnet:loft_foa_hrnetv2p_w32_2x_bonai.py; do
    tag=${c%%:*}; cfg=${c##*:}
    python "$ROOT/bench.py" --config $cfg --no-cpu-baseline > "$OUT/${tag}_bench_line.json" 2> /dev/null
    rm -rf /tmp/prof_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python "$ROOT/bench.py" --config $cfg --no-cpu-baseline \
        --no-roofline --steps 10 > /dev/null 2>&1
    f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/${tag}_kernel_stats.csv"
    cat "$OUT/${tag}_bench_line.json" | cut -c1-400
done
