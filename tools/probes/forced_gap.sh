#!/bin/bash
# kernel-trace gap profile of the forced one-rank reducer leg, default vs an env switch:  bash tools/probes/forced_gap.sh LOFT_NO_UNIT_GRAD=1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/r5
cd /tmp
for kv in X=1 "$1"; do
    rm -rf /tmp/prof_fg
    env $kv rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_fg -- python $ROOT/bench.py --force-reducer --no-cpu-baseline --no-light \
        --no-fp32 --no-roofline --steps 8 --warmup 4 > /dev/null 2>&1
    f=$(find /tmp/prof_fg -name "*kernel_trace.csv" | head -1)
    echo "=== $kv"
    python $ROOT/tools/probes/gap_profile.py $f
done
