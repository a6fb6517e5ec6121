#!/bin/bash
# Outputs of the round-2 probes quoted in DESIGN.md (run ON the GPU box; copied by hand into profiles/round2_probes/).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/probes
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/probes/wgrad_splits.py > "$OUT/wgrad_splits.txt" 2>&1
timeout 100 tools/probes/atomic_scope > "$OUT/atomic_scope.txt" 2>&1
timeout 300 python tools/probes/kshallow_variants.py > "$OUT/kshallow_variants.txt" 2>&1
timeout 300 tools/probes/gemm_quad > "$OUT/gemm_quad.txt" 2>&1
timeout 300 python tools/probes/host_tail.py 1024 > "$OUT/host_tail.txt" 2>&1
timeout 300 python tools/probes/host_tail.py 128 >> "$OUT/host_tail.txt" 2>&1
STREAM_TRACE=1 PIPE_VARS=0 timeout 300 python tools/pipe_trace.py > "$OUT/pipe_trace.txt" 2>&1
timeout 400 python tools/bench_conv.py fwd stream dgrad stream_dgrad wgrad wgrad_stream > "$OUT/bench_conv.txt" 2>&1
timeout 600 bash tools/pmc_traffic_shapes.sh fwd > "$OUT/pmc_traffic_fwd.txt" 2>&1
timeout 600 bash tools/pmc_traffic_shapes.sh wgrad > "$OUT/pmc_traffic_wgrad.txt" 2>&1
ls -la "$OUT"
