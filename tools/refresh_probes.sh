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
# timing ablations of the stream kernel (results wrong by construction): 8 = no global->LDS copies in the loop, 10 = no fragment reads, 12 = no MFMAs
STREAM_VARS=0,8,10,12 PIPE_VARS=0 timeout 300 python tools/pipe_trace.py 2>&1 | head -8 > "$OUT/pipe_ablations.txt"
timeout 300 python tools/probes/copy_sites.py > "$OUT/copy_sites.txt" 2>&1
timeout 300 python tools/probes/roi_bwd_time.py > "$OUT/roi_align_time.txt" 2>&1
timeout 300 python tools/probes/roi_pairs.py > "$OUT/roi_pairs.txt" 2>&1
# timing ablations of the RoIAlign backward: build the variant libraries first, HERE:  bash tools/probes/roi_bwd_ablate.sh build
[ -d tools/probes/_abl ] && timeout 500 bash tools/probes/roi_bwd_ablate.sh run > "$OUT/roi_bwd_ablations.txt" 2>&1
timeout 300 python tools/probes/bucket_timeline.py 2>&1 | grep -E "^  bucket|reducer.begin" > "$OUT/bucket_timeline.txt"
timeout 300 python tools/probes/hipblaslt_ref.py > "$OUT/hipblaslt_ref.txt" 2>&1
timeout 300 python tools/probes/aten_sites.py > "$OUT/aten_sites.txt" 2>&1
( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/fill_prof
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/fill_prof -- python "$ROOT/bench.py" --no-cpu-baseline --no-saturate --no-roofline > /dev/null 2>&1
  f=$(find /tmp/fill_prof -name '*kernel_trace.csv' | head -1)
  python "$ROOT/tools/probes/fill_profile.py" "$f" > "$OUT/fill_profile.txt" 2>&1
  python "$ROOT/tools/probes/gap_profile.py" "$f" >> "$OUT/fill_profile.txt" 2>&1 )
timeout 400 python tools/bench_conv.py fwd stream dgrad stream_dgrad wgrad wgrad_stream > "$OUT/bench_conv.txt" 2>&1
timeout 600 bash tools/pmc_traffic_shapes.sh fwd > "$OUT/pmc_traffic_fwd.txt" 2>&1
timeout 600 bash tools/pmc_traffic_shapes.sh wgrad > "$OUT/pmc_traffic_wgrad.txt" 2>&1
ls -la "$OUT"
