#!/bin/bash
# Timing ablations of roi_align_bwd_mfma_kernel (RBM_ABL in roi_align.hip; results are wrong by construction).
# Build HERE (no GPU needed):  bash tools/probes/roi_bwd_ablate.sh build     -> tools/probes/_abl/libabl_N.so (git-ignored, travels)
# Run on the GPU box:          bash tools/probes/roi_bwd_ablate.sh run
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
D=$ROOT/tools/probes/_abl
# name:flag,flag ...   (RBM_ABL: 1 no MFMAs [the compiler then drops the staging too], 2 no operand build, 3 no gout staging, 5 scan only)
VARIANTS=${VARIANTS:-"base:-DRBM_ABL=0 noA:-DRBM_ABL=2 nogout:-DRBM_ABL=3 scan:-DRBM_ABL=5 waves3:-DRBM_WAVES=3"}
if [ "$1" = build ]; then
    mkdir -p "$D"
    objs=$(ls $ROOT/bonai_amd/csrc/*.o | grep -v f16 | grep -v roi_align.o)
    for v in $VARIANTS; do
        n=${v%%:*}; flags=$(echo ${v#*:} | tr , ' ')
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -ffp-contract=off \
            $flags -c $ROOT/bonai_amd/csrc/roi_align.hip -o $D/roi_abl_$n.o &&
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libabl_$n.so $objs $D/roi_abl_$n.o
    done
    rm -f $D/*.o; ls -la $D
else
    for v in $VARIANTS; do
        n=${v%%:*}
        echo "$v: $(LOFT_HIP_LIB=$D/libabl_$n.so timeout 300 python $ROOT/tools/probes/roi_bwd_time.py 2>&1 | grep -E '^bwd multi')"
    done
fi
