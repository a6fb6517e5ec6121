#!/bin/bash
# A/B builds of the RoIAlign kernels on the RoI lists of a bench step (tools/probes/roi_bwd_time.py prints forward and backward):
#   bash tools/probes/roi_fwd_ab.sh build     (here)      bash tools/probes/roi_fwd_ab.sh run   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
D=$ROOT/tools/probes/_abl
VARIANTS=${VARIANTS:-"fxcd:-DRF8_XCD=1 fnoxcd:-DRF8_XCD=0"}
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
    for rep in 1 2; do
    for v in $VARIANTS; do
        n=${v%%:*}
        for sort in 0 1; do
            echo "$v roi_sort=$sort: $(LOFT_ROI_SORT=$( [ $sort = 1 ] && echo 1 ) LOFT_HIP_LIB=$D/libabl_$n.so timeout 300 python $ROOT/tools/probes/roi_bwd_time.py 2>&1 | grep -E '^fwd' | tr '\n' ' ')"
        done
    done
    done
fi
