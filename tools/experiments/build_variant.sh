#!/bin/bash
# Build a variant of libndp_hip.so with extra -D flags next to the product library (same flags as deformationpyramid_amd/_native.py):
#   bash tools/experiments/build_variant.sh tools/experiments/var/lib4w.so -DNDP_EXPERIMENT_FWD_4W
# and run anything against it with NDP_HIP_LIB=<that file>.
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
ID=$(cd $R && python -c "from deformationpyramid_amd import _native as n; print(n.source_id())")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-pass-failed "-DNDP_BUILD_ID=\"$ID\"" "$@" -o "$OUT" $R/deformationpyramid_amd/csrc/ndp_kernels.hip
