#!/bin/bash
# experiment: sensitivity of the matrix-pipe NN kernel to its error bound (compile-time override), run on the GPU box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
i=0
for v in "-DNN2_ERRC=3.814697265625e-06f" "-DNN2_ERRC=9.5367431640625e-07f" "-DNN2_ERRC=1.52587890625e-05f"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude $v -o /tmp/v$i.so deformationpyramid_amd/csrc/ndp_kernels.hip 2>/dev/null
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for v in "2^-18" "2^-20" "2^-16"; do
    echo "errc $v R^2: $(NDP_HIP_LIB=/tmp/v$i.so NDP_NN_MODE=2 python tools/tick_bench.py 128 24 2>&1 | tail -1 | cut -c50-)"
    i=$((i+1))
  done
done
