#!/bin/bash
# experiment: plane row strides of the bf16 backward kernels (compile-time overrides), interleaved A/B on one GPU box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
i=0
for v in "-DB8_ZROW1=168 -DB8_HROW=160" "-DB8_ZROW1=136 -DB8_HROW=160" "-DB8_ZROW1=136 -DB8_HROW=136"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude $v -o /tmp/v$i.so deformationpyramid_amd/csrc/ndp_kernels.hip 2>/dev/null
  i=$((i+1))
done
for rep in 1 2 3; do
  i=0
  for v in "z168/h160" "z136/h160" "z136/h136"; do
    echo "$v: $(NDP_HIP_LIB=/tmp/v$i.so NDP_GEMM_MODE=7 python tools/tick_bench.py 128 24 2>&1 | tail -1 | cut -c50-)"
    i=$((i+1))
  done
done
