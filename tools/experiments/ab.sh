#!/bin/bash
# interleaved A/B of variant libraries on one GPU box: bash tools/experiments/ab.sh "v0 v1 v2" [reps] [B] [ticks]
# (variants: tools/experiments/var/<name>.so, built by build_variant.sh; NDP_TICK_HASH=1 adds a digest of the engine state after the ticks)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
VARS=${1:-"v0 v1"}; REPS=${2:-3}; B=${3:-128}; T=${4:-24}
for rep in $(seq $REPS); do
  for v in $VARS; do
    echo "$v: $(NDP_HIP_LIB=$R/tools/experiments/var/$v.so python tools/tick_bench.py $B $T 2>&1 | tail -2 | cut -c1-200 | tr '\n' ' ')"
  done
done
