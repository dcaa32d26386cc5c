#!/bin/bash
# interleaved A/B of engine modes on one GPU box: bash tools/experiments/ab_modes.sh "7 23" [reps] [B] [ticks]   (gemm_mode values)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
MODES=${1:-"7 23"}; REPS=${2:-3}; B=${3:-128}; T=${4:-24}
for rep in $(seq $REPS); do
  for m in $MODES; do
    echo "gemm_mode $m: $(NDP_GEMM_MODE=$m python tools/tick_bench.py $B $T 2>&1 | tail -1 | cut -c40-)"
  done
done
