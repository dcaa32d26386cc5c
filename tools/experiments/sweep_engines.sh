#!/bin/bash
# engines x slots sweep of the headline workload (what the second and third engine buy): bash tools/experiments/sweep_engines.sh [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
ST=${1:-2}
for cfg in "128 1" "128 2" "128 3" "256 1" "256 2" "256 3"; do
  set -- $cfg
  python bench.py --slots $1 --engines $2 --pairs-per-step 8192 --steps $ST --warmup 1 --no-alt --no-latency --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('slots %s engines %s: %.1f pairs/s, %.3f us per pair-iteration, %.1f iters/pair' % ('$1','$2', r['value'], 1e3*r['ms_per_iter'], r['adam_iters_per_pair']))"
done
