#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for rep in 1 2; do
for cfg in "128 3 4096" "256 2 8192" "256 3 8192" "256 2 4096" "512 1 8192"; do
  set -- $cfg
  python bench.py --slots $1 --engines $2 --pairs-per-step $3 --steps 2 --warmup 1 --no-alt --no-latency --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('slots %s engines %s pairs/step %s: %.1f pairs/s, %.3f us per pair-iteration' % ('$1','$2','$3', r['value'], 1e3*r['ms_per_iter']))"
done; done
