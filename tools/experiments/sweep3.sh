#!/bin/bash
# bench.py over (engines, slots, chunk) with the steps as one stream: bash tools/experiments/sweep3.sh "2:256:4 3:256:4 ..." [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for cfg in $1; do
  IFS=: read e s c <<< "$cfg"
  python bench.py --steps ${2:-2} --warmup 1 --engines $e --slots $s --chunk $c --pairs-per-step 8192 --no-alt --no-latency --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('engines $e slots $s chunk $c: %.1f pairs/s  host cores %.2f' % (d['value'], d['host_cores_busy']))"
done
