#!/bin/bash
# socket power and clocks while the bench runs (is the tick power-limited?): bash tools/experiments/clock_watch.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
rocm-smi --showmaxpower 2>&1 | grep -i "Max Graphics"
python bench.py --steps 3 --warmup 1 --no-alt --no-latency --no-cpu-baseline --no-roofline > /tmp/b.json 2>/dev/null &
PID=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showuse 2>&1 | grep -i "Socket\|sclk\|GPU use" | sed 's/GPU\[0\]\t\t: //' | tr '\n' '|'; echo
  sleep 1
done
wait $PID
python -c "import json; d=json.load(open('/tmp/b.json')); print('bench', d['value'])"
