#!/bin/bash
# the level kernels' clock, three ways side by side.  One stage looped alone (tools/phase_timing.py, NDP_PT_STAGE)
# at 256 and at 128 workgroups (B = 256 / 128, G = 1 forced: all / half of the CUs busy): ms per launch, shader cycles per tile (s_memtime)
# and the s_memtime / s_memrealtime clock of workgroup (0, 0) -- while amd-smi samples every XCD's gfx clock and the socket power; then
# the bench itself under the same sampling, and the looped backward under `rocm-smi --setperfdeterminism` if the container allows it.
#   bash tools/experiments/clock_table.sh [launches]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
N=${1:-40000}
sample_until_exit() {                          # $1: pid
  for i in $(seq 900); do
    sleep 0.2
    kill -0 $1 2>/dev/null || break
    amd-smi metric -g 0 --clock --power 2>/dev/null | python3 -c "
import re, sys, time
t = sys.stdin.read()
gfx = re.findall(r'GFX_(\d+):\s*\n\s*CLK:\s*(\d+)', t)
pw = re.findall(r'SOCKET_POWER:\s*(\d+)', t)
print('%.1f' % time.time(), 'gfx MHz', ' '.join(c for _, c in gfx) if gfx else t[:300].replace(chr(10), ' / '), '| socket W', ' '.join(pw))
"
  done
}
looped() {                                     # $1 stage, $2 B
  echo "=== stage $1 (0 forward, 3 fused backward)  B $2  G 1  launches $N"
  NDP_PT_G=1 NDP_PT_STAGE=$1 python tools/phase_timing.py $2 $N > /tmp/pt_$1_$2.txt 2>&1 &
  PID=$!
  sample_until_exit $PID
  wait $PID
  grep -E "per-tick|shader clock|cycles per tile" /tmp/pt_$1_$2.txt | head -6
}
for ST in 0 3; do for B in 256 128; do looped $ST $B; done; done
echo "=== bench (3 engines x 256 slots, 3 steps)"
python bench.py --steps 3 --warmup 1 --no-alt --no-latency --no-cpu-baseline --no-roofline > /tmp/bench_clock.json 2>/dev/null &
PID=$!
sample_until_exit $PID
wait $PID
python -c "import json; d=json.load(open('/tmp/bench_clock.json')); print('bench', round(d['value'], 1), 'pairs/s')"
echo "=== perf determinism 1900 MHz (if permitted)"
if rocm-smi --setperfdeterminism 1900 2>&1 | tee /tmp/pd.txt | grep -qi "success\|set"; then
  cat /tmp/pd.txt | tail -3
  looped 3 256
  looped 0 256
  rocm-smi --resetperfdeterminism 2>&1 | tail -2
else
  cat /tmp/pd.txt | tail -5
fi
