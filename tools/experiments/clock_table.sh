#!/bin/bash
# the level kernels' clock, three ways side by side.  One stage looped alone (tools/phase_timing.py, NDP_PT_STAGE)
# at 256 and at 128 workgroups (B = 256 / 128, G = 1: all / half of the CUs busy): ms per launch, shader cycles per tile (s_memtime) and
# the s_memtime / s_memrealtime clock of workgroup (0, 0) -- while amd-smi samples every XCD's gfx clock and the socket power.
#   bash tools/experiments/clock_table.sh [launches]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
N=${1:-40000}
for ST in 0 3; do
  for B in 256 128; do
    echo "=== stage $ST (0 forward, 3 fused backward)  B $B  launches $N"
    NDP_PT_STAGE=$ST python tools/phase_timing.py $B $N > /tmp/pt_${ST}_${B}.txt 2>&1 &
    PID=$!
    for i in $(seq 600); do                    # (the loop itself is the last 10-15 s: the samples with the socket under load)
      sleep 0.2
      kill -0 $PID 2>/dev/null || break
      amd-smi metric -g 0 --clock --power 2>/dev/null | python3 -c "
import re, sys
t = sys.stdin.read()
gfx = re.findall(r'GFX_(\d+):\s*\n\s*CLK:\s*(\d+)', t)
pw = re.findall(r'SOCKET_POWER:\s*(\d+)', t)
import time; print('%.1f' % time.time(), 'gfx MHz', ' '.join(c for _, c in gfx) if gfx else t[:300].replace(chr(10), ' / '), '| socket W', ' '.join(pw))
"
    done
    wait $PID
    grep -E "per-tick|shader clock|bwd_f|fwd8|stage|layer" /tmp/pt_${ST}_${B}.txt | head -16
  done
done
