#!/bin/bash
# one stage alone (NDP_PT_STAGE, default 0 = forward) of phase-timing variant builds, one line each: bash tools/experiments/pt_ab.sh "v0 v1" [B] [ticks]
#   <variant>: ms per launch | thread 0's cycles per tile by phase (the fwd8 / bwd_f table of tools/phase_timing.py, zero rows dropped)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
VARS=$1; B=${2:-256}; T=${3:-16}
ST=${NDP_PT_STAGE:-0}
for v in $VARS; do
  NDP_PT_LIB=tools/experiments/var/$v.so NDP_PT_STAGE=$ST python tools/phase_timing.py $B $T 2>&1 | python -c "
import sys, re
ms, rows, on, ghz = None, [], False, ''
for ln in sys.stdin:
    if ln.startswith('per-tick'):
        ms = eval(ln.split(':', 1)[1])
    if ln.startswith('shader clock'):
        ghz = ln.split(':')[1].split('(')[0].strip()
    if ln.startswith('fwd8 (fp16') or ln.startswith('bwd_f (fused)'):
        on = ('$ST' == '0') == ln.startswith('fwd8'); tot = ln.split(':')[1].split()[0]; continue
    if on and ln.startswith('   '):
        m = re.match(r'\s+(.*?)\s+(\d+)\s+[\d.]+ %', ln)
        if m and int(m.group(2)): rows.append(m.group(2))
    elif on: on = False
print('%-16s %.4f ms  %s | %s' % ('$v', max(ms) if ms else -1, ghz, ' '.join(rows)))
"
done
