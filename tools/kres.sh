#!/bin/bash
# kernel resource usage summary: tools/kres.sh [extra hipcc flags]
cd "$(dirname "$0")/../deformationpyramid_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/kres.so ndp_kernels.hip 2>&1 |
python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    if "error" in line: print(line, end="")
    m=re.search(r"(Function Name|Name|VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k in ("Function Name","Name"):
        if cur: print(cur)
        cur={"name":v}
    else: cur[k.replace(" [bytes/lane]","").replace(" [waves/SIMD]","").replace(" [bytes/block]","")]=v
if cur: print(cur)
'
