#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy report of the HIP library (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kres.py [name-filter]      (NDP_EXTRA_FLAGS adds compiler flags)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"] + \
      os.environ.get("NDP_EXTRA_FLAGS", "").split() + \
      ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/ndp_kres.so", os.path.join(ROOT, "deformationpyramid_amd/csrc/ndp_kernels.hip")]
out = subprocess.run(cmd, capture_output=True, text=True)
cur, rows = None, {}
for ln in out.stderr.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", ln)
    if not m:
        if "error" in ln or "warning" in ln:
            print(ln)
        continue
    body = m.group(1)
    body = re.sub(r"^\S+:\d+:\d+: +", "", body)
    if body.startswith("Function Name:"):
        cur = body.split(":")[1].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.split(":", 1)
        rows[cur][k.strip()] = v.strip()
cols = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), ("TotalSGPRs", "SGPR"), ("VGPR Spill", "spill"), ("ScratchSize [bytes/lane]", "scratch"),
        ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "LDS")]
print(f"{'kernel':24s} " + " ".join(f"{c[1]:>7s}" for c in cols))
for k, r in rows.items():
    if flt in k:
        print(f"{k:24s} " + " ".join(f"{r.get(c[0], '?'):>7s}" for c in cols))
sys.exit(out.returncode)
