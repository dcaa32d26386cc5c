#!/bin/bash
# Instruction mix of the engine kernels (one rocprofv3 --pmc pass, kernel trace only): vector-ALU and MFMA instruction counts, cycles
# with a vector-ALU instruction / with the matrix pipe busy.  On gfx950 the two pipes of a SIMD do not run side by side
# (tools/experiments/micro/coexec.hip): a kernel's floor is the SUM of the two.
#   usage (on the GPU box): bash tools/pmc_mix.sh [B] [ticks]   ->  gpurun_out/pmc_mix.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${1:-256}; T=${2:-12}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mix -o pmc -- python $R/tools/tick_bench.py $B $T > $R/gpurun_out/pmc_mix.log 2>&1
python - "$R/gpurun_out/pmc_mix/pmc_counter_collection.csv" "$R/gpurun_out/pmc_mix/pmc_kernel_trace.csv" $B <<'PY'
import collections, csv, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("k_eng"):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if r["Kernel_Name"].startswith("k_eng"):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
out = {}
for k, cs in sorted(acc.items()):
    row = {c: sum(v[4:]) / max(len(v[4:]), 1) for c, v in cs.items()}          # skip warm-up ticks
    row["avg_us_under_pmc"] = sum(dur[k][4:]) / max(len(dur[k][4:]), 1)
    out[k] = row
json.dump({"pairs_per_launch": int(sys.argv[3]), "kernels": out}, open(sys.argv[1].replace("pmc_mix/pmc_counter_collection.csv", "pmc_mix.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
