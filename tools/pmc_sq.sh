#!/bin/bash
# SQ / GRBM counters of the engine kernels (one rocprofv3 --pmc pass, kernel trace only):
#   MFMA pipe busy cycles, wave cycles split into active / waiting, LDS bank conflicts, GUI-active cycles.
#   usage (on the GPU box): bash tools/pmc_sq.sh [B] [ticks]   ->  gpurun_out/pmc_sq.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${1:-128}; T=${2:-12}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o pmc -- python $R/tools/tick_bench.py $B $T > $R/gpurun_out/pmc_sq.log 2>&1
python - "$R/gpurun_out/pmc_sq/pmc_counter_collection.csv" "$R/gpurun_out/pmc_sq/pmc_kernel_trace.csv" $B <<'PY'
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
json.dump({"pairs_per_launch": int(sys.argv[3]), "kernels": out}, open(sys.argv[1].replace("pmc_sq/pmc_counter_collection.csv", "pmc_sq.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
