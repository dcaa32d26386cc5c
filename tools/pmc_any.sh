#!/bin/bash
# Any set of PMC counters per engine kernel (one rocprofv3 --pmc pass per call, kernel trace only):
#   bash tools/pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY ..." [B] [ticks] [out.json]       (NDP_HIP_LIB=<variant> to profile a variant)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CNT=$1; B=${2:-256}; T=${3:-12}; OUT=${4:-$R/gpurun_out/pmc_any.json}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
D=$R/gpurun_out/pmc_any_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $D -o pmc -- python $R/tools/tick_bench.py $B $T > $D.log 2>&1
python - "$D/pmc_counter_collection.csv" "$D/pmc_kernel_trace.csv" $B "$OUT" <<'PY'
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
json.dump({"pairs_per_launch": int(sys.argv[3]), "kernels": out}, open(sys.argv[4], "w"), indent=1)
for k, row in out.items():
    print(k, " ".join(f"{c}={v:.4g}" for c, v in row.items()))
PY
rm -rf $D $D.log
