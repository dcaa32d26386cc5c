#!/bin/bash
# HBM traffic of the engine kernels from rocprofv3 PMC counters (separate passes for FETCH_SIZE and
# WRITE_SIZE: they do not fit one TCC pass -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   usage (on the GPU box): bash tools/pmc_traffic.sh [B] [ticks]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${1:-64}; T=${2:-8}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/tools/tick_bench.py $B $T > $R/gpurun_out/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv $R/gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv $B
