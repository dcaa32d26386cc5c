#!/bin/bash
# Everything kept under profiles/ for one round, in one gpurun call (results land in gpurun_out/<tag>_*):
#   bench lines of SURVEY 8(d) configs A..E, rocprofv3 kernel stats of the roofline workload (128 pairs) and of the
#   batch-1 tick, PMC HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and the SQ counters.
#   usage (on the GPU box): bash tools/profile_round.sh r02
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_line.err
for c in B C D E; do
  python bench.py --config $c --steps 2 --warmup 1 > $O/${TAG}_bench_config_$c.json 2> $O/${TAG}_bench_config_$c.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_tick -o tick -- python $R/tools/tick_bench.py 128 24 > $O/${TAG}_prof_tick.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_tick/*.db | head -1) $O/${TAG}_tick_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_b1 -o b1 -- python $R/tools/tick_bench.py 1 96 > $O/${TAG}_prof_b1.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_b1/*.db | head -1) $O/${TAG}_batch1_kernel_stats.csv > /dev/null
bash $R/tools/pmc_traffic.sh 128 8 > $O/${TAG}_hbm_traffic_pmc.json 2> $O/${TAG}_hbm_traffic_pmc.err
bash $R/tools/pmc_sq.sh 128 12 > /dev/null 2>&1; cp $O/pmc_sq.json $O/${TAG}_sq_counters_pmc.json
ls -la $O | grep ${TAG}_
