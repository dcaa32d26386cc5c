#!/bin/bash
# Everything kept under profiles/ for one round, in one gpurun call (results land in gpurun_out/<tag>_*):
#   bench lines of SURVEY 8(d) configs A..E (the default line carries BOTH arithmetics: `value` = the default split path, `alt` = the
#   bitwise fp32-MFMA path), rocprofv3 kernel stats of the roofline workload (128 pairs per engine tick) in both arithmetics and of
#   the batch-1 tick, PMC HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and the SQ counters, both arithmetics.
#   usage (on the GPU box): bash tools/profile_round.sh r03
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_line.err
for c in B C D E; do
  python bench.py --config $c --steps 2 --warmup 1 > $O/${TAG}_bench_config_$c.json 2> $O/${TAG}_bench_config_$c.err
done
python bench.py --slots 64 --engines 1 --pairs-per-step 2048 --steps 2 --warmup 1 --no-alt --no-latency --no-cpu-baseline > $O/${TAG}_bench_batch64.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
# rocprofv3 kernel stats of the tick: default (split) arithmetic, then the bitwise one
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_tick -o tick -- python $R/tools/tick_bench.py 128 24 > $O/${TAG}_prof_tick.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_tick256 -o tick -- python $R/tools/tick_bench.py 256 24 > $O/${TAG}_prof_tick256.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_tick256/*.db | head -1) $O/${TAG}_tick_kernel_stats_256pairs.csv > /dev/null
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_tick/*.db | head -1) $O/${TAG}_tick_kernel_stats.csv > /dev/null
NDP_GEMM_MODE=0 NDP_NN_MODE=0 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_tick_bitwise -o tick -- python $R/tools/tick_bench.py 128 24 > $O/${TAG}_prof_tick_bitwise.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_tick_bitwise/*.db | head -1) $O/${TAG}_tick_kernel_stats_bitwise.csv > /dev/null
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_b1 -o b1 -- python $R/tools/tick_bench.py 1 96 > $O/${TAG}_prof_b1.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_b1/*.db | head -1) $O/${TAG}_batch1_kernel_stats.csv > /dev/null
# whole bench under the kernel trace (short): share of the tick kernels, the final warps, the slot loads
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-alt --no-roofline --no-latency --no-cpu-baseline > $O/${TAG}_prof_bench.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_bench/*.db | head -1) $O/${TAG}_bench_kernel_stats.csv > /dev/null
# the two-launch split backward of round 3 (gemm_mode 7 | 16) next to the fused one: kernel stats of the same tick
NDP_GEMM_MODE=23 rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_tick_2launch -o tick -- python $R/tools/tick_bench.py 128 24 > $O/${TAG}_prof_tick_2launch.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_tick_2launch/*.db | head -1) $O/${TAG}_tick_kernel_stats_2launch.csv > /dev/null
# what the second and third engine buy
bash $R/tools/experiments/sweep3.sh "2:256:4 3:256:4 2:256:4 3:256:4" 2 > $O/${TAG}_engines_sweep.txt 2>&1
python $R/bench.py --steps 3 --warmup 1 --drain-between-steps --no-alt --no-latency --no-cpu-baseline --no-roofline > $O/${TAG}_bench_drain_between_steps.json 2> /dev/null
cd /tmp
# HBM traffic, both arithmetics merged into ONE file (kernel names differ: k_eng_fwd8 / k_eng_fwd ...)
bash $R/tools/pmc_traffic.sh 128 8 > $O/${TAG}_hbm_split.json 2> $O/${TAG}_hbm_traffic_pmc.err
NDP_GEMM_MODE=0 NDP_NN_MODE=0 bash $R/tools/pmc_traffic.sh 128 8 > $O/${TAG}_hbm_bitwise.json 2>> $O/${TAG}_hbm_traffic_pmc.err
python - $O/${TAG}_hbm_split.json $O/${TAG}_hbm_bitwise.json > $O/${TAG}_hbm_traffic_pmc.json <<'PY'
import json, sys
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
b.update(a)                       # kernels common to both (nn variants aside: update, loss, load) keep the default arithmetic's numbers
print(json.dumps(b, indent=1))
PY
NDP_GEMM_MODE=23 bash $R/tools/pmc_traffic.sh 128 8 > $O/${TAG}_hbm_traffic_2launch_pmc.json 2>> $O/${TAG}_hbm_traffic_pmc.err
# the 256-pair tick (G = 1: the backward carries the Adam step of the two matrices) with and without that tail (gemm_mode | 1024)
bash $R/tools/pmc_traffic.sh 256 8 > $O/${TAG}_hbm_traffic_256pairs_pmc.json 2>> $O/${TAG}_hbm_traffic_pmc.err
NDP_GEMM_MODE=1031 bash $R/tools/pmc_traffic.sh 256 8 > $O/${TAG}_hbm_traffic_256pairs_mode1031_pmc.json 2>> $O/${TAG}_hbm_traffic_pmc.err
bash $R/tools/pmc_sq.sh 128 12 > /dev/null 2>&1; cp $O/pmc_sq.json $O/${TAG}_sq_counters_pmc.json
NDP_GEMM_MODE=23 bash $R/tools/pmc_sq.sh 128 12 > /dev/null 2>&1; cp $O/pmc_sq.json $O/${TAG}_sq_counters_2launch_pmc.json
NDP_GEMM_MODE=0 NDP_NN_MODE=0 bash $R/tools/pmc_sq.sh 128 12 > /dev/null 2>&1; cp $O/pmc_sq.json $O/${TAG}_sq_counters_bitwise_pmc.json
bash $R/tools/pmc_mix.sh 256 12 > /dev/null 2>&1; cp $O/pmc_mix.json $O/${TAG}_instruction_mix_pmc.json 2> /dev/null
# the microbenchmarks behind the kernels' cost model: what hides in an MFMA gap (coexec2; round 4's coexec for the record), issue cost of
# the vector instructions the level kernels are made of, the shapes of a tile's activation stores, v_fma_mix against the conversion sequence
python $R/tools/experiments/micro/coexec2_gen.py > /tmp/coexec2.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/coexec2 /tmp/coexec2.hip > /dev/null 2>&1 && /tmp/coexec2 > $O/${TAG}_micro_coexec2.txt 2>&1   # (generated: not tracked)
for m in coexec valu_rates store_patterns mixprobe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/$m $R/tools/experiments/micro/$m.hip > /dev/null 2>&1 && /tmp/$m > $O/${TAG}_micro_$m.txt 2>&1
done
python $R/tools/latency_bench.py 5 > $O/${TAG}_latency_persistent_ab.txt 2>&1
python $R/tools/surface_accuracy.py 8 > $O/${TAG}_surface_accuracy_seeds.txt 2>&1
for k in k_eng_fwd8 k_eng_bwd_f k_eng_nn_mx8; do python $R/tools/isa_cost.py $k; done > $O/${TAG}_isa_issue_cost.txt 2>&1
ls -la $O | grep ${TAG}_
