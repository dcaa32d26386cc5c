R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=r06f
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_style.json 2> /dev/null
python bench.py > $O/${TAG}_bench_line.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_tick -o tick -- python $R/tools/tick_bench.py 128 24 > $O/${TAG}_prof_tick.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_tick256 -o tick -- python $R/tools/tick_bench.py 256 24 > $O/${TAG}_prof_tick256.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_tick256/*.db | head -1) $O/${TAG}_tick_kernel_stats_256pairs.csv > /dev/null
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_tick/*.db | head -1) $O/${TAG}_tick_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-alt --no-roofline --no-latency --no-cpu-baseline > $O/${TAG}_prof_bench.log 2>&1
python $R/tools/rocprof_summary.py $(ls $O/${TAG}_prof_bench/*.db | head -1) $O/${TAG}_bench_kernel_stats.csv > /dev/null
bash $R/tools/pmc_mix.sh 256 12 > /dev/null 2>&1; cp $O/pmc_mix.json $O/${TAG}_instruction_mix_pmc.json 2> /dev/null
bash $R/tools/pmc_sq.sh 128 12 > /dev/null 2>&1; cp $O/pmc_sq.json $O/${TAG}_sq_counters_pmc.json
cd $R
python -c "
import json
for f in ('driver_style','line'):
    d=json.load(open('gpurun_out/r06f_bench_%s.json'%f)); print(f, round(d['value'],1), d['kernels_ms_per_tick'], round(d['roofline']['frac'],4), d['latency']['ms_per_pair'])"
head -8 $O/${TAG}_tick_kernel_stats_256pairs.csv
