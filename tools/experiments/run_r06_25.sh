cd $GRAFT_REPO_ROOT
for c in B C D E; do
  python bench.py --config $c --steps 2 --warmup 1 > gpurun_out/r06f_bench_config_$c.json 2> /dev/null
done
python bench.py --slots 64 --engines 1 --pairs-per-step 2048 --steps 2 --warmup 1 --no-alt --no-latency --no-cpu-baseline > gpurun_out/r06f_bench_batch64.json 2> /dev/null
python -c "
import json
for c in 'BCDE':
    e=json.load(open('gpurun_out/r06f_bench_config_%s.json'%c)); print(c, round(e['value'],1), e['roofline']['kernel'], round(e['roofline']['frac'],3), round(e['tick']['ms'],4))
e=json.load(open('gpurun_out/r06f_bench_batch64.json')); print('batch64', round(e['value'],1))"
