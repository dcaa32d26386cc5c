cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
(cd .ab_prev && python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1)
for rep in 1 2; do
  for v in prev cur; do
    d=$GRAFT_REPO_ROOT; [ $v = prev ] && d=$GRAFT_REPO_ROOT/.ab_prev
    (cd $d && python bench.py --steps 6 --warmup 1 --no-alt --no-latency --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms/step')")
  done
done > gpurun_out/r06/bench_ab_host.txt 2>&1
cat gpurun_out/r06/bench_ab_host.txt
