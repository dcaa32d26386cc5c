cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
(cd .ab_prev && python -c "from deformationpyramid_amd import _native as n; n.build(force=True); n.build_host(force=True)" > /dev/null 2>&1)
for rep in 1 2; do
  for v in prev cur; do
    d=$GRAFT_REPO_ROOT; [ $v = prev ] && d=$GRAFT_REPO_ROOT/.ab_prev
    (cd $d && python bench.py --steps 10 --warmup 2 --no-alt --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['tick']['ms']; print('$v', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms/step  tick', round(t,4), 'tick-rate', round(256/(d['loss_evals_per_pair']*t*1e-3)), d['kernels_ms_per_tick'])")
  done
done > gpurun_out/r06/bench_ab_session3.txt 2>&1
cat gpurun_out/r06/bench_ab_session3.txt
