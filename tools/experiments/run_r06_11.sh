cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
(cd .ab_prev && python -c "from deformationpyramid_amd import _native as n; n.build(force=True); n.build_host(force=True)" > /dev/null 2>&1)
for rep in 1 2; do
  for v in r05 cur; do
    d=$GRAFT_REPO_ROOT; [ $v = r05 ] && d=$GRAFT_REPO_ROOT/.ab_prev
    for c in E B; do
    (cd $d && python bench.py --config $c --steps 3 --warmup 1 --no-alt --no-latency --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v config $c', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms/step')")
    done
  done
done > gpurun_out/r06/bench_ab_r05_EB.txt 2>&1
cat gpurun_out/r06/bench_ab_r05_EB.txt
