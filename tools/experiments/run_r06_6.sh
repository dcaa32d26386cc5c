cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for v in old new; do
  echo "== NN $v"; NDP_PT_LIB=tools/experiments/var/PTnn_$v.so NDP_PT_STAGE=1 python tools/phase_timing.py 256 300 2>&1 | grep -E "per-tick|shader clock|nn_mx|setup|operands|distances|rows:|barrier|columns"
done > gpurun_out/r06/pt_nn.txt 2>&1
cat gpurun_out/r06/pt_nn.txt
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_registration_gpu.py -x -q -m gpu -k "pyramid or warp or register or sink or final" > gpurun_out/r06/t_warp.txt 2>&1
tail -5 gpurun_out/r06/t_warp.txt
python bench.py --steps 4 --warmup 1 --no-alt --no-latency --no-cpu-baseline > gpurun_out/r06/bench_quick1.json 2> gpurun_out/r06/bench_quick1.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_quick1.json')); print(d['value'], d['ms_per_step'], d.get('roofline'), d.get('tick'))"
