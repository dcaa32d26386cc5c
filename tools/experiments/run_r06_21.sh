cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_hip_parity.py tests/test_registration_gpu.py -q -x 2>&1 | tail -3
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s8 s10" 3 1 96 > gpurun_out/r06/ab_nnlat16.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s8 s10" 2 2 96 >> gpurun_out/r06/ab_nnlat16.txt 2>&1
cat gpurun_out/r06/ab_nnlat16.txt | cut -c1-220
python tools/latency_bench.py 3 2>&1 | tail -4
