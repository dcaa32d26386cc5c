cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_hip_parity.py tests/test_split_accuracy.py -q -x 2>&1 | tail -3
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s6 s7" 3 1 96 > gpurun_out/r06/ab_b1_chains.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s6 s7" 3 256 24 >> gpurun_out/r06/ab_b1_chains.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s6 s7" 2 128 24 >> gpurun_out/r06/ab_b1_chains.txt 2>&1
cat gpurun_out/r06/ab_b1_chains.txt | cut -c1-220
python tools/latency_bench.py 3 2>&1 | tail -4
