cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_hip_parity.py -q -x -k "nn or neigh or chamfer or engine_matches or engine_nn or config5 or bench_cloud" 2>&1 | tail -2
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "n0 n1" 3 256 24 > gpurun_out/r06/ab_nn_onewg.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "n0 n1" 2 128 24 >> gpurun_out/r06/ab_nn_onewg.txt 2>&1
cat gpurun_out/r06/ab_nn_onewg.txt | cut -c1-200
