cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "nn or onepass or engine or slot" > gpurun_out/r06/t_nn.txt 2>&1
tail -5 gpurun_out/r06/t_nn.txt
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "C0 N1" 2 256 24 > gpurun_out/r06/ab_N.txt 2>&1
cat gpurun_out/r06/ab_N.txt
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "C0 N1" 1 128 24 >> gpurun_out/r06/ab_N.txt 2>&1
tail -2 gpurun_out/r06/ab_N.txt
