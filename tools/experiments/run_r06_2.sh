cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_split_accuracy.py -x -q -m gpu > gpurun_out/r06/t_split.txt 2>&1
tail -5 gpurun_out/r06/t_split.txt
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "r05 A" 2 256 24 > gpurun_out/r06/ab_A.txt 2>&1
cat gpurun_out/r06/ab_A.txt
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu > gpurun_out/r06/t_parity.txt 2>&1
tail -5 gpurun_out/r06/t_parity.txt
