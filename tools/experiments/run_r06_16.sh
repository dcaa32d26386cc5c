cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_hip_parity.py -q -x 2>&1 | tail -3
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s3 s5" 3 256 24 > gpurun_out/r06/ab_loss_loads.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s3 s5" 2 128 24 >> gpurun_out/r06/ab_loss_loads.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s3 s5" 2 1 96 >> gpurun_out/r06/ab_loss_loads.txt 2>&1
cat gpurun_out/r06/ab_loss_loads.txt | cut -c1-220
