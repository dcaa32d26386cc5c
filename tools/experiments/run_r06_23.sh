cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_hip_parity.py tests/test_registration_gpu.py -q -x 2>&1 | tail -3
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "w0 w1" 3 256 24 > gpurun_out/r06/ab_warp_tail.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "w0 w1" 2 128 24 >> gpurun_out/r06/ab_warp_tail.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "w0 w1" 2 64 24 >> gpurun_out/r06/ab_warp_tail.txt 2>&1
cat gpurun_out/r06/ab_warp_tail.txt | cut -c1-200
