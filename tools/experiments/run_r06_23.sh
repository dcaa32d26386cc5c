cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "nn or onepass or engine or slot or chamfer" > gpurun_out/r06/t_nn3.txt 2>&1
tail -3 gpurun_out/r06/t_nn3.txt
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "nr0 nr1" 3 256 24 > gpurun_out/r06/ab_nnrow.txt 2>&1
cat gpurun_out/r06/ab_nnrow.txt | cut -c1-200
