cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "B0 C0" 2 256 24 > gpurun_out/r06/ab_C.txt 2>&1
cat gpurun_out/r06/ab_C.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all.txt 2>&1
tail -8 gpurun_out/r06/t_all.txt
