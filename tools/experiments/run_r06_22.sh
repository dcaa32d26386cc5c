cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "sm0 sm2" 3 256 24 > gpurun_out/r06/ab_signmask2.txt 2>&1
cat gpurun_out/r06/ab_signmask2.txt | cut -c1-200
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all6.txt 2>&1
tail -15 gpurun_out/r06/t_all6.txt | cut -c1-200
