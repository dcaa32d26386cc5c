cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "sm0 sm1" 3 256 24 > gpurun_out/r06/ab_signmask.txt 2>&1
cat gpurun_out/r06/ab_signmask.txt | cut -c1-200
