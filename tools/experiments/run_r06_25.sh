cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "x0 x1" 3 256 24 > gpurun_out/r06/ab_warp_tail2.txt 2>&1
cat gpurun_out/r06/ab_warp_tail2.txt | cut -c1-200
