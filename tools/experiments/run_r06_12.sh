cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "h1req0 h1req1" 3 256 24 > gpurun_out/r06/ab_h1req.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "h1req0 h1req1" 2 128 24 >> gpurun_out/r06/ab_h1req.txt 2>&1
cat gpurun_out/r06/ab_h1req.txt | cut -c1-200
