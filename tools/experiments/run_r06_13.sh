cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s0 s1" 3 256 24 > gpurun_out/r06/ab_stage.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s0 s1" 2 128 24 >> gpurun_out/r06/ab_stage.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s0 s1" 2 1 96 >> gpurun_out/r06/ab_stage.txt 2>&1
cat gpurun_out/r06/ab_stage.txt | cut -c1-220
