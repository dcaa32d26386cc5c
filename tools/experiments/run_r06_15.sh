cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s2 s3 s4" 3 256 24 > gpurun_out/r06/ab_nn_prologue.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s2 s3 s4" 2 128 24 >> gpurun_out/r06/ab_nn_prologue.txt 2>&1
cat gpurun_out/r06/ab_nn_prologue.txt | cut -c1-220
