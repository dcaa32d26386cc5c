cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s8 s9" 3 1 96 > gpurun_out/r06/ab_nt.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "s8 s9" 2 256 24 >> gpurun_out/r06/ab_nt.txt 2>&1
cat gpurun_out/r06/ab_nt.txt | cut -c1-220
