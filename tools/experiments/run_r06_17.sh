cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "tab0 x00 x10 x01 x11" 2 256 24 > gpurun_out/r06/ab_tab2.txt 2>&1
cat gpurun_out/r06/ab_tab2.txt | cut -c1-200
