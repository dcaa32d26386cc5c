cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "tab0 tab1" 3 256 24 > gpurun_out/r06/ab_tab.txt 2>&1
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "tab0 tab1" 1 128 24 >> gpurun_out/r06/ab_tab.txt 2>&1
cat gpurun_out/r06/ab_tab.txt | cut -c1-200
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all4.txt 2>&1
tail -4 gpurun_out/r06/t_all4.txt
