cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "A B0 B1" 2 256 24 > gpurun_out/r06/ab_B.txt 2>&1
cat gpurun_out/r06/ab_B.txt
timeout 900 python -m pytest tests/test_split_accuracy.py -x -q -m gpu > gpurun_out/r06/t_split.txt 2>&1
tail -5 gpurun_out/r06/t_split.txt
NDP_PT_STAGE=3 bash tools/experiments/pt_ab.sh "PT0 PT1" 256 200 > gpurun_out/r06/pt_B.txt 2>&1
cat gpurun_out/r06/pt_B.txt
