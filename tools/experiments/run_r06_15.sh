cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/experiments/sweep3.sh "3:256:4 2:256:4 4:256:4 3:256:8 3:256:2 3:256:4 2:256:4 4:256:4 3:256:8 3:256:2" 8 > gpurun_out/r06/engines_sweep_long.txt 2>&1
cat gpurun_out/r06/engines_sweep_long.txt
