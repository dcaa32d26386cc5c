cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python tools/tick_bench.py 256 24 > gpurun_out/r06/tick0.txt 2>&1
python tools/tick_bench.py 128 24 >> gpurun_out/r06/tick0.txt 2>&1
bash tools/experiments/clock_table.sh 40000 > gpurun_out/r06/clock_table.txt 2>&1
tail -3 gpurun_out/r06/tick0.txt
