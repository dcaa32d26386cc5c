cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python tools/generic_bench.py 128 64 > gpurun_out/r06/generic_bench.txt 2>&1
cat gpurun_out/r06/generic_bench.txt | tail -12
