cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_generic_width.py -q -m gpu > gpurun_out/r06/t_generic.txt 2>&1
tail -30 gpurun_out/r06/t_generic.txt
timeout 1500 python tools/generic_bench.py 128 64 > gpurun_out/r06/generic_bench.txt 2>&1
tail -9 gpurun_out/r06/generic_bench.txt
