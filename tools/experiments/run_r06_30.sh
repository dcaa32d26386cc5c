cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all8.txt 2>&1
tail -4 gpurun_out/r06/t_all8.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
NDP_TICK_HASH=1 python tools/tick_bench.py 256 24 2>&1 | tail -1
