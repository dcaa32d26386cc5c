cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all2.txt 2>&1
tail -6 gpurun_out/r06/t_all2.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/experiments/clock_table.sh 40000 > gpurun_out/r06/clock_table2.txt 2>&1
grep -E "^===|per-tick|shader clock|cycles per tile|bench|determinism|rror" gpurun_out/r06/clock_table2.txt | head -40
