cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/gp -o gen -- python $R/tools/generic_bench.py 64 64 128x2 > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $f $R/gpurun_out/r06/generic_128x2_kernel_stats.csv | cut -c1-100 | head -12
grep width /tmp/gp.log
