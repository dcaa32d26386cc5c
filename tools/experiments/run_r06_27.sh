cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
{
echo "# loss stage: head record + level input requested up front, pair state read field by field (no scratch copy): l0 = before, l1 = after"
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "l0 l1" 3 256 24
for v in l0 l1 l0 l1; do echo "$v: $(NDP_HIP_LIB=$PWD/tools/experiments/var/$v.so python tools/latency_bench.py 6 2>&1 | grep 'six launches' | tail -1)"; done
} > gpurun_out/r06/loss_hoist_ab.txt 2>&1
cat gpurun_out/r06/loss_hoist_ab.txt
