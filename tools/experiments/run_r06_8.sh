cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
NDP_PT_STAGE=3 bash tools/experiments/pt_ab.sh "bk_BASE bk_HALFFETCH bk_HALFWREAD bk_NOMFMA3 bk_NOEPI bk_NOS2 bk_ALL bk_BASE" 256 200 > gpurun_out/r06/bwd_knobs.txt 2>&1
cat gpurun_out/r06/bwd_knobs.txt
for mode in 7 71 7 71; do
  echo "gemm_mode $mode: $(NDP_GEMM_MODE=$mode NDP_HIP_LIB=$GRAFT_REPO_ROOT/tools/experiments/var/cur.so python tools/tick_bench.py 256 24 2>&1 | tail -1)"
done > gpurun_out/r06/update_fold_ab.txt 2>&1
cat gpurun_out/r06/update_fold_ab.txt
