cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_hip_parity.py tests/test_split_accuracy.py tests/test_registration_gpu.py -q -x 2>&1 | tail -4
for rep in 1 2 3; do for m in 1031 7; do echo "gemm_mode $m: $(NDP_TICK_HASH=1 NDP_GEMM_MODE=$m python tools/tick_bench.py 256 24 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"; done; done > gpurun_out/r06/ab_adam_tail.txt 2>&1
for m in 1031 7; do echo "gemm_mode $m: $(NDP_TICK_HASH=1 NDP_GEMM_MODE=$m python tools/tick_bench.py 128 24 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-230)"; done >> gpurun_out/r06/ab_adam_tail.txt 2>&1
cat gpurun_out/r06/ab_adam_tail.txt
