cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_split_accuracy.py -q 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_hip_parity.py -q -k "bench_geometry or early_stop or engine or pyramid" 2>&1 | tail -3
python tools/tick_bench.py 128 24
python tools/tick_bench.py 128 24
python tools/phase_timing.py 128 16 2>&1 | grep -A12 "^fwd8"
