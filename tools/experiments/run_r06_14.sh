cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "nn or onepass or engine or slot or chamfer" > gpurun_out/r06/t_nn2.txt 2>&1
tail -4 gpurun_out/r06/t_nn2.txt
NDP_TICK_HASH=1 bash tools/experiments/ab.sh "nc0 nc1" 3 256 24 > gpurun_out/r06/ab_nncol.txt 2>&1
cat gpurun_out/r06/ab_nncol.txt | cut -c1-200
echo "== NN new columns" > gpurun_out/r06/pt_nncol.txt
NDP_PT_LIB=tools/experiments/var/PTnc1.so NDP_PT_STAGE=1 python tools/phase_timing.py 256 300 2>&1 | grep -E "per-tick|shader clock|nn_mx|setup|operands|distances|rows:|barrier \(|columns" >> gpurun_out/r06/pt_nncol.txt
cat gpurun_out/r06/pt_nncol.txt
