cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python tools/phase_timing.py 256 16 > gpurun_out/r06/pt_loss_256.txt 2>&1
python tools/phase_timing.py 1 64 > gpurun_out/r06/pt_loss_1.txt 2>&1
grep -A8 "loss gradient\|loss decision\|per-tick" gpurun_out/r06/pt_loss_256.txt
grep -A8 "loss gradient\|loss decision\|per-tick" gpurun_out/r06/pt_loss_1.txt
