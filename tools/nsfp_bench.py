"""NSFP baseline timing on the HIP path next to the CPU oracle:  python tools/nsfp_bench.py [pairs]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair
from deformationpyramid_amd import nsfp
from oracle import ndp_oracle as O

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = load_config(os.path.join(ROOT, "config", "baselines", "NSFP.yaml"), device=0)
model = Registration(cfg)
torch.manual_seed(0)
tot_it, tot_t = 0, 0.0
for p in range(n_pairs + 1):
    src, tgt, _, _ = synthetic_pair(p)
    model.load_pcds(src, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    warped, _ = model.register()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if p:                                                   # pair 0 = warm-up
        tot_it += model.last_nsfp["iters"]; tot_t += dt
    print(f"pair {p}: {model.last_nsfp['iters']} Adam iterations, loss {model.last_nsfp['loss']:.4f}, {dt:.3f} s")
print(f"HIP: {1e3 * tot_t / tot_it:.3f} ms per iteration, {n_pairs / tot_t:.3f} pairs/s  (S=T=2000, 8192-pt final flow)")
# CPU oracle: 20 iterations of the same loop
src, tgt, _, _ = synthetic_pair(1)
torch.manual_seed(0)
m = nsfp.Neural_Prior()
s = (src - src.mean(0))[torch.randperm(src.shape[0])[:2000]].numpy()
t = (tgt - tgt.mean(0))[torch.randperm(tgt.shape[0])[:2000]].numpy()
t0 = time.perf_counter()
O.nsfp_optimize(m.flat.numpy(), s, t, iters=10, early_stop=False, nthreads=32)
print(f"CPU oracle (forward/NN on 32 threads, backward scalar): {1e3 * (time.perf_counter() - t0) / 10:.1f} ms per iteration")
