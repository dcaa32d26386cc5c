"""Nerfies baseline timing on the HIP path next to the torch-CPU oracle:  python tools/nerfies_bench.py [pairs] [max_iters]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import Config, load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import surface_pair
from deformationpyramid_amd.loss import compute_flow_metrics
from deformationpyramid_amd import nerfies
from oracle import nerfies_ref as R

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = load_config(os.path.join(ROOT, "config", "baselines", "Nerfies.yaml"), device=0)
if len(sys.argv) > 2:
    cfg = Config(cfg, iters=int(sys.argv[2]))
model = Registration(cfg)
torch.manual_seed(0)
tot_it, tot_t = 0, 0.0
for p in range(n_pairs + 1):
    src, tgt, flow_gt, overlap = surface_pair(p)
    model.load_pcds(src, tgt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    warped, _ = model.register()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m = compute_flow_metrics(warped.cpu() - src, flow_gt, overlap)
    if p:                                                   # pair 0 = warm-up
        tot_it += model.last_nerfies["iters"]; tot_t += dt
    print(f"pair {p}: {model.last_nerfies['iters']} Adam iterations, loss {model.last_nerfies['loss']:.4f}, {dt:.3f} s, "
          f"full-epe {m['full-epe']:.2f} AccS {m['full-AccS']:.1f}")
print(f"HIP: {1e3 * tot_t / max(tot_it, 1):.3f} ms per iteration, {n_pairs / tot_t:.3f} pairs/s  (S=T=2000, 8192-pt final warp)")
src, tgt, _, _ = surface_pair(1)
torch.manual_seed(0)
torch.set_num_threads(32)
net = nerfies.Nerfies_Deformation(max_iter=cfg.iters)
s = (src - src.mean(0))[torch.randperm(src.shape[0])[:2000]]
t = (tgt - tgt.mean(0))[torch.randperm(tgt.shape[0])[:2000]]
t0 = time.perf_counter()
R.optimize(net.flat[:R.P_COUNT], s, t, iters=5)
print(f"torch-CPU oracle (32 threads): {1e3 * (time.perf_counter() - t0) / 5:.1f} ms per iteration")
