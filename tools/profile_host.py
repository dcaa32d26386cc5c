"""Host-side profile of one register_batch step (cProfile).  Dev tool."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.set_num_threads(thr)
dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
pairs = []
for i in range(B):
    s, t, _, _ = synthetic_pair(i)
    pairs.append((s.to(dev), t.to(dev)))
model = Registration(cfg)
model.register_batch(pairs, slots=B)
torch.cuda.synchronize()
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
model.register_batch(pairs, slots=B)
torch.cuda.synchronize()
pr.disable()
print("step s:", time.time() - t0, "B", B, "threads", thr)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
