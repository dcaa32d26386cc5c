import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair
import deformationpyramid_amd.registration as R
dev = torch.device("cuda:0")
cfg = load_config("config/NDP.yaml", device=0)
NP = 1024
pairs = [tuple(t.to(dev) for t in synthetic_pair(i)[:2]) for i in range(NP)]
model = Registration(cfg)
torch.manual_seed(0)
model.register_batch(pairs, slots=128, chunk=4, engines=3)
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
out = model.register_batch(pairs, slots=128, chunk=4, engines=3)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("call %.3fs, sync after %.3fs" % (t1 - t0, t2 - t1))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
