"""Fixed small workload for profiling: B synthetic pairs, N ticks at level 0 (all slots active).
    python tools/tick_bench.py [B] [ticks]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 24
torch.set_num_threads(8)
torch.manual_seed(0)                      # (the pyramid initialisation and the sampling permutations: NDP_TICK_HASH digests are comparable between runs)
dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _modes import from_env
model = Registration(cfg, **from_env())
preps = []
for i in range(B):
    s, t, _, _ = synthetic_pair(i)
    preps.append(model._prepare(s.to(dev), t.to(dev), None))
eng = model._engine(B, preps[0], n_hint=int(os.environ.get('NDP_TICK_NHINT', '0')))      # (NDP_TICK_NHINT: a larger point capacity -> other strides between the pairs' buffers)
for b, p in enumerate(preps):
    eng.load_jobs([p.load_job(b)])
eng.run_ticks(4)
torch.cuda.synchronize()
ms = eng.run_ticks_timed(ticks)
if os.environ.get("NDP_TICK_HASH"):        # a digest of the state the ticks left: variants that claim bitwise equality can be compared
    import hashlib
    torch.cuda.synchronize()
    h = hashlib.sha1()
    for tname in ("params", "pts", "adam_m", "adam_v", "heads"):
        h.update(getattr(eng, tname).cpu().numpy().tobytes())
    print("state digest", h.hexdigest()[:16])
print("B", B, "G", eng.G, "per-tick ms [fwd nn loss bwd2 bwd1 upd]:", [round(x / ticks, 4) for x in ms], "sum", round(sum(ms) / ticks, 4))
