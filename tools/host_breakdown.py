"""Where does the host time go in register_batch?  Wraps the engine/registration methods with timers.
   python tools/host_breakdown.py [slots] [pairs]"""
import os, sys, time, collections, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd import registration as R, engine as E
from deformationpyramid_amd.synthetic import synthetic_pair

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 128
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.set_num_threads(8)
dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
pairs = []
for i in range(npairs):
    s, t, _, _ = synthetic_pair(i)
    pairs.append((s.to(dev), t.to(dev)))
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            key = label + ("[producer]" if threading.current_thread() is not threading.main_thread() else "")
            acc[key] += time.perf_counter() - t0; cnt[key] += 1
    setattr(obj, name, g)
for n in ("load_jobs", "run_ticks", "snapshot_async", "wait_snapshot"):
    wrap(E.BatchedEngine, n, "eng." + n)
wrap(R.Registration, "_prepare", "prepare")
wrap(R.Registration, "_finish", "finish")
model = R.Registration(cfg)
model.register_batch(pairs[:slots], slots=slots)
acc.clear(); cnt.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
model.register_batch(pairs, slots=slots)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{npairs} pairs, {slots} slots: {dt:.3f} s  -> {npairs/dt:.1f} pairs/s")
for k in sorted(acc, key=acc.get, reverse=True):
    print(f"  {k:28s} total {acc[k]*1e3:8.1f} ms  calls {cnt[k]:6d}  per call {acc[k]/cnt[k]*1e3:7.3f} ms")
