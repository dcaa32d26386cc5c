"""Host memory held per pair by a long register_batch(..., sink=...) stream: tracemalloc top sites + RSS + gc object counts.
    python tools/stream_memory.py [pairs]"""
import gc, os, sys, tracemalloc, resource
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
kw = dict(slots=256, engines=int(os.environ.get("SM_ENGINES", "3")), prefetch=os.environ.get("SM_PREFETCH", "1") == "1", workers=int(os.environ.get("SM_WORKERS", "3")))
dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
base = [tuple(t.to(dev) for t in synthetic_pair(i)[:2]) for i in range(512)]
model = Registration(cfg)
model.register_batch(base, sink=lambda i, w, s: None, **kw)       # warm-up: engines, pinned rings, caches
def rss():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6
gc.collect()
tracemalloc.start(10)
r0, o0 = rss(), len(gc.get_objects())
s0 = tracemalloc.take_snapshot()
model.register_batch(base * (n // 512), sink=lambda i, w, s: None, **kw)
rm = rss()
model.register_batch(base * (n // 512), sink=lambda i, w, s: None, **kw)
print(f"second stream of {n}: RSS {rm:.0f} -> {rss():.0f} MB ({(rss() - rm) * 1e6 / n:.0f} B per pair)  {kw}")
for rep in range(int(os.environ.get("SM_MORE", "0"))):
    rm = rss()
    model.register_batch(base * (n // 512), sink=lambda i, w, s: None, **kw)
    print(f"stream {rep + 3} of {n}: RSS {rm:.0f} -> {rss():.0f} MB ({(rss() - rm) * 1e6 / n:.0f} B per pair); device {torch.cuda.memory_allocated() / 1e6:.0f} MB allocated, {torch.cuda.memory_reserved() / 1e6:.0f} reserved; pinned rings {[ (len(getattr(r, 'bufs', [])) if hasattr(r, 'bufs') else '?') for r in getattr(model, '_pin_rings', {}).values()]}")
gc.collect()
r1, o1 = rss(), len(gc.get_objects())
s1 = tracemalloc.take_snapshot()
print(f"{n} pairs: RSS {r0:.0f} -> {r1:.0f} MB ({(r1 - r0) * 1e6 / n:.0f} B per pair), gc-tracked objects {o0} -> {o1} ({(o1 - o0) / n:.2f} per pair)")
for st in s1.compare_to(s0, "lineno")[:12]:
    print("  ", st)
