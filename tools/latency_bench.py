"""Batch-1 latency: wall time of Registration.register() on single synthetic 8192-pt pairs (the reference's call pattern),
the six launches per tick (default) against the persistent small-batch tick (gemm_mode | 256: a measured variant).
    python tools/latency_bench.py [repeats]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
pairs = [tuple(t.to(dev) for t in synthetic_pair(i)[:2]) for i in range(4)]
for name, mode in (("persistent tick", 7 | 256), ("six launches per tick", 7), ("persistent tick", 7 | 256), ("six launches per tick", 7)):
    model = Registration(cfg, gemm_mode=mode)
    torch.manual_seed(0)
    walls, iters, outs = [], [], []
    for r in range(reps + 1):
        src, tgt = pairs[r % len(pairs)]
        model.load_pcds(src, tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w, _, _ = model.register()
        torch.cuda.synchronize()
        if r:
            walls.append(1e3 * (time.perf_counter() - t0)); iters.append(model.last_state.total_steps); outs.append(float(w.double().sum()))
    med = sorted(range(len(walls)), key=lambda i: walls[i])[len(walls) // 2]
    print(f"{name:24s}: {walls[med]:.2f} ms per pair, {iters[med]} Adam iterations, {1e3 * walls[med] / iters[med]:.1f} us per iteration; all {[round(x, 2) for x in walls]}; checksum {outs[0]:.6f}")
