"""Throughput of the generic level kernels (csrc/ndp_generic.inc) next to the MFMA kernels of the shipped 128 / 3:
whole register_batch() jobs of synthetic 8192-pt pairs (NDP.yaml otherwise) at several width / depth, pairs/s and ms per tick.
    python tools/generic_bench.py [pairs] [slots] [WxD ...]       (e.g. 128x2 256x4; default: a table of eight shapes)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import Config, load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
base = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
pairs = [tuple(t.to(dev) for t in synthetic_pair(i)[:2]) for i in range(n_pairs)]
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[3:]] or [(128, 3), (64, 2), (64, 3), (128, 2), (128, 4), (256, 3), (256, 4), (32, 1)]
for width, depth in shapes:
    cfg = Config(base, width=width, depth=depth)
    model = Registration(cfg)
    torch.manual_seed(0)
    model.register_batch(pairs[:slots], slots=slots)          # warm-up: allocations, first launches
    torch.cuda.synchronize()
    torch.manual_seed(0)
    t0 = time.perf_counter()
    outs = model.register_batch(pairs, slots=slots)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    evals = sum(sum(ic.values()) for _, ic in outs)
    print(f"width {width:3d} depth {depth}: {n_pairs / dt:8.1f} pairs/s, {evals / n_pairs:6.1f} loss evaluations per pair, "
          f"{1e3 * dt * slots / evals:7.3f} ms per tick of {slots} pairs ({'MFMA kernels' if (width, depth) == (128, 3) else 'generic fp32 kernels'})", flush=True)
