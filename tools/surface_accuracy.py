"""The 8 partial-overlap SURFACE pairs of the bench line's `accuracy_surface_pairs` under several seeds of the pyramid init / sampling
permutations, both arithmetics: how much of a difference between two builds is trajectory noise.
    python tools/surface_accuracy.py [seeds]        (reference, tests/golden/F10b: full-EPE 6.12, AccS 35.6 %, AccR 62.6 %)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.loss import compute_flow_metrics
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import surface_pair

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
sp = [surface_pair(p) for p in range(8)]
for mode, nn in ((7, 2), (0, 0)):
    model = Registration(cfg, gemm_mode=mode, nn_mode=nn)
    rows = []
    for seed in range(n_seeds):
        torch.manual_seed(seed)
        res = model.register_batch([(a.to(dev), b.to(dev)) for a, b, _, _ in sp], slots=8, engines=1)
        acc = None
        for (w, _), (a, _, fg, ov) in zip(res, sp):
            m = compute_flow_metrics(w - a.to(dev), fg.to(dev), ov.to(dev))
            v = np.array([m["full-epe"], m["full-AccS"], m["full-AccR"]], dtype=np.float64)
            acc = v if acc is None else acc + v
        rows.append(acc / 8)
        print(f"gemm_mode {mode} seed {seed}: full-EPE {rows[-1][0]:.2f}  AccS {rows[-1][1]:.1f} %  AccR {rows[-1][2]:.1f} %")
    r = np.array(rows)
    print(f"gemm_mode {mode}: mean over {n_seeds} seeds  EPE {r[:, 0].mean():.2f} (min {r[:, 0].min():.2f}, max {r[:, 0].max():.2f})  "
          f"AccS {r[:, 1].mean():.1f} ({r[:, 1].min():.1f} .. {r[:, 1].max():.1f})  AccR {r[:, 2].mean():.1f} ({r[:, 2].min():.1f} .. {r[:, 2].max():.1f})")
