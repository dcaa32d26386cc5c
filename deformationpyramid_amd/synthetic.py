"""Seeded synthetic point-cloud pairs (the 14 GB 4DMatch download is not
available offline).  Pair p: 16384 points u ~ U[-0.5,0.5]^3 split even/odd into source and target
base clouds (no exact correspondences), target deformed by phi(q) = q + 0.05 sin(3q), rotated by
Rz(0.3 rad), translated by (0.1, 0, -0.05); partial overlap keeps target points with x < 0.25."""
import math

import numpy as np
import torch


def synthetic_pair(p, n_total=16384, partial=True):
    g = torch.Generator().manual_seed(1000 + p)
    u = torch.rand(n_total, 3, generator=g, dtype=torch.float32) - 0.5
    src, tgt_base = u[0::2].contiguous(), u[1::2].contiguous()
    c, s = float(math.cos(0.3)), float(math.sin(0.3))
    Rz = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    t = torch.tensor([0.1, 0.0, -0.05], dtype=torch.float32)

    def phi(q):
        return q + 0.05 * torch.sin(3.0 * q)

    tgt = phi(tgt_base) @ Rz.T + t
    flow_gt = phi(src) @ Rz.T + t - src
    if partial:
        tgt = tgt[tgt_base[:, 0] < 0.25].contiguous()
        overlap = src[:, 0] < 0.25
    else:
        overlap = torch.ones(src.shape[0], dtype=torch.bool)
    return src, tgt, flow_gt, overlap


def surface_pair(p, n_total=16384, partial=True):
    """Seeded pair of SURFACE samples (what 4DMatch scans are): a star-shaped bumpy closed surface, sampled twice (source /
    target base: no exact correspondences), target deformed by phi(q) = q + 0.05 sin(2.5 q + a_p), rotated about z by
    0.25 rad and translated; partial overlap keeps the target samples whose base point has x < 0.2.  NDP solves these
    (the volume-filling cubes of synthetic_pair are not what it is built for)."""
    g = torch.Generator().manual_seed(3000 + p)
    d = torch.randn(n_total, 3, generator=g, dtype=torch.float32)
    d = d / d.norm(dim=1, keepdim=True)
    ph = torch.rand(4, generator=g, dtype=torch.float32) * 6.2831853
    r = 0.35 * (1.0 + 0.18 * torch.sin(3.0 * d[:, 0] + ph[0]) * torch.sin(2.0 * d[:, 1] + ph[1])
                + 0.10 * torch.cos(4.0 * d[:, 2] + ph[2]))
    q = d * r[:, None]
    src, tgt_base = q[0::2].contiguous(), q[1::2].contiguous()
    c, s = float(math.cos(0.25)), float(math.sin(0.25))
    Rz = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    t = torch.tensor([0.08, -0.03, 0.05], dtype=torch.float32)

    def phi(x):
        return x + 0.05 * torch.sin(2.5 * x + ph[3])

    tgt = phi(tgt_base) @ Rz.T + t
    flow_gt = phi(src) @ Rz.T + t - src
    if partial:
        tgt = tgt[tgt_base[:, 0] < 0.2].contiguous()
        overlap = src[:, 0] < 0.2
    else:
        overlap = torch.ones(src.shape[0], dtype=torch.bool)
    return src, tgt, flow_gt, overlap


def synthetic_landmarks(p, src, flow_gt, k=500, noise=0.005):
    g = torch.Generator().manual_seed(5000 + p)
    idx = torch.randperm(src.shape[0], generator=g)[:k]
    ls = src[idx].contiguous()
    lt = (ls + flow_gt[idx] + noise * torch.randn(k, 3, generator=g)).contiguous()
    return ls, lt


def synthetic_depth_pair(seed=0, H=120, W=160):
    """Two small 16-bit depth maps (millimetres) of a bumpy surface, the second one deformed and shifted, for the embedded-
    deformation (N-ICP) baseline that starts from depth images (registration.py:38-90); a hole and a depth step exercise the
    validity and max_triangle_distance rules.  -> (depth_src, depth_tgt, K 3x3).  (tests/golden/make_golden.py carries the
    same generator; golden F15 holds its output for seed 0.)"""
    g = np.random.default_rng(seed)
    v, u = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    x, y = (u - W / 2) / W, (v - H / 2) / H
    ph = g.uniform(0, 6.28, 4)
    z0 = 1.0 + 0.10 * np.sin(5 * x + ph[0]) * np.cos(4 * y + ph[1]) + 0.05 * np.cos(9 * x * y + ph[2])
    z1 = z0 + 0.03 * np.sin(6 * x + ph[3]) + 0.02 * y + 0.045     # > 0 everywhere: no pixel keeps its depth (a point that
                                                                      # coincides with its target is upstream's sqrt(0) NaN trap)
    out = []
    for z in (z0, z1):
        z = z.copy()
        z[(x - 0.2) ** 2 + (y + 0.1) ** 2 < 0.01] = 0.0
        z[:, : W // 8] += 0.25
        z[:6, :] = 0.0
        out.append(np.round(z * 1000).astype(np.uint16))
    K = np.array([[150.0, 0, W / 2], [0, 150.0, H / 2], [0, 0, 1]], dtype=np.float32)
    return out[0], out[1], K
