#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

This script is build-container tooling: it imports rabbityl/DeformationPyramid from
/root/reference (read-only) and records numeric inputs/outputs only.  Nothing from the
reference's source text is written anywhere; the .npz files hold numbers.  It never
runs on the GPU box (/root/reference does not exist there) and nothing in the product
imports it.

The reference depends on packages this image lacks (pytorch3d, open3d, skimage,
easydict, mayavi).  They are replaced by empty module stubs, except for
pytorch3d.ops.knn.knn_points whose *semantics* (exact brute-force K=1 nearest
neighbour, squared L2, differentiable distances) are supplied by a small torch
stand-in below -- pytorch3d itself is un-vendored and un-pinned upstream
(README.md:15-16), see DESIGN.md "oracle pinning".

Usage:  python tests/golden/make_golden.py [--only F3,F4] [--bench-pairs 8]
"""
import argparse
import collections
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- stubs
def _knn_points(p1, p2, lengths1=None, lengths2=None, K=1, **kw):
    """Exact brute-force 1-NN with squared-L2, lowest index on ties (torch stand-in)."""
    assert K == 1
    with torch.no_grad():
        idx = []
        for b in range(p1.shape[0]):
            a, c = p1[b], p2[b]
            best = torch.empty(a.shape[0], dtype=torch.int64)
            for s in range(0, a.shape[0], 1024):
                d = ((a[s:s + 1024, None, :] - c[None, :, :]) ** 2).sum(-1)
                best[s:s + 1024] = d.argmin(dim=1)
            idx.append(best)
        idx = torch.stack(idx)[..., None]
    nn = torch.gather(p2, 1, idx[..., 0, None].expand(-1, -1, p2.shape[-1]))
    dists = ((p1 - nn) ** 2).sum(-1, keepdim=True)
    return collections.namedtuple("KNN", "dists idx knn")(dists, idx, None)


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Pointclouds:  # only used in isinstance() checks
        pass

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __getattr__ = dict.__getitem__
        __setattr__ = __setitem__

    mod("pytorch3d")
    mod("pytorch3d.ops")
    mod("pytorch3d.ops.knn", knn_points=_knn_points, knn_gather=None)
    mod("pytorch3d.structures")
    mod("pytorch3d.structures.pointclouds", Pointclouds=Pointclouds)
    sk = mod("skimage")
    sk.io = mod("skimage.io")
    mod("open3d")
    mod("easydict", EasyDict=EasyDict)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return EasyDict


# ------------------------------------------------------------------ synthetic pairs
def synthetic_pair(p, n_total=16384, partial=True):
    """SURVEY.md section 8(d) generator.  Mirrors deformationpyramid_amd.synthetic (kept
    separate on purpose: this file must not import the product)."""
    g = torch.Generator().manual_seed(1000 + p)
    u = torch.rand(n_total, 3, generator=g, dtype=torch.float32) - 0.5
    src = u[0::2].contiguous()
    tgt_base = u[1::2].contiguous()
    c, s = float(np.cos(0.3)), float(np.sin(0.3))
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
    """(mirrors deformationpyramid_amd.synthetic.surface_pair; kept separate on purpose)  Seeded pair of SURFACE samples (what 4DMatch scans are): a star-shaped bumpy closed surface, sampled twice (source /
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
    c, s = float(np.cos(0.25)), float(np.sin(0.25))
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


def ndp_config(EasyDict, **over):
    cfg = dict(deformation_model="NDP", device=torch.device("cpu"), iters=500, lr=0.01,
               max_break_count=15, break_threshold_ratio=0.001, w_reg=0.0, samples=2000,
               m=9, k0=-8, depth=3, width=128, motion_type="SE3",
               rotation_format="axis_angle", w_cd=0.0, trunc_cd=0.25)
    cfg.update(over)
    return EasyDict(cfg)


def layer_params(layer):
    """name -> float32 array, in module registration order."""
    return {k: v.detach().numpy().copy() for k, v in layer.named_parameters()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------ fixtures
def F1_init(nets, **_):
    out = {}
    for tag, kw in {
        "se3aa_m9": dict(m=9, rotation_format="axis_angle", motion="SE3"),
        "sim3eu_m9": dict(m=9, rotation_format="euler", motion="Sim3"),
        "se3aa_m10": dict(m=10, rotation_format="axis_angle", motion="SE3"),
        "se3quat_nr_m3": dict(m=3, rotation_format="quaternion", motion="SE3", nonrigidity_est=True),
        "sflow6d_m2": dict(m=2, rotation_format="6D", motion="sflow"),
    }.items():
        torch.manual_seed(0)
        pyr = nets.Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, **kw)
        names, sums, asums, heads, shapes = [], [], [], [], []
        for li, layer in enumerate(pyr.pyramid):
            for k, v in layer.named_parameters():
                a = v.detach().double().numpy().ravel()
                names.append(f"{li}.{k}")
                sums.append(a.sum())
                asums.append(np.abs(a).sum())
                h = np.zeros(8)
                h[:min(8, a.size)] = a[:8]
                heads.append(h)
                shapes.append(list(v.shape) + [0] * (2 - v.dim()))
        perm_a = torch.randperm(8192)[:16].numpy()
        perm_b = torch.randperm(6100)[:16].numpy()
        out[f"{tag}.names"] = np.array(names)
        out[f"{tag}.sum"] = np.array(sums)
        out[f"{tag}.abssum"] = np.array(asums)
        out[f"{tag}.head8"] = np.array(heads)
        out[f"{tag}.shape"] = np.array(shapes)
        out[f"{tag}.perm8192"] = perm_a
        out[f"{tag}.perm6100"] = perm_b
    save("F1_init", **out)


def F2_layer_forward(nets, **_):
    out = {}
    g = torch.Generator().manual_seed(7)
    x = torch.rand(256, 3, generator=g) - 0.5
    out["x"] = x.numpy()
    variants = {
        "se3aa": dict(rotation_format="axis_angle", motion="SE3"),
        "sim3eu": dict(rotation_format="euler", motion="Sim3"),
        "sflow": dict(rotation_format="axis_angle", motion="sflow"),
        "se3eu": dict(rotation_format="euler", motion="SE3"),
        "sim3aa": dict(rotation_format="axis_angle", motion="Sim3"),
        "se3quat": dict(rotation_format="quaternion", motion="SE3"),
        "se36d": dict(rotation_format="6D", motion="SE3"),
        "sim3quat": dict(rotation_format="quaternion", motion="Sim3"),
    }
    out["head_scale"] = np.float32(30.0)
    out["seed"] = np.int64(11)
    for tag, kw in variants.items():
        torch.manual_seed(11)
        pyr = nets.Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, m=9, **kw)
        for lvl in ((0, 4, 8) if tag in ("se3aa", "sim3eu") else (4,)):
            layer = pyr.pyramid[lvl]
            # make the warp non-trivial: scale head weights so the motion is O(0.1)
            with torch.no_grad():
                for k, v in layer.named_parameters():
                    if "branch" in k or "brach" in k:
                        v.mul_(30.0)
            # weights are NOT stored: tests replay the seeded init (pinned by F1) and apply
            # head_scale; the checksum below catches any drift of that replay.
            out[f"{tag}.L{lvl}.wsum"] = np.float64(sum(v.double().abs().sum().item() for v in layer.parameters()))
            y, data = pyr.warp(x, max_level=lvl, min_level=lvl)
            out[f"{tag}.L{lvl}.out"] = y.detach().numpy()
            # gradient of a fixed linear functional of the output wrt every parameter
            coef = torch.linspace(-1.0, 1.0, 256 * 3).reshape(256, 3)
            for p in layer.parameters():
                p.grad = None
            (y * coef).sum().backward()
            for k, v in layer.named_parameters():
                out[f"{tag}.L{lvl}.grad.{k}"] = v.grad.numpy().copy()
        # full pyramid inference on the same points
        with torch.no_grad():
            yfull, _ = pyr.warp(x)
        out[f"{tag}.full_out"] = yfull.numpy()
    save("F2_layer_forward", **out)


def F3_chamfer(loss_mod, **_):
    out = {}
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(300, 3, generator=g) - 0.5).requires_grad_(True)
    y = (torch.rand(257, 3, generator=g) - 0.5) * 1.1 + 0.02
    for tag, trunc in (("full", 1e9), ("trunc", 0.01)):
        x.grad = None
        L = loss_mod.compute_truncated_chamfer_distance(x[None], y[None], trunc=trunc)
        L.backward()
        nx = _knn_points(x.detach()[None], y[None])
        ny = _knn_points(y[None], x.detach()[None])
        out[f"{tag}.loss"] = np.float32(L.item())
        out[f"{tag}.grad_x"] = x.grad.numpy().copy()
        out[f"{tag}.idx_x"] = nx.idx[0, :, 0].numpy().astype(np.int32)
        out[f"{tag}.idx_y"] = ny.idx[0, :, 0].numpy().astype(np.int32)
        out[f"{tag}.d2_x"] = nx.dists[0, :, 0].numpy()
        out[f"{tag}.d2_y"] = ny.dists[0, :, 0].numpy()
    out["x"] = x.detach().numpy()
    out["y"] = y.numpy()
    save("F3_chamfer", **out)


def _one_level_run(nets, loss_mod, kw, S, T, level, n_iter, seed, landmarks=None, trunc=1e9, keep_step1=False):
    torch.manual_seed(seed)
    pyr = nets.Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, m=9, **kw)
    layer = pyr.pyramid[level]
    g = torch.Generator().manual_seed(seed + 1)
    xs = torch.rand(S, 3, generator=g) - 0.5
    c, s = float(np.cos(0.2)), float(np.sin(0.2))
    Rz = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    if landmarks:
        yt = (xs + 0.04 * torch.sin(4.0 * xs)) @ Rz.T + torch.tensor([0.03, -0.02, 0.01])
    else:
        yt = (torch.rand(T, 3, generator=g) - 0.5) @ Rz.T + torch.tensor([0.03, -0.02, 0.01])
    rec = {"x": xs.numpy(), "y": yt.numpy(), "seed": np.int64(seed),
           "wsum": np.float64(sum(v.double().abs().sum().item() for v in layer.parameters()))}
    pyr.gradient_setup(optimized_level=level)
    opt = torch.optim.Adam(layer.parameters(), lr=0.01)
    losses = []
    for it in range(n_iter):
        w, _ = pyr.warp(xs, max_level=level, min_level=level)
        if landmarks:
            L = torch.mean(torch.sum((w - yt) ** 2, dim=-1))
        else:
            L = loss_mod.compute_truncated_chamfer_distance(w[None], yt[None], trunc=trunc)
        losses.append(L.item())
        opt.zero_grad()
        L.backward()
        if it == 0:
            rec["warp0"] = w.detach().numpy().copy()
            for k, v in layer.named_parameters():
                rec[f"grad0.{k}"] = v.grad.numpy().copy()
        opt.step()
        if it == 2 or (it == 0 and keep_step1):
            for k, v in layer.named_parameters():
                rec[f"step{it + 1}.{k}"] = v.detach().numpy().copy()
    w, _ = pyr.warp(xs, max_level=level, min_level=level)
    rec["warp_final"] = w.detach().numpy()
    rec["losses"] = np.array(losses, dtype=np.float64)
    return rec


def F4_F5_iteration(nets, loss_mod, **_):
    out = {}
    for tag, kw, lvl in (
        ("se3aa.L0", dict(rotation_format="axis_angle", motion="SE3"), 0),
        ("se3aa.L5", dict(rotation_format="axis_angle", motion="SE3"), 5),
        ("sim3eu.L2", dict(rotation_format="euler", motion="Sim3"), 2),
        ("sflow.L3", dict(rotation_format="axis_angle", motion="sflow"), 3),
    ):
        rec = _one_level_run(nets, loss_mod, kw, S=384, T=333, level=lvl, n_iter=20, seed=21,
                             keep_step1=(tag == "se3aa.L0"))
        for k, v in rec.items():
            out[f"{tag}.{k}"] = v
    save("F4F5_iteration", **out)


def F9_landmarks(nets, loss_mod, **_):
    out = {}
    rec = _one_level_run(nets, loss_mod, dict(rotation_format="axis_angle", motion="SE3"),
                         S=200, T=200, level=0, n_iter=8, seed=33, landmarks=True)
    for k, v in rec.items():
        out[f"ldmk.L0.{k}"] = v
    save("F9_landmarks", **out)


def F9c_mixed_landmark_chamfer(nets, loss_mod, **_):
    """Mixed objective of registration.py:189-197: landmarks and samples warped together, loss = landmark MSE +
    w_cd * truncated Chamfer.  Level 0, 8 forced iterations: loss trace, gradients of step 0, parameters after step 3."""
    K, S, T, w_cd, trunc, seed, level = 120, 300, 280, 0.5, 0.012, 41, 0   # trunc in squared units: cuts about a third of the terms
    torch.manual_seed(seed)
    pyr = nets.Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, m=9, rotation_format="axis_angle", motion="SE3")
    layer = pyr.pyramid[level]
    g = torch.Generator().manual_seed(seed + 1)
    c, s = float(np.cos(0.2)), float(np.sin(0.2))
    Rz = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    src_ldmk = torch.rand(K, 3, generator=g) - 0.5
    tgt_ldmk = (src_ldmk + 0.04 * torch.sin(4.0 * src_ldmk)) @ Rz.T + torch.tensor([0.03, -0.02, 0.01])
    s_sample = torch.rand(S, 3, generator=g) - 0.5
    t_sample = ((torch.rand(T, 3, generator=g) - 0.5) + 0.04) @ Rz.T + torch.tensor([0.03, -0.02, 0.01])
    rec = {"src_ldmk": src_ldmk.numpy(), "tgt_ldmk": tgt_ldmk.numpy(), "s_sample": s_sample.numpy(), "t_sample": t_sample.numpy(),
           "seed": np.int64(seed), "w_cd": np.float32(w_cd), "trunc": np.float32(trunc),
           "wsum": np.float64(sum(v.double().abs().sum().item() for v in layer.parameters()))}
    pyr.gradient_setup(optimized_level=level)
    opt = torch.optim.Adam(layer.parameters(), lr=0.01)
    losses, l_ld, l_cd = [], [], []
    for it in range(8):
        src_pts = torch.cat([src_ldmk, s_sample])                                 # registration.py:190-197
        warped_pts, _ = pyr.warp(src_pts, max_level=level, min_level=level)
        warped_ldmk, s_warped = warped_pts[:K], warped_pts[K:]
        loss_ldmk = torch.mean(torch.sum((warped_ldmk - tgt_ldmk) ** 2, dim=-1))
        loss_cd = loss_mod.compute_truncated_chamfer_distance(s_warped[None], t_sample[None], trunc=trunc)
        loss = loss_ldmk + w_cd * loss_cd
        losses.append(loss.item()); l_ld.append(loss_ldmk.item()); l_cd.append(loss_cd.item())
        opt.zero_grad()
        loss.backward()
        if it == 0:
            rec["warp0"] = warped_pts.detach().numpy().copy()
            for k, v in layer.named_parameters():
                rec[f"grad0.{k}"] = v.grad.numpy().copy()
        opt.step()
        if it == 2:
            for k, v in layer.named_parameters():
                rec[f"step3.{k}"] = v.detach().numpy().copy()
    w, _ = pyr.warp(torch.cat([src_ldmk, s_sample]), max_level=level, min_level=level)
    rec["warp_final"] = w.detach().numpy()
    rec["losses"] = np.array(losses, dtype=np.float64)
    rec["losses_ldmk"] = np.array(l_ld, dtype=np.float64)
    rec["losses_cd"] = np.array(l_cd, dtype=np.float64)
    rec["n_truncated0"] = np.int64(-1)
    save("F9c_mixed", **rec)


def _read_ply_numbers(path):
    """Vertices and (fan-triangulated) faces of an ASCII PLY -- this tool's own few lines, not the product's reader."""
    with open(path) as f:
        lines = f.read().split("\n")
    nv = nf = 0
    props, cur, body = [], None, 0
    for i, ln in enumerate(lines):
        tok = ln.split()
        if tok[:1] == ["element"]:
            cur = tok[1]
            if cur == "vertex":
                nv = int(tok[2])
            if cur == "face":
                nf = int(tok[2])
        elif tok[:1] == ["property"] and cur == "vertex":
            props.append(tok[-1])
        elif tok[:1] == ["end_header"]:
            body = i + 1
            break
    cols = [props.index(c) for c in "xyz"]
    v = np.array([[float(x) for x in ln.split()] for ln in lines[body:body + nv]], dtype=np.float64)[:, cols].astype(np.float32)
    faces = []
    for ln in lines[body + nv:body + nv + nf]:
        t = [int(x) for x in ln.split()]
        for j in range(2, t[0]):
            faces.append((t[1], t[j], t[j + 1]))
    return v, np.array(faces, dtype=np.int64)


def F13_shape_transfer(nets, loss_mod, **_):
    """The inline loop of shape_transfer.py:116-157 (Sim3 / euler, samples = 6000, every sample used) on 6000 seeded vertices
    of each demo mesh: first 10 iterations of level 0 (loss trace, gradient checksums of step 0, parameters after step 3),
    then the inference warp of ALL 24 856 source vertices through the nine levels (:160-164).  Also the mesh numbers a PLY
    reader must reproduce (counts, bounding box, total area) -- numbers only, the files stay in /root/reference."""
    sv, sf = _read_ply_numbers(os.path.join(REF, "sim3_demo", "AlienSoldier.ply"))
    tv, tf = _read_ply_numbers(os.path.join(REF, "sim3_demo", "Ortiz.ply"))
    out = {}
    for tag, v, f in (("src", sv, sf), ("tgt", tv, tf)):
        a = v[f[:, 0]].astype(np.float64); b = v[f[:, 1]].astype(np.float64); c = v[f[:, 2]].astype(np.float64)
        out[f"mesh.{tag}.counts"] = np.array([v.shape[0], f.shape[0]])
        out[f"mesh.{tag}.bbox"] = np.stack([v.min(0), v.max(0)])
        out[f"mesh.{tag}.area"] = np.float64(0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum())
        out[f"mesh.{tag}.vsum"] = v.astype(np.float64).sum(0)
        out[f"mesh.{tag}.fsum"] = np.int64(f.sum())
    rng = np.random.default_rng(13)
    src_pcd = torch.from_numpy(sv[rng.choice(sv.shape[0], 6000, replace=False)])
    tgt_pcd = torch.from_numpy(tv[rng.choice(tv.shape[0], 6000, replace=False)])
    torch.manual_seed(0)
    NDP = nets.Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, m=9, nonrigidity_est=False,
                                   rotation_format="euler", motion="Sim3")
    src_mean = src_pcd.mean(dim=0, keepdims=True)
    tgt_mean = tgt_pcd.mean(dim=0, keepdims=True)
    s_sample, t_sample = src_pcd - src_mean, tgt_pcd - tgt_mean
    out["s_sample"], out["t_sample"] = s_sample.numpy(), t_sample.numpy()
    level = 0
    NDP.gradient_setup(optimized_level=level)
    layer = NDP.pyramid[level]
    out["wsum"] = np.float64(sum(v.double().abs().sum().item() for v in layer.parameters()))
    optimizer = torch.optim.Adam(layer.parameters(), lr=0.01)
    losses = []
    for it in range(10):
        s_warped, _ = NDP.warp(s_sample, max_level=level, min_level=level)
        loss = loss_mod.compute_truncated_chamfer_distance(s_warped[None], t_sample[None], trunc=1e+9)
        losses.append(loss.item())
        optimizer.zero_grad()
        loss.backward()
        if it == 0:
            for k, v in layer.named_parameters():
                gr = v.grad.numpy()
                out[f"grad0.{k}"] = gr.copy() if gr.size <= 1024 else gr.reshape(-1)[::37].copy()
                out[f"gsum0.{k}"] = np.float64(gr.astype(np.float64).sum())
                out[f"gabs0.{k}"] = np.float64(np.abs(gr.astype(np.float64)).sum())
        optimizer.step()
        if it == 2:
            for k, v in layer.named_parameters():
                a = v.detach().numpy()
                out[f"step3.{k}"] = a.copy() if a.size <= 1024 else a.reshape(-1)[::37].copy()
        print(f"  iter {it}: loss {losses[-1]:.6f}", flush=True)
    out["losses"] = np.array(losses, dtype=np.float64)
    for k, v in layer.named_parameters():
        out[f"final.{k}"] = v.detach().numpy().copy()              # the trained level 0, so that the warp below can be replayed
    NDP.gradient_setup(optimized_level=-1)
    mesh_vert = torch.from_numpy(sv) - src_mean
    with torch.no_grad():
        warped_vert, _ = NDP.warp(mesh_vert)
    out["mesh_vert"] = mesh_vert.numpy()
    out["warped_vert"] = warped_vert.numpy()
    save("F13_shape_transfer", **out)


def F11_nonrigidity(nets, loss_mod, reg_mod, EasyDict, **_):
    """w_reg > 0: the nonrigidity gate (nets.py:100-103,132-135) and the BCE regulariser (registration.py:216-220)."""
    out = {}
    g = torch.Generator().manual_seed(7)
    x = torch.rand(256, 3, generator=g) - 0.5
    out["x"] = x.numpy()
    out["seed"] = np.int64(13)
    out["head_scale"] = np.float32(30.0)
    for tag, kw in (("se3aa", dict(rotation_format="axis_angle", motion="SE3")),
                    ("sim3quat", dict(rotation_format="quaternion", motion="Sim3")),
                    ("sflow", dict(rotation_format="axis_angle", motion="sflow"))):
        torch.manual_seed(13)
        pyr = nets.Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, m=6, nonrigidity_est=True, **kw)
        lvl = 4
        layer = pyr.pyramid[lvl]
        with torch.no_grad():
            for k, v in layer.named_parameters():
                if "branch" in k or "brach" in k:
                    v.mul_(30.0)
        out[f"{tag}.wsum"] = np.float64(sum(v.double().abs().sum().item() for v in layer.parameters()))
        y, data = pyr.warp(x, max_level=lvl, min_level=lvl)
        nr = data[lvl][1]
        out[f"{tag}.out"] = y.detach().numpy()
        out[f"{tag}.nonrig"] = nr.detach().numpy()
        coef = torch.linspace(-1.0, 1.0, 256 * 3).reshape(256, 3)
        c2 = torch.linspace(0.5, -0.25, 256)
        ((y * coef).sum() + (nr * c2).sum()).backward()
        for k, v in layer.named_parameters():
            out[f"{tag}.grad.{k}"] = v.grad.numpy().copy()
        with torch.no_grad():
            yfull, dfull = pyr.warp(x)
        out[f"{tag}.full_out"] = yfull.numpy()
        out[f"{tag}.level0_has_gate"] = np.bool_(data[lvl][1] is not None and pyr.pyramid[0].nonrigidity_est)
    # iterations with the BCE term, exactly the loop body of registration.py:208-237 at level 2
    torch.manual_seed(17)
    pyr = nets.Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, m=4, nonrigidity_est=True,
                                   rotation_format="axis_angle", motion="SE3")
    lvl, w_reg = 2, 0.5
    g = torch.Generator().manual_seed(18)
    xs = torch.rand(300, 3, generator=g) - 0.5
    yt = (torch.rand(280, 3, generator=g) - 0.5) * 1.05 + 0.02
    out["it.x"], out["it.y"], out["it.seed"], out["it.w_reg"] = xs.numpy(), yt.numpy(), np.int64(17), np.float32(w_reg)
    layer = pyr.pyramid[lvl]
    out["it.wsum"] = np.float64(sum(v.double().abs().sum().item() for v in layer.parameters()))
    pyr.gradient_setup(optimized_level=lvl)
    opt = torch.optim.Adam(layer.parameters(), lr=0.01)
    BCE = torch.nn.BCELoss()
    losses = []
    for it in range(12):
        w, data = pyr.warp(xs, max_level=lvl, min_level=lvl)
        loss = loss_mod.compute_truncated_chamfer_distance(w[None], yt[None], trunc=1e9)
        nonrigidity = data[lvl][1]
        loss = loss + w_reg * BCE(nonrigidity, torch.zeros_like(nonrigidity))
        losses.append(loss.item())
        opt.zero_grad()
        loss.backward()
        if it == 0:
            out["it.warp0"], out["it.nonrig0"] = w.detach().numpy().copy(), nonrigidity.detach().numpy().copy()
            for k, v in layer.named_parameters():
                out[f"it.grad0.{k}"] = v.grad.numpy().copy()
        opt.step()
        if it == 2:
            for k, v in layer.named_parameters():
                out[f"it.step3.{k}"] = v.detach().numpy().copy()
    out["it.losses"] = np.array(losses, dtype=np.float64)
    # end to end with w_reg > 0
    src, tgt, flow_gt, overlap = synthetic_pair(11, n_total=2048)
    cfg = ndp_config(EasyDict, samples=256, w_reg=0.3, m=5)
    warped, trace = _register_traced(reg_mod, EasyDict, cfg, src, tgt, seed=4)
    out["e2e.warped"] = warped.numpy()
    out["e2e.iters_per_level"] = np.array([len(t) for t in trace])
    out["e2e.loss_trace"] = np.array(sum(trace, []), dtype=np.float64)
    save("F11_nonrigidity", **out)


def _register_traced(reg_mod, EasyDict, cfg, src, tgt, landmarks=None, seed=0):
    """Run the reference's register() recording every loss it evaluates, per level."""
    trace = []
    orig_cd = reg_mod.compute_truncated_chamfer_distance
    orig_adam = reg_mod.optim.Adam

    def cd(*a, **k):
        v = orig_cd(*a, **k)
        if not landmarks:
            trace[-1].append(v.item())
        return v

    def adam(*a, **k):
        trace.append([])
        return orig_adam(*a, **k)

    orig_mean = torch.mean

    reg_mod.compute_truncated_chamfer_distance = cd
    reg_mod.optim.Adam = adam
    if landmarks:
        def mean(t, *a, **k):
            v = orig_mean(t, *a, **k)
            if v.dim() == 0 and v.requires_grad:
                trace[-1].append(v.item())
            return v
        reg_mod.torch.mean = mean
    try:
        if seed is not None:                                   # (None: the generator runs on from wherever the caller's sequence left it)
            torch.manual_seed(seed)
        model = reg_mod.Registration(cfg)
        model.load_pcds(src.numpy(), tgt.numpy(), landmarks=landmarks)
        warped, _, _ = model.register()
    finally:
        reg_mod.compute_truncated_chamfer_distance = orig_cd
        reg_mod.optim.Adam = orig_adam
        reg_mod.torch.mean = orig_mean
    return warped.detach(), trace


def F7_end_to_end(reg_mod, loss_mod, EasyDict, **_):
    out = {}
    src, tgt, flow_gt, overlap = synthetic_pair(5, n_total=2048)
    cfg = ndp_config(EasyDict, samples=256)
    warped, trace = _register_traced(reg_mod, EasyDict, cfg, src, tgt, seed=0)
    out["src"], out["tgt"] = src.numpy(), tgt.numpy()
    out["flow_gt"], out["overlap"] = flow_gt.numpy(), overlap.numpy()
    out["warped"] = warped.numpy()
    out["iters_per_level"] = np.array([len(t) for t in trace])
    out["loss_trace"] = np.array(sum(trace, []), dtype=np.float64)
    m = loss_mod.compute_flow_metrics(warped - src, flow_gt, overlap)
    out["metric_keys"] = np.array(list(m.keys()))
    out["metric_vals"] = np.array(list(m.values()), dtype=np.float64)
    save("F7_end_to_end", **out)


def F8_metrics(loss_mod, **_):
    g = torch.Generator().manual_seed(8)
    gt = (torch.rand(500, 3, generator=g) - 0.5) * 0.3
    flow = gt + torch.randn(500, 3, generator=g) * 0.03
    flow[:40] = gt[:40]            # exact hits
    gt[40:50] = 0.0                # zero-length GT (relative error blows up)
    overlap = torch.rand(500, generator=g) < 0.7
    m = loss_mod.compute_flow_metrics(flow, gt, overlap)
    save("F8_metrics", flow=flow.numpy(), flow_gt=gt.numpy(), overlap=overlap.numpy(),
         keys=np.array(list(m.keys())), vals=np.array(list(m.values()), dtype=np.float64))


def F9b_lndp_end_to_end(reg_mod, EasyDict, **_):
    src, tgt, flow_gt, overlap = synthetic_pair(9, n_total=2048)
    g = torch.Generator().manual_seed(99)
    idx = torch.randperm(src.shape[0], generator=g)[:120]
    ls = src[idx]
    lt = ls + flow_gt[idx] + 0.005 * torch.randn(120, 3, generator=g)
    cfg = ndp_config(EasyDict, m=10, samples=256, w_cd=0.0)
    warped, trace = _register_traced(reg_mod, EasyDict, cfg, src, tgt, landmarks=(ls, lt), seed=0)
    save("F9b_lndp_end_to_end", src=src.numpy(), tgt=tgt.numpy(), ldmk_s=ls.numpy(), ldmk_t=lt.numpy(),
         warped=warped.numpy(), iters_per_level=np.array([len(t) for t in trace]),
         loss_trace=np.array(sum(trace, []), dtype=np.float64), flow_gt=flow_gt.numpy())


def F10_benchmark(reg_mod, loss_mod, EasyDict, bench_pairs=8, **_):
    """Reference metric rows on the synthetic benchmark (stand-in for 4DMatch-F, whose 14 GB
    download is not available offline).  8192-pt pairs, NDP.yaml settings, one seed per pair."""
    rows, iters, keys = [], [], None
    for p in range(bench_pairs):
        src, tgt, flow_gt, overlap = synthetic_pair(p)
        cfg = ndp_config(EasyDict)
        warped, trace = _register_traced(reg_mod, EasyDict, cfg, src, tgt, seed=p)
        m = loss_mod.compute_flow_metrics(warped - src, flow_gt, overlap)
        keys = list(m.keys())
        rows.append(list(m.values()))
        iters.append([len(t) for t in trace])
        print(f"pair {p}: iters {iters[-1]}  full-epe {m['full-epe']:.3f} AccS {m['full-AccS']:.2f}", flush=True)
    save("F10_benchmark", keys=np.array(keys), rows=np.array(rows, dtype=np.float64),
         iters=np.array(iters), seeds=np.arange(bench_pairs))


def F10b_surface_benchmark(reg_mod, loss_mod, EasyDict, bench_pairs=8, **_):
    """Reference metric rows on SURFACE pairs NDP actually solves (AccS far above the do-nothing answers, which are
    recorded next to it): the accuracy bar of the GPU path."""
    rows, iters, keys, zero_rows, cent_rows = [], [], None, [], []
    for p in range(bench_pairs):
        src, tgt, flow_gt, overlap = surface_pair(p)
        cfg = ndp_config(EasyDict)
        warped, trace = _register_traced(reg_mod, EasyDict, cfg, src, tgt, seed=p)
        m = loss_mod.compute_flow_metrics(warped - src, flow_gt, overlap)
        keys = list(m.keys())
        rows.append(list(m.values()))
        zero_rows.append(list(loss_mod.compute_flow_metrics(torch.zeros_like(src), flow_gt, overlap).values()))
        cent = (tgt.mean(0) - src.mean(0))[None].expand_as(src)
        cent_rows.append(list(loss_mod.compute_flow_metrics(cent, flow_gt, overlap).values()))
        iters.append([len(t) for t in trace])
        print(f"pair {p}: iters {sum(iters[-1])}  full-epe {m['full-epe']:.3f} AccS {m['full-AccS']:.2f} AccR {m['full-AccR']:.2f}"
              f"   zero-flow epe {zero_rows[-1][0]:.3f}  centroid epe {cent_rows[-1][0]:.3f}", flush=True)
    first = surface_pair(0)
    save("F10b_surface_benchmark", keys=np.array(keys), rows=np.array(rows, dtype=np.float64), iters=np.array(iters),
         seeds=np.arange(bench_pairs), zero_flow_rows=np.array(zero_rows, dtype=np.float64),
         centroid_rows=np.array(cent_rows, dtype=np.float64),
         gen_src_head=first[0][:16].numpy(), gen_tgt_head=first[1][:16].numpy(), gen_flow_head=first[2][:16].numpy(),
         gen_counts=np.array([first[0].shape[0], first[1].shape[0], int(first[3].sum())]))


def F10c_surface_benchmark_seeds(reg_mod, loss_mod, EasyDict, bench_pairs=8, **_):
    """The reference's own SEED-TO-SEED distribution on the F10b surface pairs: eval_nolearned.py:22 seeds the process ONCE and then
    registers pair after pair, so one "run" here is torch.manual_seed(s) followed by the eight pairs in order, s = 0..7 -- 64 metric
    rows.  (F10b holds one draw per pair, seeded per pair: whether a +6 % EPE of another arithmetic is noise or bias cannot be told
    from a single draw.)  The GPU path's runs are seeded the same way (tests/test_registration_gpu.py, bench.py)."""
    n_seeds = 8
    all_rows, all_iters, keys = [], [], None
    for s in range(n_seeds):
        torch.manual_seed(s)
        rs, its = [], []
        for p in range(bench_pairs):
            src, tgt, flow_gt, overlap = surface_pair(p)
            cfg = ndp_config(EasyDict)
            warped, trace = _register_traced(reg_mod, EasyDict, cfg, src, tgt, seed=None)
            m = loss_mod.compute_flow_metrics(warped - src, flow_gt, overlap)
            keys = list(m.keys())
            rs.append(list(m.values()))
            its.append(sum(len(t) for t in trace))
        all_rows.append(rs); all_iters.append(its)
        a = np.array(rs)
        print(f"seed {s}: full-epe {a[:, keys.index('full-epe')].mean():.3f} AccS {a[:, keys.index('full-AccS')].mean():.2f} "
              f"AccR {a[:, keys.index('full-AccR')].mean():.2f}  iters {np.mean(its):.1f}", flush=True)
    save("F10c_surface_benchmark_seeds", keys=np.array(keys), rows=np.array(all_rows, dtype=np.float64), iters=np.array(all_iters),
         seeds=np.arange(n_seeds), pairs=np.arange(bench_pairs))


def F12_nsfp(nets, loss_mod, reg_mod, EasyDict, **_):
    """NSFP baseline: Neural_Prior init / forward / parameter gradients through the Chamfer loss,
    and optimize_neural_SFlow end to end with every evaluated loss recorded."""
    out = {}
    torch.manual_seed(21)
    model = nets.Neural_Prior()
    names = [k for k, _ in model.named_parameters()]
    out["names"] = np.array(names)
    for k, v in model.named_parameters():
        a = v.detach().numpy()
        out[f"init.{k}.sum"] = np.float64(a.astype(np.float64).sum())
        out[f"init.{k}.abs"] = np.float64(np.abs(a.astype(np.float64)).sum())
        out[f"init.{k}.head"] = a.reshape(-1)[:8].copy()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(200, 3, generator=g) - 0.5
    y = (torch.rand(180, 3, generator=g) - 0.5) * 1.1 + 0.03
    # scale the weights up so that the flow is not vanishingly small next to x
    with torch.no_grad():
        for k, v in model.named_parameters():
            if k.endswith("weight"):
                v.mul_(1.5)
    out["fb.x"], out["fb.y"] = x.numpy(), y.numpy()
    flow = model(x)
    out["fb.flow"] = flow.detach().numpy()
    loss = loss_mod.compute_truncated_chamfer_distance((x + flow)[None], y[None], trunc=1e9)
    loss.backward()
    out["fb.loss"] = np.float64(loss.item())
    for k, v in model.named_parameters():
        out[f"fb.grad.{k}"] = v.grad.numpy().copy() if v.numel() <= 4096 else v.grad.numpy().reshape(-1)[::37].copy()
        out[f"fb.gsum.{k}"] = np.float64(v.grad.numpy().astype(np.float64).sum())
        out[f"fb.gabs.{k}"] = np.float64(np.abs(v.grad.numpy().astype(np.float64)).sum())
    out["fb.scale"] = np.float32(1.5)
    # end to end
    src, tgt, flow_gt, overlap = synthetic_pair(7, n_total=2048)
    cfg = EasyDict(dict(deformation_model="NSFP", device=torch.device("cpu"), iters=60, lr=0.01, max_break_count=70,
                        break_threshold_ratio=0.001, samples=256))
    trace = []
    orig_cd = reg_mod.compute_truncated_chamfer_distance

    def cd(*a, **k):
        v = orig_cd(*a, **k)
        trace.append(v.item())
        return v

    reg_mod.compute_truncated_chamfer_distance = cd
    try:
        torch.manual_seed(9)
        m = reg_mod.Registration(cfg)
        m.load_pcds(src.numpy(), tgt.numpy())
        warped, smpl = m.register()
    finally:
        reg_mod.compute_truncated_chamfer_distance = orig_cd
    assert smpl is None
    out["e2e.src"], out["e2e.tgt"] = src.numpy(), tgt.numpy()
    out["e2e.flow_gt"], out["e2e.overlap"] = flow_gt.numpy(), overlap.numpy()
    out["e2e.warped"] = warped.detach().numpy()
    out["e2e.loss_trace"] = np.array(trace, dtype=np.float64)
    out["e2e.seed"] = np.int64(9)
    mt = loss_mod.compute_flow_metrics(warped.detach() - src, flow_gt, overlap)
    out["e2e.metric_keys"] = np.array(list(mt.keys()))
    out["e2e.metric_vals"] = np.array(list(mt.values()), dtype=np.float64)
    save("F12_nsfp", **out)


def F14_nerfies(nets, loss_mod, reg_mod, EasyDict, **_):
    """Nerfies baseline (second half): Nerfies_Deformation init / windowed posenc / SE(3) exp warp /
    per-point Jacobian / log-singular-value regulariser / parameter gradients of cd + 0.001 reg at two annealing stages,
    and optimize_Nerfies end to end on a small pair with every evaluated loss recorded."""
    out = {}
    torch.manual_seed(23)
    net = nets.Nerfies_Deformation(max_iter=5000)
    names = [k for k, _ in net.named_parameters()]
    out["names"] = np.array(names)
    for k, v in net.named_parameters():
        a = v.detach().numpy()
        out[f"init.{k}.sum"] = np.float64(a.astype(np.float64).sum())
        out[f"init.{k}.abs"] = np.float64(np.abs(a.astype(np.float64)).sum())
        out[f"init.{k}.head"] = a.reshape(-1)[:8].copy()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(200, 3, generator=g) - 0.5
    y = (torch.rand(180, 3, generator=g) - 0.5) * 1.1 + 0.03
    out["fb.x"], out["fb.y"] = x.numpy(), y.numpy()
    for it in (0, 700, 2999):
        for v in net.parameters():
            v.grad = None
        warped, J = net(x, iter=it)
        reg = loss_mod.nerfies_regularization(J)
        cd = loss_mod.compute_truncated_chamfer_distance(warped[None], y[None], trunc=1e+9)
        loss = cd + 0.001 * reg
        loss.backward()
        out[f"fb{it}.pe"] = net.posenc(x, it).detach().numpy()
        out[f"fb{it}.warped"] = warped.detach().numpy()
        out[f"fb{it}.J"] = J.detach().numpy()
        out[f"fb{it}.reg"] = np.float64(reg.item())
        out[f"fb{it}.cd"] = np.float64(cd.item())
        out[f"fb{it}.loss"] = np.float64(loss.item())
        for k, v in net.named_parameters():
            gr = v.grad.numpy()
            out[f"fb{it}.grad.{k}"] = gr.copy() if gr.size <= 4992 else gr.reshape(-1)[::37].copy()
            out[f"fb{it}.gsum.{k}"] = np.float64(gr.astype(np.float64).sum())
            out[f"fb{it}.gabs.{k}"] = np.float64(np.abs(gr.astype(np.float64)).sum())
        # upstream builds J with torch.autograd.functional.jacobian(create_graph=False): it is a constant for autograd, so
        # the regulariser moves the loss VALUE (and the stop rule) but contributes no parameter gradient
        out[f"fb{it}.J_requires_grad"] = np.int64(int(J.requires_grad))
        print(f"  iter {it}: cd {cd.item():.6f} reg {reg.item():.6e}", flush=True)
    # end to end
    src, tgt, flow_gt, overlap = synthetic_pair(9, n_total=2048)
    cfg = EasyDict(dict(deformation_model="Nerfies", device=torch.device("cpu"), iters=40, lr=0.01, max_break_count=70,
                        break_threshold_ratio=0.001, samples=256))
    trace = []
    orig_cd, orig_reg = reg_mod.compute_truncated_chamfer_distance, reg_mod.nerfies_regularization

    def cd_hook(*a, **k):
        v = orig_cd(*a, **k)
        trace.append([v.item(), None])
        return v

    def reg_hook(*a, **k):
        v = orig_reg(*a, **k)
        trace_reg.append(v.item())
        return v

    trace_reg = []
    reg_mod.compute_truncated_chamfer_distance = cd_hook
    reg_mod.nerfies_regularization = reg_hook
    try:
        torch.manual_seed(3)
        model = reg_mod.Registration(cfg)
        model.load_pcds(src.numpy(), tgt.numpy())
        warped, _ = model.register()
    finally:
        reg_mod.compute_truncated_chamfer_distance = orig_cd
        reg_mod.nerfies_regularization = orig_reg
    out["e2e.src"], out["e2e.tgt"] = src.numpy(), tgt.numpy()
    out["e2e.warped"] = warped.detach().numpy()
    out["e2e.cd_trace"] = np.array([t[0] for t in trace], dtype=np.float64)
    out["e2e.reg_trace"] = np.array(trace_reg, dtype=np.float64)
    out["e2e.seed"] = np.int64(3)
    save("F14_nerfies", **out)


def synthetic_depth_pair(seed=0, H=120, W=160):
    """Two small 16-bit depth maps (millimetres) of a bumpy surface, the second one deformed and shifted; a hole and a depth
    step exercise the validity and max_triangle_distance rules.  Intrinsics for a 160x120 pinhole camera."""
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
        z[(x - 0.2) ** 2 + (y + 0.1) ** 2 < 0.01] = 0.0                  # a hole
        z[:, : W // 8] += 0.25                                           # a depth step: no triangles across it
        z[:6, :] = 0.0
        out.append(np.round(z * 1000).astype(np.uint16))
    K = np.array([[150.0, 0, W / 2], [0, 150.0, H / 2], [0, 0, 1]], dtype=np.float32)
    return out[0], out[1], K


def _aa_to_matrix(aa):
    """pytorch3d.transforms.axis_angle_to_matrix stand-in (pytorch3d is absent and un-pinned upstream): the published
    route through the unit quaternion (axis_angle_to_quaternion, quaternion_to_matrix)."""
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    s = torch.empty_like(angles)
    s[~small] = torch.sin(half[~small]) / angles[~small]
    s[small] = 0.5 - (angles[small] * angles[small]) / 48
    q = torch.cat([torch.cos(half), aa * s], dim=-1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def F15_embedded_deformation(reg_mod, loss_mod, EasyDict, **_):
    """N-ICP baseline: the deformation graph the reference's own MVRegC build (oracle/_ref, compiled from
    /root/reference/cxx by oracle/Makefile.ref) produces from a synthetic depth map, and optimize_Embeded_deformation end
    to end on the synthetic depth pair with every evaluated (cd, arap) recorded."""
    import importlib.util
    import tempfile
    from PIL import Image
    so = [f for f in os.listdir(os.path.join(os.path.dirname(OUT), "..", "oracle", "_ref")) if f.startswith("MVRegC")]
    assert so, "build oracle/_ref first: make -C oracle -f Makefile.ref"
    spec = importlib.util.spec_from_file_location("MVRegC", os.path.join(os.path.dirname(OUT), "..", "oracle", "_ref", so[0]))
    MVRegC = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(MVRegC)
    import model.geometry as geo
    geo.MVRegC = MVRegC
    sys.modules["skimage.io"].imread = lambda path: np.array(Image.open(path))
    tr = types.ModuleType("pytorch3d.transforms")
    tr.axis_angle_to_matrix = _aa_to_matrix
    sys.modules["pytorch3d.transforms"] = tr
    sys.modules["pytorch3d"].transforms = tr
    reg_mod.pytorch3d = sys.modules["pytorch3d"]
    for name in ("get_deformation_graph_from_depthmap", "map_pixel_to_pcd", "depth_2_pc", "ED_warp", "pc_2_uv"):
        setattr(reg_mod, name, getattr(geo, name))          # what `Runbaselines = True` (registration.py:9-11) would import
    reg_mod.io.imread = sys.modules["skimage.io"].imread
    d0, d1, K = synthetic_depth_pair(0)
    cfg = EasyDict(dict(deformation_model="ED", device=torch.device("cpu"), iters=25, lr=0.02, max_break_count=30,
                        break_threshold_ratio=0.01, w_ldmk=1, w_cd=1, w_arap=0.5, samples=600, max_triangle_distance=0.06,
                        node_coverage=0.09, USE_ONLY_VALID_VERTICES=True, num_neighbors=8, ENFORCE_TOTAL_NUM_NEIGHBORS=False,
                        SAMPLE_RANDOM_SHUFFLE=False, REMOVE_NODES_WITH_NOT_ENOUGH_NEIGHBORS=False))
    out = {"depth_src": d0, "depth_tgt": d1, "K": K}
    for tag, remove, cov in (("g", False, 0.09), ("gr", True, 0.05)):
        c = EasyDict(dict(cfg, REMOVE_NODES_WITH_NOT_ENOUGH_NEIGHBORS=remove, node_coverage=cov))
        data = geo.get_deformation_graph_from_depthmap(d0.copy(), K, c)
        out[f"{tag}.coverage"] = np.float32(cov)
        out[f"{tag}.graph_nodes"] = data["graph_nodes"].numpy()
        out[f"{tag}.graph_edges"] = data["graph_edges"].numpy().astype(np.int32)
        out[f"{tag}.graph_edges_weights"] = data["graph_edges_weights"].numpy()
        pa, pw = data["pixel_anchors"].numpy(), data["pixel_weights"].numpy()
        out[f"{tag}.pixel_anchors_rows"] = pa[::7, ::5].copy()
        out[f"{tag}.pixel_weights_rows"] = pw[::7, ::5].copy()
        out[f"{tag}.pixel_anchors_sum"] = np.int64(pa.astype(np.int64).sum())
        out[f"{tag}.pixel_anchors_hash"] = np.int64((pa.astype(np.int64) * (1 + np.arange(pa.size).reshape(pa.shape) % 9973)).sum())
        out[f"{tag}.pixel_weights_sum"] = np.float64(pw.astype(np.float64).sum())
        out[f"{tag}.valid_pixels"] = np.int64((pa.sum(-1) > -4).sum())
        print(f"  graph {tag}: {data['graph_nodes'].shape[0]} nodes, {int((pa.sum(-1) > -4).sum())} anchored pixels", flush=True)
    vertices, faces, vpix, pim = geo.depth_to_mesh(d0.copy(), d0 > 0, K, max_triangle_distance=0.06, depth_scale=1000.)
    out["mesh.counts"] = np.array([vertices.shape[0], faces.shape[0]])
    out["mesh.vsum"] = vertices.astype(np.float64).sum(0)
    out["mesh.fhash"] = np.int64((faces.astype(np.int64) * np.array([1, 3, 7])).sum())
    out["mesh.faces_head"] = faces[:32].copy()
    out["mesh.vpix_head"] = vpix[:32].copy()
    # ---- the optimisation loop on the depth pair
    with tempfile.TemporaryDirectory() as td:
        ps, pt = os.path.join(td, "s.png"), os.path.join(td, "t.png")
        Image.fromarray(d0).save(ps)
        Image.fromarray(d1).save(pt)
        trace = []
        orig_cd, orig_arap = reg_mod.compute_truncated_chamfer_distance, reg_mod.arap_cost

        def cd_hook(*a, **k):
            v = orig_cd(*a, **k)
            trace.append([v.item(), None])
            return v

        def arap_hook(*a, **k):
            v = orig_arap(*a, **k)
            trace[-1][1] = v.item()
            return v

        reg_mod.compute_truncated_chamfer_distance, reg_mod.arap_cost = cd_hook, arap_hook
        try:
            torch.manual_seed(11)
            model = reg_mod.Registration(cfg)
            pim0 = geo.depth_2_pc(d0 / 1000.0, K).transpose(1, 2, 0)
            src_pcd = torch.from_numpy(pim0[d0 > 0]).float()[::9].contiguous()        # the "sampled" cloud of the dataset item
            pim1 = geo.depth_2_pc(d1 / 1000.0, K).transpose(1, 2, 0)
            tgt_pcd = torch.from_numpy(pim1[d1 > 0]).float()[::9].contiguous()
            model.load_pcds(src_pcd, tgt_pcd)
            model.load_raw_pcds_from_depth(ps, pt, K, landmarks=None)
            warped, valid_id = model.register()
        finally:
            reg_mod.compute_truncated_chamfer_distance, reg_mod.arap_cost = orig_cd, orig_arap
    out["e2e.src_pcd"], out["e2e.tgt_pcd"] = src_pcd.numpy(), tgt_pcd.numpy()
    out["e2e.cd_trace"] = np.array([t[0] for t in trace], dtype=np.float64)
    out["e2e.arap_trace"] = np.array([t[1] for t in trace], dtype=np.float64)
    out["e2e.warped"] = warped.detach().numpy()
    out["e2e.valid_id"] = valid_id.numpy()
    out["e2e.seed"] = np.int64(11)
    print(f"  loop: {len(trace)} evaluations, cd {trace[0][0]:.5f} -> {trace[-1][0]:.5f}, arap {trace[-1][1]:.3e}", flush=True)
    save("F15_embedded_deformation", **out)


# width / depth other than the shipped 128 / 3 (model/nets.py:65-110,295-304 build whatever the YAML names): the seeded initialisation,
# one level's forward + gradients, the whole pyramid, and a short traced register() -- for the generic kernels (csrc/ndp_generic.inc)
F16_SHAPES = {
    "w64d2_se3aa": dict(width=64, depth=2, rotation_format="axis_angle", motion="SE3"),
    "w256d4_sim3eu": dict(width=256, depth=4, rotation_format="euler", motion="Sim3"),
    "w32d1_sflow": dict(width=32, depth=1, rotation_format="axis_angle", motion="sflow"),
    "w100d3_se3quat_nr": dict(width=100, depth=3, rotation_format="quaternion", motion="SE3", nonrigidity_est=True),
}


def F16_generic_width(nets, reg_mod, loss_mod, EasyDict, **_):
    out = {}
    g = torch.Generator().manual_seed(16)
    x = torch.rand(200, 3, generator=g) - 0.5
    coef = torch.linspace(-1.0, 1.0, 200 * 3).reshape(200, 3)
    out["x"] = x.numpy()
    out["head_scale"] = np.float32(30.0)
    out["seed"] = np.int64(16)
    out["level"] = np.int64(3)
    for tag, kw in F16_SHAPES.items():
        torch.manual_seed(16)
        pyr = nets.Deformation_Pyramid(device="cpu", k0=-8, m=5, **kw)
        names, sums, asums, heads = [], [], [], []
        for li, layer in enumerate(pyr.pyramid):
            for k, v in layer.named_parameters():
                a = v.detach().double().numpy().ravel()
                names.append(f"{li}.{k}")
                sums.append(a.sum())
                asums.append(np.abs(a).sum())
                h = np.zeros(8)
                h[:min(8, a.size)] = a[:8]
                heads.append(h)
        out[f"{tag}.names"] = np.array(names)
        out[f"{tag}.sum"] = np.array(sums)
        out[f"{tag}.abssum"] = np.array(asums)
        out[f"{tag}.head8"] = np.array(heads)
        out[f"{tag}.perm4096"] = torch.randperm(4096)[:16].numpy()
        layer = pyr.pyramid[3]
        with torch.no_grad():
            for k, v in layer.named_parameters():
                if "branch" in k or "brach" in k:
                    v.mul_(30.0)
        y, data = pyr.warp(x, max_level=3, min_level=3)
        out[f"{tag}.out"] = y.detach().numpy()
        for p in layer.parameters():
            p.grad = None
        (y * coef).sum().backward()
        for k, v in layer.named_parameters():
            out[f"{tag}.grad.{k}"] = v.grad.numpy().copy()
        with torch.no_grad():
            yfull, _ = pyr.warp(x)
        out[f"{tag}.full_out"] = yfull.numpy()
    # a short register() at width 64 / depth 2 (NDP.yaml otherwise): loss trace, evaluations per level, warped points
    src, tgt, flow_gt, overlap = synthetic_pair(6, n_total=2048)
    cfg = ndp_config(EasyDict, samples=256, width=64, depth=2, m=5, iters=60)
    warped, trace = _register_traced(reg_mod, EasyDict, cfg, src, tgt, seed=0)
    out["reg.src"], out["reg.tgt"] = src.numpy(), tgt.numpy()
    out["reg.warped"] = warped.numpy()
    out["reg.iters_per_level"] = np.array([len(t) for t in trace])
    out["reg.loss_trace"] = np.array(sum(trace, []), dtype=np.float64)
    save("F16_generic_width", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--bench-pairs", type=int, default=8)
    args = ap.parse_args()
    EasyDict = install_stubs()
    torch.set_num_threads(8)
    import model.nets as nets
    import model.loss as loss_mod
    import model.registration as reg_mod
    ctx = dict(nets=nets, loss_mod=loss_mod, reg_mod=reg_mod, EasyDict=EasyDict,
               bench_pairs=args.bench_pairs)
    todo = {
        "F1": F1_init, "F2": F2_layer_forward, "F3": F3_chamfer, "F4": F4_F5_iteration,
        "F7": F7_end_to_end, "F8": F8_metrics, "F9": F9_landmarks, "F9b": F9b_lndp_end_to_end,
        "F9c": F9c_mixed_landmark_chamfer,
        "F10": F10_benchmark, "F10b": F10b_surface_benchmark, "F10c": F10c_surface_benchmark_seeds, "F11": F11_nonrigidity, "F12": F12_nsfp,
        "F13": F13_shape_transfer, "F14": F14_nerfies, "F15": F15_embedded_deformation, "F16": F16_generic_width,
    }
    only = [s for s in args.only.split(",") if s]
    for k, fn in todo.items():
        if only and k not in only:
            continue
        print(f"--- {k}", flush=True)
        fn(**ctx)


if __name__ == "__main__":
    main()
