"""CPU restatement of the embedded-deformation N-ICP loop -- TEST INFRASTRUCTURE ONLY (imported by
tests/ and tools/, never by the product).  Plain torch-CPU fp32 + autograd; pinned by tests/test_ed.py against golden F15
captured from the reference (whose graph came from its own MVRegC build).

What it follows, by reference line (nothing is imported from /root/reference):
    ED_warp                          model/geometry.py:37-49       y = sum_k w_k (R_k (x - g_k) + g_k + t_k)
    arap_cost                        model/loss.py:261-285         mean_{i, slot} w_ij |R_i (g_j - g_i) + g_i + t_i - g_j - t_j|^2
    optimize_Embeded_deformation     model/registration.py:342-467 Adam([phi, t], lr) + ExponentialLR(0.999), fresh randperm
                                                                   samples every iteration, loss = w_cd cd + w_arap arap
    axis_angle_to_matrix             pytorch3d.transforms (absent, un-pinned upstream): the published route through the unit
                                     quaternion, with the Taylor branch below 1e-6 rad
Two upstream quirks are reproduced: a -1 anchor / edge slot indexes the LAST node (python negative indexing) with weight 0,
and `loss_prev` is never updated inside this loop (:428-433), so the relative-change stop never fires -- only loss < 1e-5
or the iteration cap end it.
"""
import torch


def axis_angle_to_matrix(aa):
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * angles
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    s = torch.where(small, 0.5 - angles * angles / 48, torch.sin(half) / safe)
    q = torch.cat([torch.cos(half), aa * s], dim=-1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def ed_warp(x, anchors, weights, nodes, R, t):
    """x [n,3]; anchors [n,6] (long, -1 = last node with weight 0); weights [n,6]."""
    g = nodes[anchors]                                    # [n,6,3]
    y = ((R[anchors] @ (x[:, None] - g)[..., None])[..., 0] + g + t[anchors]) * weights[..., None]
    return y.sum(dim=1)


def arap(R, t, nodes, edges, w):
    gi, ti = nodes[:, None], t[:, None]
    gj, tj = nodes[edges], t[edges]
    e = (((R[:, None] @ (gj - gi)[..., None])[..., 0] + gi + ti - gj - tj) ** 2).sum(-1)
    return (w * e).mean()


def nearest(a, b):
    with torch.no_grad():
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1).argmin(dim=1)


def chamfer_l1(x, y):
    return ((x - y[nearest(x, y)]) ** 2).sum(-1).sqrt().mean() + ((y - x[nearest(y, x)]) ** 2).sum(-1).sqrt().mean()


def optimize(src_raw, tgt_raw, anchors, weights, nodes, edges, edge_w, iters, samples, lr=0.02, w_cd=1.0, w_arap=0.5):
    """registration.py:352-447 -> (phi, t, [(cd, arap)] per evaluation).  Consumes torch's CPU generator like upstream."""
    n = nodes.shape[0]
    phi = torch.zeros(n, 3, requires_grad=True)
    t = torch.zeros(n, 3, requires_grad=True)
    opt = torch.optim.Adam([phi, t], lr=lr)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.999)
    trace = []
    for _ in range(iters):
        R = axis_angle_to_matrix(phi)
        si = torch.randperm(src_raw.shape[0])[:samples]
        ti = torch.randperm(tgt_raw.shape[0])[:samples]
        warped = ed_warp(src_raw[si], anchors[si], weights[si], nodes, R, t)
        cd = chamfer_l1(warped, tgt_raw[ti])
        reg = arap(R, t, nodes, edges, edge_w)
        loss = cd * w_cd + reg * w_arap
        trace.append((cd.item(), reg.item()))
        if loss.item() < 1e-5:
            break
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
    return phi.detach(), t.detach(), trace
