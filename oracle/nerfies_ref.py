"""CPU restatement of the Nerfies comparison baseline (second half) -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and tools/, never by the product).  Plain torch-CPU fp32; pinned by tests/test_nerfies.py against
golden F14 captured from the reference.

What it follows, by reference line (nothing is imported from /root/reference):
    Nerfies_Deformation          model/nets.py:187-253   (39-wide windowed posenc :218-240, input 39->128, six 128x128 ReLU layers
                                                          (MLP depth 7, :295-304), w / v heads, SE(3) exponential :242-253)
    exp_se3                      model/rigid_body.py:97-110
    nerfies_regularization       model/loss.py:373-379   (mean over points of log(max singular value of J)^2, double, clamp 1e-6)
    optimize_Nerfies             model/registration.py:265-339

Stated differently from upstream on purpose: the per-point Jacobian d warp / d x is carried FORWARD through the network as
three tangent vectors per point (dh = (W dh_prev) * [z > 0]), which is what the HIP kernels do, instead of calling autograd's
jacobian.  Upstream builds J with create_graph=False, so the regulariser is a constant for autograd: it moves the loss value
and the stop rule, not the gradients -- reproduced here (J is computed under no_grad).

Flat parameter layout (the product's, include/ndp_types.h):  [W_in 128x39 | b_in 128 | (W_l 128x128 | b_l 128) x 6 | W_h 6x128 | b_h 6]
with head rows 0..2 = w_branch, 3..5 = v_branch.   P = 104 966.
"""
import math

import torch
import torch.nn.functional as F

W, M_FREQ, K0, DIM_PE = 128, 6, -3, 39
PI = 3.14                                  # (sic) the reference's constant, nets.py:220
P_COUNT = W * DIM_PE + W + 6 * (W * W + W) + 6 * W + 6


def split(flat):
    o, out = 0, []
    for shape in [(W, DIM_PE), (W,)] + [(W, W), (W,)] * 6 + [(6, W), (6,)]:
        n = math.prod(shape)
        out.append(flat[o:o + n].reshape(shape))
        o += n
    return out


def window(it, max_iter):
    """Annealing weights of the six frequency bands at iteration `it` (nets.py:223-225)."""
    a = M_FREQ * it / (0.6 * max_iter)
    return (1 - torch.cos(torch.clamp(a - torch.arange(M_FREQ).float(), min=0, max=1) * PI)) / 2


def posenc(x, it, max_iter):
    """-> pe [n,39] = [x | sin_x(6) cos_x(6) | sin_y cos_y | sin_z cos_z] and its derivative dpe [n,3,39] wrt x."""
    n = x.shape[0]
    w = window(it, max_iter)
    f = (2.0 ** (torch.arange(M_FREQ).float() + K0)) * PI
    pe = torch.zeros(n, DIM_PE)
    dpe = torch.zeros(n, 3, DIM_PE)
    pe[:, :3] = x
    for a in range(3):
        arg = x[:, a:a + 1] * f[None]
        s, c = torch.sin(arg) * w, torch.cos(arg) * w
        pe[:, 3 + 12 * a:9 + 12 * a] = s
        pe[:, 9 + 12 * a:15 + 12 * a] = c
        dpe[:, a, a] = 1.0
        dpe[:, a, 3 + 12 * a:9 + 12 * a] = c * f
        dpe[:, a, 9 + 12 * a:15 + 12 * a] = -s * f
    return pe, dpe


def skew(w):
    z = torch.zeros_like(w[..., 0])
    return torch.stack([z, -w[..., 2], w[..., 1], w[..., 2], z, -w[..., 0], -w[..., 1], w[..., 0], z], -1).reshape(*w.shape[:-1], 3, 3)


def se3_warp(o, x):
    """o [..,6] = (w, v) raw head outputs; x [..,3] -> R(w) x + t(w, v)."""
    w, v = o[..., :3], o[..., 3:]
    th = w.norm(dim=-1, keepdim=True)
    K = skew(w / th)
    vh = v / th
    th = th[..., None]
    eye = torch.eye(3).expand_as(K)
    R = eye + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)
    Pm = eye + (1 - torch.cos(th)) * K + (th - torch.sin(th)) * (K @ K)
    return (R @ x[..., None] + Pm @ vh[..., None])[..., 0]


def forward(flat, x, it, max_iter, want_jacobian=True):
    """-> (warped [n,3] (differentiable wrt flat), J [n,3,3] (constant), pe)."""
    p = split(flat)
    pe, dpe = posenc(x, it, max_iter)
    h = F.relu(F.linear(pe, p[0], p[1]))
    hs = [h]
    for l in range(6):
        h = F.relu(F.linear(h, p[2 + 2 * l], p[3 + 2 * l]))
        hs.append(h)
    o = F.linear(h, p[14], p[15])
    warped = se3_warp(o, x)
    J = None
    if want_jacobian:
        with torch.no_grad():
            t = (dpe @ p[0].T) * (hs[0] > 0)[:, None, :]                     # [n,3,128] tangents of h0
            for l in range(6):
                t = (t @ p[2 + 2 * l].T) * (hs[l + 1] > 0)[:, None, :]
            D = t @ p[14].T                                                     # [n,3,6]  d o / d x_k
            cols = []
            for k in range(3):
                e = torch.zeros_like(x)
                e[:, k] = 1.0
                _, jv = torch.func.jvp(se3_warp, (o.detach(), x), (D[:, k], e))
                cols.append(jv)
            J = torch.stack(cols, dim=-1)                                       # J[n, a, k] = d warped_a / d x_k
    return warped, J, pe


def regularization(J, eps=1e-6):
    sv = torch.linalg.svdvals(J.double())
    sv = torch.clamp(sv, min=eps)
    return (torch.log(sv.max(dim=1)[0]) ** 2).mean()


def nearest(a, b):
    with torch.no_grad():
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1).argmin(dim=1)


def chamfer_l1(x, y):
    dx = ((x - y[nearest(x, y)]) ** 2).sum(-1).sqrt()
    dy = ((y - x[nearest(y, x)]) ** 2).sum(-1).sqrt()
    return dx.mean() + dy.mean()


def optimize(flat, s_sample, t_sample, iters, lr=0.01, max_break_count=70, ratio=0.001):
    """registration.py:300-331 -> (flat, last iteration index, [(cd, reg)] per evaluation)."""
    flat = flat.clone().requires_grad_(True)
    opt = torch.optim.Adam([flat], lr=lr)
    break_counter, loss_prev, trace, i = 0, 1e6, [], 0
    for i in range(iters):
        warped, J, _ = forward(flat, s_sample, i, iters)
        reg = regularization(J)
        cd = chamfer_l1(warped, t_sample)
        loss = cd + 0.001 * reg
        trace.append((cd.item(), reg.item()))
        L = loss.item()
        if L < 1e-4:
            break
        if abs(loss_prev - L) < loss_prev * ratio:
            break_counter += 1
        if break_counter >= max_break_count:
            break
        loss_prev = L
        opt.zero_grad()
        loss.backward()
        opt.step()
    return flat.detach(), i, trace
