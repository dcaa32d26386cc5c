"""BLAS-backed torch-CPU restatement of the NDP per-pair loop (SE3 / axis-angle, Chamfer objective).

TEST INFRASTRUCTURE ONLY, like oracle/ndp_oracle.c: imported by tests/ and by bench.py's cpu_baseline leg, never by the
product.  It exists because the C oracle is a scalar fmaf-chain port (faithful bits, but no BLAS): the reference's own
CPU path runs its three linear layers through the host BLAS, and a baseline that does not would flatter the GPU.  This
file states the same algorithm on plain torch ops -- F.linear (sgemm), autograd, torch.optim.Adam -- so that
`cpu_baseline` can quote both ("port-c" and "port-torch").  Pinned by tests/test_oracle_golden.py against the reference's
goldens (F4/F5 loss traces) and against the C oracle.

What it follows, by reference line (nothing is imported from /root/reference):
    level forward       model/nets.py:111-140 (posenc :164-177, MLP :295-304, axis-angle exp map :150-153 + rigid_body.py:113-119)
    truncated Chamfer   model/loss.py:94-258 with knn_points semantics (exact brute-force 1-NN, squared L2)
    level / Adam loop   model/registration.py:170-249 (fresh Adam per level :176, stop rule :226-232, hand-over :242-249)
    inference warp      model/registration.py:253-258
The parameters arrive as the product's flat per-level blocks ([W0 | b0 | W1 | b1 | W2 | b2 | Wr | Wt | br | bt] in the
layout of include/ndp_types.h); they are re-viewed as leaf tensors per level.
"""
import torch
import torch.nn.functional as F

W = 128


def split_level(flat):
    """One level's flat SE3 / axis-angle block (P = 34 694) -> list of leaf tensors [W0,b0,W1,b1,W2,b2,Wh,bh] (copies)."""
    o = 0
    out = []
    for shape in ((W, 6), (W,), (W, W), (W,), (W, W), (W,), (6, W), (6,)):
        n = 1
        for s in shape:
            n *= s
        out.append(flat[o:o + n].reshape(shape).clone().requires_grad_(True))
        o += n
    return out


def level_forward(p, x, level, k0=-8, mlp_scale=0.001):
    W0, b0, W1, b1, W2, b2, Wh, bh = p
    f = 2.0 ** (level + 1 + k0)
    fx = f * x
    pe = torch.stack([fx.sin(), fx.cos()], dim=-1).reshape(x.shape[0], 6)       # [sin fx0, cos fx0, sin fx1, ...]
    h = F.relu(F.linear(pe, W0, b0))
    h = F.relu(F.linear(h, W1, b1))
    h = F.relu(F.linear(h, W2, b2))
    o = mlp_scale * F.linear(h, Wh, bh)                                          # rows: 3 rotation, 3 translation
    r, t = o[:, :3], o[:, 3:]
    theta = r.norm(dim=1, keepdim=True)
    w = r / theta
    z = torch.zeros_like(w[:, 0])
    K = torch.stack([z, -w[:, 2], w[:, 1], w[:, 2], z, -w[:, 0], -w[:, 1], w[:, 0], z], dim=1).reshape(-1, 3, 3)
    eye = torch.eye(3, dtype=x.dtype).expand_as(K)
    R = eye + theta.sin()[..., None] * K + ((1.0 - theta.cos())[..., None] * K) @ K
    return (R @ x[..., None])[..., 0] + t


def nearest(a, b, chunk=1024):
    """Index of the exact nearest point of b for every point of a (lowest index on ties)."""
    with torch.no_grad():
        idx = torch.empty(a.shape[0], dtype=torch.int64)
        for s in range(0, a.shape[0], chunk):
            d = ((a[s:s + chunk, None, :] - b[None, :, :]) ** 2).sum(-1)
            idx[s:s + chunk] = d.argmin(dim=1)
    return idx


def chamfer_l1(x, y, trunc=1e9):
    dx = ((x - y[nearest(x, y)]) ** 2).sum(-1)
    dy = ((y - x[nearest(y, x)]) ** 2).sum(-1)
    cx = torch.where(dx >= trunc, torch.zeros_like(dx), dx.sqrt())
    cy = torch.where(dy >= trunc, torch.zeros_like(dy), dy.sqrt())
    return cx.sum() / x.shape[0] + cy.sum() / y.shape[0]


def optimize(params_all, s_sample, t_sample, m=9, k0=-8, iters=500, lr=0.01, max_break_count=15, ratio=0.001,
             early_stop=True):
    """params_all: [m][P] flat blocks.  -> (levels' parameter lists, warped samples, loss trace, Adam steps)."""
    levels = [split_level(params_all[i]) for i in range(m)]
    x = s_sample
    trace, steps = [], 0
    for level in range(m):
        opt = torch.optim.Adam(levels[level], lr=lr)
        break_counter, loss_prev = 0, 1e6
        warped = x
        for _ in range(iters):
            warped = level_forward(levels[level], x, level, k0)
            loss = chamfer_l1(warped, t_sample)
            L = loss.item()
            trace.append(L)
            if early_stop:
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
            steps += 1
        x = warped.detach()
    return levels, x, trace, steps


def pyramid_forward(levels, x, k0=-8):
    with torch.no_grad():
        for level, p in enumerate(levels):
            x = level_forward(p, x, level, k0)
    return x
