"""Nerfies comparison baseline on the HIP path (second half).

    Nerfies_Deformation               /root/reference/model/nets.py:187-253   (windowed 39-wide posenc -> 39->128 -> six
                                      128x128 ReLU layers -> w / v heads -> SE(3) exponential warp, + per-point Jacobian)
    nerfies_regularization            /root/reference/model/loss.py:373-379
    optimize_Nerfies(reg)             /root/reference/model/registration.py:265-339

One launch per layer in libndp_hip.so (`ndp_nerfies_fwd` / `ndp_nerfies_bwd`).  The Jacobian d warp / d x that the
regulariser needs is carried FORWARD through the network as three tangent rows per point (the 128x128 layers run on the
fp32 MFMA for the primal row and its three tangents, the tangents masked by the primal's ReLU pattern); the log-singular-
value regulariser is evaluated per point on the device in double.  Upstream builds the Jacobian with
`torch.autograd.functional.jacobian(create_graph=False)`, so the regulariser is a constant for autograd: it enters the
loss value and the stop rule, not the gradients -- the backward here is therefore the Chamfer gradient through the primal
chain, on the same layer kernels as NDP and NSFP.  The loop is host driven with one `loss.item()` per iteration, as
upstream (it is a comparison baseline, not the hot path).  No CPU fallback.
"""
import math

import torch

from . import _native as N
from . import ops

W, DIM_PE, N_HID = 128, 39, 6
OFF_WIN, OFF_BIN = 0, W * DIM_PE
OFF_L0 = OFF_BIN + W


def off_W(l):                       # hidden layer l = 1..6
    return OFF_L0 + (l - 1) * (W * W + W)


def off_b(l):
    return off_W(l) + W * W


OFF_WH = off_W(N_HID + 1)
OFF_BH = OFF_WH + 6 * W
PARAM_COUNT = OFF_BH + 6
P_STRIDE = (PARAM_COUNT + 63) // 64 * 64


class Nerfies_Deformation:
    """Flat parameter block with the reference's module names as views.  Initialised by constructing the `torch.nn.Linear`
    modules on the CPU generator in the reference's order (nets.py:195-201: input, six mlp layers, w_branch, v_branch), so
    the RNG stream is consumed exactly as upstream consumes it."""

    def __init__(self, depth=7, width=W, max_iter=5000, device="cpu"):
        if depth != 7 or width != W:
            raise N.NdpError("the HIP kernels serve Nerfies_Deformation(depth=7, width=128) only")
        self.max_iter = max_iter
        flat = torch.zeros(P_STRIDE, dtype=torch.float32)
        lin = torch.nn.Linear(DIM_PE, W)
        flat[OFF_WIN:OFF_BIN] = lin.weight.detach().reshape(-1)
        flat[OFF_BIN:OFF_L0] = lin.bias.detach()
        for l in range(1, N_HID + 1):
            lin = torch.nn.Linear(W, W)
            flat[off_W(l):off_b(l)] = lin.weight.detach().reshape(-1)
            flat[off_b(l):off_b(l) + W] = lin.bias.detach()
        for r in (0, 3):                                        # w_branch then v_branch: head rows 0..2 and 3..5
            lin = torch.nn.Linear(W, 3)
            flat[OFF_WH + r * W:OFF_WH + (r + 3) * W] = lin.weight.detach().reshape(-1)
            flat[OFF_BH + r:OFF_BH + r + 3] = lin.bias.detach()
        self.flat = flat.to(device)

    def to(self, device):
        self.flat = self.flat.to(device)
        return self

    def split_like(self, v):
        """Views of a flat [>= PARAM_COUNT] vector in named_parameters() order."""
        out = [v[OFF_WIN:OFF_BIN].view(W, DIM_PE), v[OFF_BIN:OFF_L0]]
        for l in range(1, N_HID + 1):
            out += [v[off_W(l):off_b(l)].view(W, W), v[off_b(l):off_b(l) + W]]
        out += [v[OFF_WH:OFF_WH + 3 * W].view(3, W), v[OFF_BH:OFF_BH + 3],
                v[OFF_WH + 3 * W:OFF_WH + 6 * W].view(3, W), v[OFF_BH + 3:OFF_BH + 6]]
        return out

    def named_parameters(self):
        names = ["input.0.weight", "input.0.bias"] + [f"mlp.pts_linears.{i}.{k}" for i in range(N_HID) for k in ("weight", "bias")] + \
                ["w_branch.weight", "w_branch.bias", "v_branch.weight", "v_branch.bias"]
        return zip(names, self.split_like(self.flat))


def window(it, max_iter):
    """Annealing weights of the six frequency bands (nets.py:223-225), as the float32 values torch produces upstream."""
    a = 6 * it / (0.6 * max_iter)
    w = (1 - torch.cos(torch.clamp(a - torch.arange(6).float(), min=0, max=1) * 3.14)) / 2
    return [float(v) for v in w]


def nerfies_fwd(params, x, it, max_iter, save=False):
    """x [n,3] -> (warped [n,3], J [n,3,3], reg [1] = mean log(sigma_max)^2) ; with save=True also the primal activation
    planes and the head record for nerfies_bwd."""
    ops._chk(params, "params"); ops._chk(x, "x")
    n, c = x.shape[0], ops.cap(x.shape[0])
    dev = x.device
    out = torch.empty_like(x)
    J = torch.empty(n, 9, device=dev, dtype=torch.float32)
    reg = torch.empty(1, device=dev, dtype=torch.float32)
    act = torch.empty(2 if not save else N_HID + 1, 4 * c, W, device=dev, dtype=torch.float32)
    pe = torch.empty(c, 40, device=dev, dtype=torch.float32)
    heads = torch.empty(4 * c, 8, device=dev, dtype=torch.float32)
    work = torch.empty(c, device=dev, dtype=torch.float64)
    w = (ctypes_f6)(*window(it, max_iter))
    N.check(N.lib().ndp_nerfies_fwd(ops._p(params), ops._p(x), n, w, ops._p(out), ops._p(J), ops._p(reg), ops._p(act),
                                    1 if save else 0, ops._p(pe), ops._p(heads), ops._p(work), N.stream_ptr(dev)), "ndp_nerfies_fwd")
    J = J.view(n, 3, 3)
    return (out, J, reg, (act, pe, heads)) if save else (out, J, reg)


def nerfies_bwd(params, x, saved, g, n_part=None):
    """-> grads [PARAM_COUNT] of a scalar loss given g = dL/d(warped) [n,3] (the regulariser carries no gradient upstream).
    `saved` (from nerfies_fwd(save=True)) is consumed."""
    act, pe, heads = saved
    ops._chk(params, "params"); ops._chk(x, "x"); ops._chk(g, "g")
    n = x.shape[0]
    tiles = (n + N.TILE - 1) // N.TILE
    if n_part is None:
        n_part = min(tiles, 256)
    part = torch.empty(n_part, P_STRIDE, device=x.device, dtype=torch.float32)
    work = torch.empty(ops.cap(n), N.NHMAX, device=x.device, dtype=torch.float32)
    st = N.stream_ptr(x.device)
    N.check(N.lib().ndp_nerfies_bwd(ops._p(params), ops._p(x), n, ops._p(act), ops._p(pe), ops._p(heads), ops._p(g),
                                    ops._p(work), ops._p(part), n_part, P_STRIDE, st), "ndp_nerfies_bwd")
    grads = torch.empty(PARAM_COUNT, device=x.device, dtype=torch.float32)
    N.check(N.lib().ndp_grad_reduce(ops._p(part), n_part, P_STRIDE, PARAM_COUNT, ops._p(grads), st), "ndp_grad_reduce")
    return grads


import ctypes  # noqa: E402
ctypes_f6 = ctypes.c_float * 6


def optimize_Nerfies(reg_obj, visualize=False):
    """registration.py:265-339, same order of operations: model, centring, two randperms, Adam(lr), loss = cd + 0.001 reg,
    the early-stop rule on the Python float of the loss, final warp of ALL source points at the last iteration index,
    `+ tgt_mean`.  Returns (warped, None)."""
    if visualize:
        raise NotImplementedError("mayavi visualisation is outside the hot path")
    config = reg_obj.config
    dev = reg_obj._dev()
    net = Nerfies_Deformation(max_iter=config.iters).to(dev)                       # :276
    reg_obj.src_pcd = reg_obj.src_pcd.to(dev).float()
    tgt_all = reg_obj.tgt_pcd.to(dev).float()
    src_mean = reg_obj.src_pcd.mean(dim=0, keepdim=True)                           # :286-289
    tgt_mean = tgt_all.mean(dim=0, keepdim=True)
    src_pcd = (reg_obj.src_pcd - src_mean).contiguous()
    tgt_pcd = tgt_all - tgt_mean
    src = torch.randperm(src_pcd.shape[0])                                         # :294-297 (CPU RNG)
    tgt = torch.randperm(tgt_pcd.shape[0])
    s_sample = src_pcd[src[: config.samples].to(dev)].contiguous()
    t_sample = tgt_pcd[tgt[: config.samples].to(dev)].contiguous()
    params = net.flat
    m = torch.zeros(PARAM_COUNT, device=dev)
    v = torch.zeros(PARAM_COUNT, device=dev)
    break_counter, loss_prev, steps, i, L = 0, 1e6, 0, 0, float("nan")
    trace = []
    for i in range(config.iters):                                                  # :306-331
        warped, J, reg, saved = nerfies_fwd(params, s_sample, i, config.iters, save=True)
        cd, gx, _ = ops.chamfer_l1(warped, t_sample, 1e9)
        L = cd.item() + 0.001 * reg.item()
        trace.append((cd.item(), reg.item()))
        if L < 1e-4:
            break
        if abs(loss_prev - L) < loss_prev * config.break_threshold_ratio:
            break_counter += 1
        if break_counter >= config.max_break_count:
            break
        loss_prev = L
        grads = nerfies_bwd(params, s_sample, saved, gx)
        steps += 1
        ops.adam_step(params[:PARAM_COUNT], grads, m, v, steps, lr=config.lr)
    reg_obj.last_nerfies = dict(iters=steps, loss=L, trace=trace)
    warped_pcd, _, _ = nerfies_fwd(params, src_pcd, i, config.iters)               # :334
    return warped_pcd + tgt_mean, None
