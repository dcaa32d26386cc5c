"""Neural scene-flow prior baseline on the HIP path.

    Neural_Prior                      /root/reference/model/nets.py:256-292   (3 -> 128 x 8 -> 3 ReLU MLP)
    optimize_neural_SFlow(reg)        /root/reference/model/registration.py:470-540

The network runs one launch per layer in libndp_hip.so (`ndp_nsfp_fwd` / `ndp_nsfp_bwd`: the seven 128x128 layers on
the fp32 MFMA with weight-stationary registers, forward and backward sharing the level kernels of the NDP path), the
Chamfer loss and Adam are the NDP operators.  Unlike the NDP engine the loop itself is host driven, one `loss.item()`
per iteration, exactly like the reference: this is a comparison baseline, not the hot path.  No CPU fallback.
"""
import ctypes
import math

import torch

from . import _native as N
from . import ops

W = 128
N_LAYERS = 9


def off_W(l):
    return 0 if l == 1 else W * 4 + (l - 2) * (W * W + W)


def off_b(l):
    return off_W(l) + (W * 3 if l == 1 else (3 * W if l == N_LAYERS else W * W))


PARAM_COUNT = off_b(N_LAYERS) + 3
P_STRIDE = (PARAM_COUNT + 63) // 64 * 64


def layer_shapes():
    return [(W, 3)] + [(W, W)] * 7 + [(3, W)]


class Neural_Prior:
    """Flat parameter block with the reference's module names as views (`layer1.weight` ... `layer9.bias`).
    Initialised by constructing the nine `torch.nn.Linear` on the CPU generator, in the reference's order, so the
    RNG stream is consumed exactly as `Neural_Prior()` consumes it (nets.py:260-272)."""

    def __init__(self, dim_x=3, filter_size=W, act_fn="relu", device="cpu"):
        if dim_x != 3 or filter_size != W or act_fn != "relu":
            raise N.NdpError("the HIP kernels serve Neural_Prior(dim_x=3, filter_size=128, act_fn='relu') only")
        flat = torch.zeros(P_STRIDE, dtype=torch.float32)
        for l, (o, i) in enumerate(layer_shapes(), start=1):
            lin = torch.nn.Linear(i, o)
            flat[off_W(l):off_W(l) + o * i] = lin.weight.detach().reshape(-1)
            flat[off_b(l):off_b(l) + o] = lin.bias.detach()
        self.flat = flat.to(device)

    def to(self, device):
        self.flat = self.flat.to(device)
        return self

    def named_parameters(self):
        for l, (o, i) in enumerate(layer_shapes(), start=1):
            yield f"layer{l}.weight", self.flat[off_W(l):off_W(l) + o * i].view(o, i)
            yield f"layer{l}.bias", self.flat[off_b(l):off_b(l) + o]

    def __call__(self, x):
        """flow(x) [n,3] (inference: no activations kept)."""
        return nsfp_fwd(self.flat, x) - x


def _cap(n):
    return ops.cap(n)


def nsfp_fwd(params, x, save=False):
    """x [n,3] -> x + MLP(x); with save=True also the activation planes [8, cap, 128] for nsfp_bwd."""
    ops._chk(params, "params"); ops._chk(x, "x")
    n = x.shape[0]
    out = torch.empty_like(x)
    c = _cap(n)
    act = torch.empty(8 if save else 2, c, W, device=x.device, dtype=torch.float32)
    N.check(N.lib().ndp_nsfp_fwd(ops._p(params), ops._p(x), n, ops._p(out), ops._p(act) if save else None,
                                 None if save else ops._p(act), N.stream_ptr(x.device)), "ndp_nsfp_fwd")
    return (out, act) if save else out


def nsfp_bwd(params, x, act, g, n_part=None):
    """-> grads [PARAM_COUNT] of a scalar loss given g = dL/d(x + MLP(x)) [n,3].  `act` is consumed."""
    ops._chk(params, "params"); ops._chk(x, "x"); ops._chk(act, "act"); ops._chk(g, "g")
    n = x.shape[0]
    tiles = (n + N.TILE - 1) // N.TILE
    if n_part is None:
        n_part = min(tiles, 256)
    part = torch.empty(n_part, P_STRIDE, device=x.device, dtype=torch.float32)
    work = torch.empty(_cap(n), N.NHMAX, device=x.device, dtype=torch.float32)
    st = N.stream_ptr(x.device)
    N.check(N.lib().ndp_nsfp_bwd(ops._p(params), ops._p(x), n, ops._p(act), ops._p(g), ops._p(work), ops._p(part),
                                 n_part, P_STRIDE, st), "ndp_nsfp_bwd")
    grads = torch.empty(PARAM_COUNT, device=x.device, dtype=torch.float32)
    N.check(N.lib().ndp_grad_reduce(ops._p(part), n_part, P_STRIDE, PARAM_COUNT, ops._p(grads), st), "ndp_grad_reduce")
    return grads


def optimize_neural_SFlow(reg, visualize=False):
    """registration.py:470-540, same order of operations: model, centring, two randperms, Adam(lr), the early-stop
    rule on the Python float of the loss, final flow of ALL source points, `+ tgt_mean`.  Returns (warped, None)."""
    if visualize:
        raise NotImplementedError("mayavi visualisation is outside the hot path")
    config = reg.config
    dev = reg._dev()
    model = Neural_Prior().to(dev)                                              # :478
    reg.src_pcd = reg.src_pcd.to(dev).float()
    tgt_all = reg.tgt_pcd.to(dev).float()
    src_mean = reg.src_pcd.mean(dim=0, keepdim=True)                            # :484-487
    tgt_mean = tgt_all.mean(dim=0, keepdim=True)
    src_pcd = (reg.src_pcd - src_mean).contiguous()
    tgt_pcd = tgt_all - tgt_mean
    src = torch.randperm(src_pcd.shape[0])                                      # :494-497 (CPU RNG)
    tgt = torch.randperm(tgt_pcd.shape[0])
    s_sample = src_pcd[src[: config.samples].to(dev)].contiguous()
    t_sample = tgt_pcd[tgt[: config.samples].to(dev)].contiguous()
    params = model.flat
    m = torch.zeros(PARAM_COUNT, device=dev)
    v = torch.zeros(PARAM_COUNT, device=dev)
    break_counter, loss_prev, steps, L = 0, 1e6, 0, float("nan")   # iters == 0: the untrained warp, as upstream
    for i in range(config.iters):                                               # :511-529
        warped, act = nsfp_fwd(params, s_sample, save=True)
        loss, gx, _ = ops.chamfer_l1(warped, t_sample, 1e9)
        L = loss.item()
        if L < 1e-4:
            break
        if abs(loss_prev - L) < loss_prev * config.break_threshold_ratio:
            break_counter += 1
        if break_counter >= config.max_break_count:
            break
        loss_prev = L
        grads = nsfp_bwd(params, s_sample, act, gx)
        steps += 1
        ops.adam_step(params[:PARAM_COUNT], grads, m, v, steps, lr=config.lr)
    reg.last_nsfp = dict(iters=steps, loss=L)
    warped_pcd = nsfp_fwd(params, src_pcd) + tgt_mean                            # :534-536
    return warped_pcd, None
