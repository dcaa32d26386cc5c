"""torch-tensor wrappers around the single-pair operators of libndp_hip.so.

Torch is plumbing here: it owns device memory and the stream; all arithmetic is in the HIP
library.  Every function requires CUDA(HIP) float32 contiguous tensors and raises otherwise --
there is no CPU / eager fallback.
"""
import ctypes
import math

import torch

from . import _native as N
from .layout import LayerDesc

TILE = N.TILE


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise N.NdpError(f"{name}: the HIP path needs a tensor on the GPU (got {getattr(t, 'device', type(t))}); "
                         "there is no CPU fallback")
    if t.dtype != dtype:
        raise N.NdpError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise N.NdpError(f"{name}: tensor must be contiguous")
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def cap(n):
    return max(TILE, (n + TILE - 1) // TILE * TILE)


def level_fwd(desc: LayerDesc, params, level, k0, x, save=False, want_nonrig=False):
    """x [n,3] -> x_out [n,3]; with save also (act [depth,cap,width], heads [cap,24]; [3,cap,128] for the shipped 128 / 3); with
    want_nonrig the gate values [n] of a level that carries the nonrigidity head are appended.  width / depth other than 128 / 3 run on
    the generic fp32 kernels (csrc/ndp_generic.inc), bitwise the oracle's chain."""
    _chk(params, "params"); _chk(x, "x")
    n = x.shape[0]
    out = torch.empty_like(x)
    act = heads = nr = None
    if save:
        c = cap(n)
        act = torch.empty(desc.n_hidden + 1, c, desc.width, device=x.device, dtype=torch.float32)
        heads = torch.empty(c, N.HROW, device=x.device, dtype=torch.float32)
    if want_nonrig and desc.nonrigidity:
        nr = torch.empty(n, device=x.device, dtype=torch.float32)
    cd = desc.c_struct()
    N.check(N.lib().ndp_level_fwd(ctypes.byref(cd), _p(params), int(level), int(k0), _p(x), n, _p(out),
                                  _p(act), _p(heads), _p(nr), N.stream_ptr(x.device)), "ndp_level_fwd")
    res = (out, act, heads) if save else (out,)
    if want_nonrig:
        res = res + (nr,)
    return res if len(res) > 1 else res[0]


def level_bwd(desc: LayerDesc, params, level, k0, x, act, heads, g, n_part=None, g_nr=None):
    """-> grads [P] (partials folded in index order on the device).  `act` is consumed."""
    _chk(params, "params"); _chk(x, "x"); _chk(act, "act"); _chk(heads, "heads"); _chk(g, "g")
    if g_nr is not None:
        _chk(g_nr, "g_nr")
    n = x.shape[0]
    if tuple(act.shape) != (desc.n_hidden + 1, cap(n), desc.width) or tuple(heads.shape) != (cap(n), N.HROW):
        raise N.NdpError(f"level_bwd: act must be [{desc.n_hidden + 1}, {cap(n)}, {desc.width}] and heads [{cap(n)}, {N.HROW}] as level_fwd(save=True) left them")
    P = desc.param_count
    stride = (P + 3) // 4 * 4
    tiles = (n + TILE - 1) // TILE
    if n_part is None:
        n_part = min(tiles, 256)
    part = torch.empty(n_part, stride, device=x.device, dtype=torch.float32)
    work = torch.empty(cap(n), N.NHMAX, device=x.device, dtype=torch.float32)
    cd = desc.c_struct()
    st = N.stream_ptr(x.device)
    N.check(N.lib().ndp_level_bwd(ctypes.byref(cd), _p(params), int(level), int(k0), _p(x), n, _p(act), _p(heads),
                                  _p(g), _p(g_nr), _p(work), _p(part), n_part, stride, st), "ndp_level_bwd")
    grads = torch.empty(P, device=x.device, dtype=torch.float32)
    N.check(N.lib().ndp_grad_reduce(_p(part), n_part, stride, P, _p(grads), st), "ndp_grad_reduce")
    return grads


def pyramid_fwd(desc: LayerDesc, m, k0, store, x):
    """store [m, p_stride] (level l in row l) ; x [n,3] -> [n,3]."""
    _chk(store, "store"); _chk(x, "x")
    out = torch.empty_like(x)
    cd = desc.c_struct()
    N.check(N.lib().ndp_pyramid_fwd(ctypes.byref(cd), int(m), int(k0), _p(store), store.stride(0), _p(x), x.shape[0],
                                    _p(out), N.stream_ptr(x.device)), "ndp_pyramid_fwd")
    return out


def pyramid_fwd_batch(desc: LayerDesc, m, k0, jobs, device=None, split=False, tiles=None):
    """jobs: [(store [m,p_stride], x [n,3], shift_in [>=3] | None, shift_out [>=3] | None)] -> [x_out [n,3]]: every
    cloud through its whole pyramid in ONE launch per 32 jobs; x_out = pyramid(x - shift_in) + shift_out.
    split: the engine's fp16-split arithmetic (k_pyramid_fwd8) instead of the fp32 MFMA (bitwise the level chain).
    tiles (split only): 64-point tiles per workgroup, 1..8 (default 4; the batched engine passes 8: half the weight prologues)."""
    if not jobs:
        return []
    arr = (N.WarpJob * len(jobs))()
    outs = []
    stride = None
    for q, (store, x, s_in, s_out) in zip(arr, jobs):
        _chk(store, "store"); _chk(x, "x")
        if s_in is not None:
            _chk(s_in, "shift_in")
        if s_out is not None:
            _chk(s_out, "shift_out")
        if stride is None:
            stride = store.stride(0)
        elif stride != store.stride(0):
            raise N.NdpError("pyramid_fwd_batch: all stores must share one row stride")
        out = torch.empty_like(x)
        outs.append(out)
        q.params, q.x, q.x_out = store.data_ptr(), x.data_ptr(), out.data_ptr()
        q.shift_in = s_in.data_ptr() if s_in is not None else None
        q.shift_out = s_out.data_ptr() if s_out is not None else None
        q.n = x.shape[0]
    cd = desc.c_struct()
    dev = device if device is not None else jobs[0][1].device
    if split and tiles is not None:
        N.check(N.lib().ndp_pyramid_fwd_batch_split_tiles(ctypes.byref(cd), int(m), int(k0), int(stride), arr, len(jobs), int(tiles),
                                                          N.stream_ptr(dev)), "ndp_pyramid_fwd_batch_split_tiles")
        return outs
    fn = N.lib().ndp_pyramid_fwd_batch_split if split else N.lib().ndp_pyramid_fwd_batch
    N.check(fn(ctypes.byref(cd), int(m), int(k0), int(stride), arr, len(jobs), N.stream_ptr(dev)), "ndp_pyramid_fwd_batch")
    return outs


def pair_means(src, tgt, out=None):
    """-> means [8] on the device: [0:3] mean of src, [4:7] mean of tgt (registration.py:150-153)."""
    _chk(src, "src"); _chk(tgt, "tgt")
    if out is None:
        out = torch.empty(8, device=src.device, dtype=torch.float32)
    N.check(N.lib().ndp_pair_means(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(out), N.stream_ptr(src.device)),
            "ndp_pair_means")
    return out


def chamfer_nn(x, y):
    _chk(x, "x"); _chk(y, "y")
    S, T = x.shape[0], y.shape[0]
    d2x = torch.empty(S, device=x.device); d2y = torch.empty(T, device=x.device)
    ix = torch.empty(S, device=x.device, dtype=torch.int32); iy = torch.empty(T, device=x.device, dtype=torch.int32)
    N.check(N.lib().ndp_chamfer_nn_fwd(_p(x), S, _p(y), T, _p(d2x), _p(ix), _p(d2y), _p(iy), N.stream_ptr(x.device)),
            "ndp_chamfer_nn_fwd")
    return d2x, ix, d2y, iy


def chamfer_nn_onepass(x, y, matrix=False):
    """Same result as chamfer_nn from one pass over the S x T distances (the engine's per-tick kernels as an operator);
    matrix: the variant that forms the distances on the bf16 matrix pipe and re-evaluates the candidates exactly (engine nn_mode 2)."""
    _chk(x, "x"); _chk(y, "y")
    S, T = x.shape[0], y.shape[0]
    nr = ctypes.c_longlong()
    N.check(N.lib().ndp_engine_nn_workspace(cap(S), cap(T), ctypes.byref(nr)), "ndp_engine_nn_workspace")
    ws_row = torch.empty(nr.value, device=x.device)
    d2x = torch.empty(S, device=x.device); d2y = torch.empty(T, device=x.device)
    ix = torch.empty(S, device=x.device, dtype=torch.int32); iy = torch.empty(T, device=x.device, dtype=torch.int32)
    fn = N.lib().ndp_chamfer_nn_matrix if matrix else N.lib().ndp_chamfer_nn_onepass
    N.check(fn(_p(x), S, _p(y), T, _p(d2x), _p(ix), _p(d2y), _p(iy), _p(ws_row), N.stream_ptr(x.device)),
            "ndp_chamfer_nn_matrix" if matrix else "ndp_chamfer_nn_onepass")
    return d2x, ix, d2y, iy


def chamfer_l1(x, y, trunc, nn=None, want_grad=True, point_sum=False):
    """-> (loss [1], gx [S,3] | None, nn tuple).  point_sum: point_reduction="sum" of loss.py:233-235."""
    if nn is None:
        nn = chamfer_nn(x, y)
    d2x, ix, d2y, iy = nn
    loss = torch.empty(1, device=x.device)
    gx = torch.empty_like(x) if want_grad else None
    N.check(N.lib().ndp_chamfer_l1_bwd(_p(x), x.shape[0], _p(y), y.shape[0], float(trunc), _p(d2x), _p(ix), _p(d2y),
                                       _p(iy), _p(loss), _p(gx), 1 if point_sum else 0, N.stream_ptr(x.device)), "ndp_chamfer_l1_bwd")
    return loss, gx, nn


def landmark_mse(x, t):
    _chk(x, "x"); _chk(t, "t")
    loss = torch.empty(1, device=x.device)
    gx = torch.empty_like(x)
    N.check(N.lib().ndp_landmark_mse_fwd_bwd(_p(x), _p(t), x.shape[0], _p(loss), _p(gx), N.stream_ptr(x.device)),
            "ndp_landmark_mse_fwd_bwd")
    return loss, gx


def adam_scalars(t, lr=0.01, b1=0.9, b2=0.999):
    """(neg_step, bc2_sqrt) of torch.optim.Adam's single-tensor path, computed in Python doubles."""
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    return -(lr / bc1), math.sqrt(bc2)


def adam_step(p, g, m, v, t, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
    for a, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(a, nm)
    ns, bc2s = adam_scalars(t, lr, b1, b2)
    N.check(N.lib().ndp_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), 1 - b1, b2, 1 - b2, ns, bc2s, eps,
                                  N.stream_ptr(p.device)), "ndp_adam_step")


# ------------------------------------------------------------------------------ autograd wrappers
class _LevelWarpFn(torch.autograd.Function):
    """NDPLayer.forward (nets.py:111-140) for callers that own their optimisation loop
    (shape_transfer.py:116-157).  x is treated as detached, as on the reference's hot path.
    Returns (x', nonrigidity) -- the gate tensor is empty for levels without the nonrigidity head."""

    @staticmethod
    def forward(ctx, x, layer, *params):
        need = any(p.requires_grad for p in params)
        flat = layer.flat.detach()
        xd = x.detach().contiguous()
        gate = layer.desc.nonrigidity
        if need:
            res = level_fwd(layer.desc, flat, layer.level, layer.k0, xd, save=True, want_nonrig=True)
            out, act, heads, nr = res
            ctx.save_for_backward(xd, act, heads)
            ctx.layer = layer
        else:
            out, nr = level_fwd(layer.desc, flat, layer.level, layer.k0, xd, want_nonrig=True)
        if not gate:
            nr = torch.empty(0, device=x.device)
            ctx.mark_non_differentiable(nr)
        return out, nr

    @staticmethod
    def backward(ctx, g, g_nr):
        x, act, heads = ctx.saved_tensors
        layer = ctx.layer
        gnr = g_nr.contiguous() if (layer.desc.nonrigidity and g_nr is not None) else None
        grads = level_bwd(layer.desc, layer.flat.detach(), layer.level, layer.k0, x, act, heads, g.contiguous(), g_nr=gnr)
        outs = []
        for name, off, shape in layer.desc.named_slices():
            n = 1
            for s in shape:
                n *= s
            outs.append(grads[off:off + n].view(shape))
        return (None, None) + tuple(outs)


def level_warp(layer, x):
    """-> (x', nonrigidity | None).  Squeeze semantics of nets.py:140 are not reproduced (n >= 2)."""
    if x.dim() != 2 or x.shape[-1] != 3:
        raise ValueError("expected points of shape [n, 3]")
    params = tuple(layer.parameters())
    out, nr = _LevelWarpFn.apply(x.float(), layer, *params)
    return out, (nr if layer.desc.nonrigidity else None)


class _ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, trunc, point_sum=False):
        xd, yd = x.detach().contiguous(), y.detach().contiguous()
        loss, gx, _ = chamfer_l1(xd, yd, trunc, want_grad=x.requires_grad, point_sum=point_sum)
        ctx.save_for_backward(gx if gx is not None else torch.empty(0, device=x.device))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (gx,) = ctx.saved_tensors
        return (gx * g if gx.numel() else None), None, None, None


def chamfer_distance(x, y, trunc, point_sum=False):
    """Differentiable (wrt x) truncated L1 Chamfer of two [n,3] clouds."""
    return _ChamferFn.apply(x, y, float(trunc), bool(point_sum))


def flow_metrics(flow, flow_gt, overlap=None):
    """Device-side sums behind compute_flow_metrics -> [3,5] float64 on the host: per subset {all, overlap, ~overlap}
    {sum err, #AccS, #AccR, #outlier, #points}."""
    _chk(flow, "flow"); _chk(flow_gt, "flow_gt")
    n = flow.shape[0]
    ov = None
    if overlap is not None:
        ov = overlap.to(device=flow.device, dtype=torch.uint8).contiguous()
    out = torch.empty(15, device=flow.device, dtype=torch.float64)
    N.check(N.lib().ndp_flow_metrics(_p(flow), _p(flow_gt), _p(ov), n, _p(out), N.stream_ptr(flow.device)), "ndp_flow_metrics")
    return out.cpu().view(3, 5)
