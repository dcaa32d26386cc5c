"""ctypes binding of the CPU oracle (oracle/ndp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See ndp_oracle.h for what it restates.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_double_p = ctypes.POINTER(ctypes.c_double)


class CLayerDesc(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("n_hidden", ctypes.c_int), ("motion", ctypes.c_int),
                ("rotfmt", ctypes.c_int), ("nonrigidity", ctypes.c_int), ("mlp_scale", ctypes.c_float)]


class COptCfg(ctypes.Structure):
    _fields_ = [("m", ctypes.c_int), ("k0", ctypes.c_int), ("iters", ctypes.c_int),
                ("max_break_count", ctypes.c_int), ("break_threshold_ratio", ctypes.c_double),
                ("lr", ctypes.c_double), ("w_cd", ctypes.c_float), ("trunc", ctypes.c_float),
                ("w_reg", ctypes.c_float), ("early_stop", ctypes.c_int)]


_MOTION = {"SE3": 0, "Sim3": 1, "sflow": 2}
_ROT = {"axis_angle": 0, "euler": 1, "quaternion": 2, "6D": 3}


def make_desc(width=128, n_hidden=2, motion="SE3", rotfmt="axis_angle", nonrigidity=False, mlp_scale=0.001):
    return CLayerDesc(width, n_hidden, _MOTION[motion], _ROT.get(rotfmt, 0), int(bool(nonrigidity)), mlp_scale)


def build(force=False):
    so = os.path.join(_HERE, "libndp_oracle.so")
    src = os.path.join(_HERE, "ndp_oracle.c")
    stale = os.environ.get("NDP_REBUILD") == "1" and os.path.exists(so) and os.path.getmtime(so) < os.path.getmtime(src)
    if force or not os.path.exists(so) or stale:
        tmp = f"{so}.{os.getpid()}.tmp"               # same flags as oracle/Makefile; atomic rename for concurrent ranks
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c11",
                               "-shared", "-o", tmp, src, "-lm"])
        os.replace(tmp, so)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.ndp_o_chamfer.restype = ctypes.c_float
        L.ndp_o_landmark.restype = ctypes.c_float
        L.ndp_o_stop_check.restype = ctypes.c_int
        L.ndp_o_optimize.restype = ctypes.c_int
        L.ndp_o_nsfp_optimize.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_float_p)


def param_count(desc):
    nrot = 0 if desc.motion == 2 else {2: 4, 3: 6}.get(desc.rotfmt, 3)
    nh = nrot + (1 if desc.motion == 1 else 0) + 3 + (1 if desc.nonrigidity else 0)
    W = desc.width
    return W * 7 + desc.n_hidden * (W * W + W) + nh * (W + 1)


def level_fwd(desc, params, level, k0, x, nthreads=1, want_nonrig=False):
    params, pp = _f(params)
    x, xp = _f(x)
    n = x.shape[0]
    out = np.empty_like(x)
    nr = np.zeros(n, dtype=np.float32)
    lib().ndp_o_level_fwd(ctypes.byref(desc), pp, int(level), int(k0), xp, n,
                          out.ctypes.data_as(c_float_p), nr.ctypes.data_as(c_float_p), int(nthreads))
    return (out, nr) if want_nonrig else out


def level_bwd(desc, params, level, k0, x, g, g_nr=None, nthreads=1):
    params, pp = _f(params)
    x, xp = _f(x)
    g, gp = _f(g)
    grads = np.zeros(param_count(desc), dtype=np.float32)
    gnp = None
    if g_nr is not None:
        g_nr, gnp = _f(g_nr)
    lib().ndp_o_level_bwd(ctypes.byref(desc), pp, int(level), int(k0), xp, x.shape[0], gp, gnp,
                          grads.ctypes.data_as(c_float_p), int(nthreads))
    return grads


def pyramid_fwd(descs, k0, params_all, x, nthreads=1):
    arr = (CLayerDesc * len(descs))(*descs)
    params_all, pp = _f(params_all)
    x, xp = _f(x)
    out = np.empty_like(x)
    lib().ndp_o_pyramid_fwd(arr, len(descs), int(k0), pp, xp, x.shape[0], out.ctypes.data_as(c_float_p), int(nthreads))
    return out


def chamfer(x, y, trunc=1e9, want_grad=True, nthreads=1):
    x, xp = _f(x)
    y, yp = _f(y)
    S, T = x.shape[0], y.shape[0]
    d2x = np.empty(S, np.float32); d2y = np.empty(T, np.float32)
    ix = np.empty(S, np.int32); iy = np.empty(T, np.int32)
    gx = np.zeros_like(x)
    loss = lib().ndp_o_chamfer(xp, S, yp, T, ctypes.c_float(trunc), d2x.ctypes.data_as(c_float_p),
                               ix.ctypes.data_as(c_int_p), d2y.ctypes.data_as(c_float_p),
                               iy.ctypes.data_as(c_int_p), gx.ctypes.data_as(c_float_p) if want_grad else None,
                               int(nthreads))
    return dict(loss=np.float32(loss), d2x=d2x, idx_x=ix, d2y=d2y, idx_y=iy, gx=gx)


def landmark(x, t):
    x, xp = _f(x)
    t, tp = _f(t)
    gx = np.zeros_like(x)
    loss = lib().ndp_o_landmark(xp, tp, x.shape[0], gx.ctypes.data_as(c_float_p))
    return np.float32(loss), gx


def adam(p, g, m, v, t, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
    """In-place on float32 contiguous arrays p, m, v."""
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    g, gp = _f(g)
    lib().ndp_o_adam(p.ctypes.data_as(c_float_p), gp, m.ctypes.data_as(c_float_p), v.ctypes.data_as(c_float_p),
                     p.size, int(t), ctypes.c_double(lr), ctypes.c_double(b1), ctypes.c_double(b2), ctypes.c_double(eps))


def stop_trace(losses, max_break_count=15, ratio=0.001):
    """Feed a loss sequence to the early-stop rule; return the index at which the level breaks
    (len(losses) if it never does) and the final (break_counter, loss_prev)."""
    bc = ctypes.c_int(0)
    lp = ctypes.c_double(1e6)
    for i, L in enumerate(losses):
        if lib().ndp_o_stop_check(ctypes.c_double(float(L)), ctypes.byref(bc), ctypes.byref(lp),
                                  int(max_break_count), ctypes.c_double(ratio)):
            return i, bc.value, lp.value
    return len(losses), bc.value, lp.value


def optimize(descs, params_all, pts, K, S, ldmk_t, tgt, *, k0=-8, iters=500, max_break_count=15,
             ratio=0.001, lr=0.01, w_cd=1.0, trunc=1e9, w_reg=0.0, early_stop=True, nthreads=1,
             trace_cap=8192):
    """Runs ndp_o_optimize.  Returns dict(params_all, pts, iters_per_level, loss_trace, steps)."""
    m = len(descs)
    arr = (CLayerDesc * m)(*descs)
    cfg = COptCfg(m, k0, iters, max_break_count, ratio, lr, w_cd, trunc, w_reg, int(bool(early_stop)))
    params_all = np.array(params_all, dtype=np.float32, order="C", copy=True)
    pts = np.array(pts, dtype=np.float32, order="C", copy=True).reshape(-1, 3)
    assert pts.shape[0] == K + S
    tgt, tp = _f(tgt if tgt is not None else np.zeros((0, 3), np.float32))
    lt, ltp = _f(ldmk_t if ldmk_t is not None else np.zeros((0, 3), np.float32))
    ipl = np.zeros(m, dtype=np.int32)
    trace = np.zeros(trace_cap, dtype=np.float64)
    steps = lib().ndp_o_optimize(arr, ctypes.byref(cfg), params_all.ctypes.data_as(c_float_p),
                                 pts.ctypes.data_as(c_float_p), int(K), int(S), ltp, tp, tgt.shape[0],
                                 ipl.ctypes.data_as(c_int_p), trace.ctypes.data_as(c_double_p), trace_cap,
                                 int(nthreads))
    return dict(params_all=params_all, pts=pts, iters_per_level=ipl,
                loss_trace=trace[:int(ipl.sum())], steps=steps)


# ------------------------------------------------------------------ NSFP baseline
NSFP_P = 128 * 4 + 7 * (128 * 128 + 128) + 3 * 128 + 3


def nsfp_fwd(params, x, nthreads=1):
    params, pp = _f(params)
    x, xp = _f(x)
    out = np.empty_like(x)
    lib().ndp_o_nsfp_fwd(pp, xp, x.shape[0], out.ctypes.data_as(c_float_p), int(nthreads))
    return out


def nsfp_bwd(params, x, g):
    params, pp = _f(params)
    x, xp = _f(x)
    g, gp = _f(g)
    grads = np.zeros(NSFP_P, dtype=np.float32)
    lib().ndp_o_nsfp_bwd(pp, xp, x.shape[0], gp, grads.ctypes.data_as(c_float_p))
    return grads


def nsfp_optimize(params, s_sample, t_sample, iters, max_break_count=70, ratio=0.001, lr=0.01, early_stop=True,
                  nthreads=1, trace_cap=8192):
    params = np.array(params[:NSFP_P], dtype=np.float32, order="C", copy=True)
    s, sp = _f(s_sample)
    t, tp = _f(t_sample)
    warped = np.empty_like(s)
    trace = np.zeros(trace_cap, dtype=np.float64)
    steps = lib().ndp_o_nsfp_optimize(params.ctypes.data_as(c_float_p), sp, s.shape[0], tp, t.shape[0], int(iters),
                                      int(max_break_count), ctypes.c_double(ratio), ctypes.c_double(lr),
                                      int(bool(early_stop)), warped.ctypes.data_as(c_float_p),
                                      trace.ctypes.data_as(c_double_p), trace_cap, int(nthreads))
    n_eval = steps + 1 if steps < iters else steps
    return dict(params=params, warped=warped, steps=steps, loss_trace=trace[:min(n_eval, trace_cap)])
