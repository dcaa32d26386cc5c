"""Deformation pyramid: host-side mirror of the reference's model/nets.py interface.

Public surface kept from the reference (/root/reference/model/nets.py:10-62):

    Deformation_Pyramid(depth, width, device, k0, m, rotation_format,
                        nonrigidity_est=False, motion='SE3')
        .pyramid[i]            per-level nn.Module exposing .parameters()/.named_parameters()
        .n_hierarchy
        .warp(x, max_level=None, min_level=0) -> (x', {level: (x_level, nonrigidity|None)})
        .gradient_setup(optimized_level)

What is different underneath: every level's parameters live in ONE flat float32 block
(layout.py / include/ndp_types.h) and all m blocks are rows of a single [m, Pmax] tensor in
HBM; the per-name nn.Parameters are views into it.  `warp` runs the hand-written HIP level
kernels (csrc/) through torch.autograd.Function wrappers, so callers that own their Adam loop
(shape_transfer.py:116-157 style) keep working.  There is no PyTorch fallback: on a CUDA/HIP
device the native library must be loaded; on CPU tensors `warp` raises.

Initialisation replays the reference's RNG consumption exactly (nets.py:75-109,180-183):
default nn.Linear init for weights and biases in module-registration order, then
xavier_uniform_ on every matrix, for all m levels up front, on the CPU generator.
"""
import math

import torch
import torch.nn as nn

from .layout import LayerDesc


_KAIMING_GAIN = nn.init.calculate_gain("leaky_relu", math.sqrt(5))


def _init_level_flat(desc: LayerDesc, depth: int, out: torch.Tensor = None) -> torch.Tensor:
    """Fill one level's flat block consuming the CPU generator exactly as NDPLayer.__init__ +
    _reset_parameters do (nets.py:67-109,180-183): per nn.Linear, in module-registration order,
    the default init (kaiming_uniform_(a=sqrt(5)) on the weight, then uniform bias); afterwards
    xavier_uniform_ on every matrix in the same order.  The draws go straight into views of the
    flat block with the bounds torch.nn.init computes -- no nn.Module objects are built.
    Call under torch.no_grad()."""
    assert depth - 1 == desc.n_hidden
    flat = out if out is not None else torch.empty(desc.param_count, dtype=torch.float32)
    slices = desc.named_slices()                       # (weight, bias) pairs in registration order
    mats = []
    for (wname, woff, wshape), (bname, boff, bshape) in zip(slices[0::2], slices[1::2]):
        fan_out, fan_in = wshape
        w = flat[woff:woff + fan_out * fan_in]
        # nn.Linear.reset_parameters: kaiming_uniform_(weight, a=sqrt(5)); bias ~ U(-1/sqrt(fan_in), +)
        kb = math.sqrt(3.0) * (_KAIMING_GAIN / math.sqrt(fan_in))
        w.uniform_(-kb, kb)
        bb = 1 / math.sqrt(fan_in)
        flat[boff:boff + fan_out].uniform_(-bb, bb)
        mats.append((w, math.sqrt(3.0) * (1.0 * math.sqrt(2.0 / float(fan_in + fan_out)))))
    for w, a in mats:                                                    # nets.py:180-183 xavier_uniform_
        w.uniform_(-a, a)
    return flat


def _draw_ops(descs, depth, p_stride):
    """The reference's RNG consumption as a flat list of (count, lo, hi, offset) draw ops over the
    [m, p_stride] store; offset -1 = values the reference overwrites later (only advance the generator)."""
    ops = []
    for lvl, d in enumerate(descs):
        assert depth - 1 == d.n_hidden
        base = lvl * p_stride
        slices = d.named_slices()
        xav = []
        for (wname, woff, wshape), (bname, boff, bshape) in zip(slices[0::2], slices[1::2]):
            fan_out, fan_in = wshape
            ops.append((fan_out * fan_in, 0.0, 1.0, -1))                  # kaiming_uniform_ draw, overwritten by xavier
            bb = 1 / math.sqrt(fan_in)
            ops.append((fan_out, -bb, bb, base + boff))                   # nn.Linear bias init
            a = math.sqrt(3.0) * (1.0 * math.sqrt(2.0 / float(fan_in + fan_out)))
            xav.append((fan_out * fan_in, -a, a, base + woff))            # xavier_uniform_
        ops += xav
    return ops


_NATIVE_RNG = {"checked": False, "ok": False}


def _native_rng_ok():
    """One-time self check of the native generator replay against torch itself (same draws, same bits,
    same generator state afterwards).  If it ever disagrees the torch-call replay is used instead."""
    if not _NATIVE_RNG["checked"]:
        _NATIVE_RNG["checked"] = True
        saved = torch.get_rng_state()
        try:
            from . import _native as N
            torch.manual_seed(987654321)
            a = torch.empty(700).uniform_(-1.5, -0.25)
            torch.empty(333).uniform_(0.0, 1.0)
            b = torch.empty(1300).uniform_(-0.0883883461356163, 0.0883883461356163)
            r1 = torch.randperm(17)
            torch.manual_seed(987654321)
            out = torch.zeros(2000)
            N.rng_replay([(700, -1.5, -0.25, 0), (333, 0.0, 1.0, -1), (1300, -0.0883883461356163, 0.0883883461356163, 700)],
                         out)
            r2 = torch.randperm(17)
            ok = bool(torch.equal(out[:700], a) and torch.equal(out[700:], b) and torch.equal(r1, r2))
            # the native randperm replay (ndp_pair_init's permutation prefixes, ndp_rng_skip's end state) against torch.randperm itself:
            # a torch release that changes randperm_cpu sends every caller back to the torch-call path instead of silently changing the samples
            import ctypes
            torch.manual_seed(13579)
            w1, w2 = torch.randperm(41), torch.randperm(7)
            end = torch.get_rng_state().clone()
            torch.manual_seed(13579)
            st = torch.get_rng_state().clone()
            L = N.host_lib()
            ops0 = N.make_draw_ops([])
            hi = torch.zeros(64, dtype=torch.int32)
            scratch = torch.zeros(64, dtype=torch.int64)
            dummy = torch.zeros(8)
            rc = L.ndp_pair_init(ctypes.c_void_p(st.data_ptr()), st.numel(), ops0, 0, ctypes.c_void_p(dummy.data_ptr()), 41, 7, 20,
                                 ctypes.c_void_p(hi.data_ptr()), ctypes.c_void_p(hi[20:].data_ptr()), ctypes.c_void_p(scratch.data_ptr()))
            ok = ok and rc == 0 and torch.equal(hi[:20].long(), w1[:20]) and torch.equal(hi[20:27].long(), w2)
            ok = ok and L.ndp_rng_skip(ctypes.c_void_p(st.data_ptr()), st.numel(), L.ndp_pair_draws(ops0, 0, 41, 7)) == 0 and torch.equal(st, end)
            _NATIVE_RNG["ok"] = bool(ok)
        except Exception:
            _NATIVE_RNG["ok"] = False
        finally:
            torch.set_rng_state(saved)                    # (also when the check itself raised: the caller's generator is not ours to reseed)
    return _NATIVE_RNG["ok"]


_OPS_CACHE = {}


def init_pyramid_store(descs, depth, p_stride, native=True, out=None):
    """[m, p_stride] CPU tensor with every level initialised in order (the RNG replay of
    Deformation_Pyramid.__init__, nets.py:20-30) -- the part of the constructor the batched
    registration path needs, without the per-name Parameter views.  With native=True the draws
    are produced by libndp_host.so (bit-identical to torch's generator, ~4x faster)."""
    descs = list(descs)
    store = out if out is not None else torch.empty(len(descs), p_stride, dtype=torch.float32)
    assert store.shape == (len(descs), p_stride) and store.dtype == torch.float32 and store.is_contiguous()
    if native and _native_rng_ok():
        from . import _native as N
        key = (tuple(descs), depth, p_stride)
        if key not in _OPS_CACHE:
            _OPS_CACHE[key] = N.make_draw_ops(_draw_ops(descs, depth, p_stride))
        for i, d in enumerate(descs):
            store[i, d.param_count:] = 0.0
        N.rng_replay(_OPS_CACHE[key], store)
        return store
    with torch.no_grad():
        for i, d in enumerate(descs):
            _init_level_flat(d, depth, out=store[i, :d.param_count])
            store[i, d.param_count:] = 0.0                # padding reads as zero
    return store


class NDPLevel(nn.Module):
    """One pyramid level.  Parameters are views into a row of the pyramid's flat store and
    carry the reference's names (input.0.weight, mlp.pts_linears.0.weight, rot_brach.weight, ...)."""

    def __init__(self, desc: LayerDesc, level: int, k0: int, flat_row: torch.Tensor):
        super().__init__()
        self.desc = desc
        self.level = level
        self.k0 = k0
        self.m = level + 1                 # reference attribute name (nets.py:71)
        self.mlp_scale = desc.mlp_scale
        self.flat = flat_row               # [P] view, shares storage with the store
        self._names = []
        for name, off, shape in desc.named_slices():
            n = 1
            for s in shape:
                n *= s
            p = nn.Parameter(flat_row[off:off + n].view(shape), requires_grad=True)
            reg = name.replace(".", "__")
            self.register_parameter(reg, p)
            self._names.append((name, reg))

    def named_parameters(self, *a, **k):
        # expose the reference's dotted names
        for name, reg in self._names:
            yield name, getattr(self, reg)

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def forward(self, x):
        from .ops import level_warp
        return level_warp(self, x)


class Deformation_Pyramid:
    def __init__(self, depth, width, device, k0, m, rotation_format, nonrigidity_est=False, motion='SE3'):
        assert motion in ["Sim3", "SE3", "sflow"]
        self.depth, self.width, self.k0 = depth, width, k0
        self.n_hierarchy = m
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.descs = [
            LayerDesc(width=width, n_hidden=depth - 1, motion=motion, rotfmt=rotation_format,
                      nonrigidity=bool(nonrigidity_est) and (i != 0))     # nets.py:26
            for i in range(m)
        ]
        self.pmax = max(d.param_count for d in self.descs) if m else 0
        self.p_stride = (self.pmax + 63) // 64 * 64          # rows stay 16-byte aligned for the kernels
        # all levels are initialised on the CPU generator first (nets.py:20-30), then moved
        store = init_pyramid_store(self.descs, depth, self.p_stride)
        self.store = store.to(self.device)
        self.pyramid = [NDPLevel(d, i, k0, self.store[i, :d.param_count]) for i, d in enumerate(self.descs)]

    def warp(self, x, max_level=None, min_level=0):
        if max_level is None:
            max_level = self.n_hierarchy - 1
        assert max_level < self.n_hierarchy, "more level than defined"
        data = {}
        for i in range(min_level, max_level + 1):
            x, nonrigidity = self.pyramid[i](x)
            data[i] = (x, nonrigidity)
        return x, data

    def gradient_setup(self, optimized_level):
        assert optimized_level < self.n_hierarchy, "more level than defined"
        for i in range(self.n_hierarchy):
            for param in self.pyramid[i].parameters():
                param.requires_grad = (i == optimized_level)
