"""Shared helpers for the parity tests (host-side only; no oracle / GPU imports here)."""
import numpy as np
import torch

from deformationpyramid_amd.nets import Deformation_Pyramid

VARIANTS = {
    "se3aa": dict(rotation_format="axis_angle", motion="SE3"),
    "sim3eu": dict(rotation_format="euler", motion="Sim3"),
    "sflow": dict(rotation_format="axis_angle", motion="sflow"),
    "se3eu": dict(rotation_format="euler", motion="SE3"),
    "sim3aa": dict(rotation_format="axis_angle", motion="Sim3"),
    "se3quat": dict(rotation_format="quaternion", motion="SE3"),
    "se36d": dict(rotation_format="6D", motion="SE3"),
    "sim3quat": dict(rotation_format="quaternion", motion="Sim3"),
}


def seeded_pyramid(seed, m=9, device="cpu", **kw):
    torch.manual_seed(seed)
    return Deformation_Pyramid(depth=3, width=128, device=device, k0=-8, m=m, **kw)


def scale_heads(pyr, level, factor):
    """Multiply every head weight and bias of one level (what make_golden.py does for F2)."""
    d = pyr.descs[level]
    with torch.no_grad():
        pyr.store[level, d.off_Wh:d.param_count] *= factor


def flat_from_named(desc, named):
    """{reference name: array} -> flat block."""
    flat = np.zeros(desc.param_count, dtype=np.float32)
    for name, off, shape in desc.named_slices():
        a = np.asarray(named[name], dtype=np.float32)
        assert a.shape == tuple(shape), (name, a.shape, shape)
        flat[off:off + a.size] = a.ravel()
    return flat


def wsum(pyr, level):
    d = pyr.descs[level]
    return float(pyr.store[level, :d.param_count].double().abs().sum())


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# The two arithmetic configurations every engine-level parity test runs in (same tolerances for both):
#   bitwise: level kernels on the fp32 MFMA (the oracle's fma chain) + the vector-pipe nearest-neighbour kernels;
#   split  : level kernels' 128x128 contractions as two-way fp16 splits on the fp16 MFMA (gemm_mode 7) + the one-pass
#            nearest-neighbour kernel with the distances on the bf16 matrix pipe (nn_mode 2) wherever its table fits LDS.
ARITH = ("bitwise", "split")


def engine_modes(arith, n_cap=None, nn_mode=None):
    """BatchedEngine keyword arguments of an arithmetic configuration.  nn_mode: force a shape (None: matrix kernel in
    `split` when n_cap sources fit it, engine's choice otherwise)."""
    from deformationpyramid_amd import _native as N
    from deformationpyramid_amd.ops import cap
    kw = dict(gemm_mode=7 if arith == "split" else 0)
    if nn_mode is not None:
        kw["nn_mode"] = nn_mode
    elif arith == "split" and n_cap is not None and N.lib().ndp_engine_nn_matrix_fits(cap(max(n_cap, 1))):
        kw["nn_mode"] = 2
    return kw


def registration_modes(arith):
    """Registration keyword arguments of an arithmetic configuration."""
    return dict(gemm_mode=7, nn_matrix=True) if arith == "split" else dict(gemm_mode=0, nn_matrix=False)


# width / depth other than the shipped 128 / 3 (fixture F16, tests/golden/make_golden.py: F16_SHAPES)
GENERIC_SHAPES = {
    "w64d2_se3aa": dict(width=64, depth=2, rotation_format="axis_angle", motion="SE3"),
    "w256d4_sim3eu": dict(width=256, depth=4, rotation_format="euler", motion="Sim3"),
    "w32d1_sflow": dict(width=32, depth=1, rotation_format="axis_angle", motion="sflow"),
    "w100d3_se3quat_nr": dict(width=100, depth=3, rotation_format="quaternion", motion="SE3", nonrigidity_est=True),
}


def generic_pyramid(seed, tag=None, m=5, device="cpu", **kw):
    torch.manual_seed(seed)
    kw = dict(GENERIC_SHAPES[tag], **kw) if tag else kw
    return Deformation_Pyramid(device=device, k0=-8, m=m, **kw)
