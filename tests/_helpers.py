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
