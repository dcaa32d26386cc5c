"""Engine modes for the measurement tools: NDP_GEMM_MODE / NDP_NN_MODE in the environment -> Registration / BatchedEngine
keyword arguments.  (The package itself reads no environment variable: modes are constructor arguments.)"""
import os

if os.environ.get("NDP_HIP_LIB"):                    # an experiment build (tools/experiments/build_variant.sh) under a TOOL -- never under the
    from deformationpyramid_amd import _native as _N     # product or bench.py, which do not import this module
    _N.use_variant(os.environ["NDP_HIP_LIB"])


def from_env():
    kw = {}
    if os.environ.get("NDP_GEMM_MODE"):
        kw["gemm_mode"] = int(os.environ["NDP_GEMM_MODE"])
    if os.environ.get("NDP_NN_MODE"):
        nn = int(os.environ["NDP_NN_MODE"])
        if nn not in (0, 1, 2):
            raise SystemExit(f"NDP_NN_MODE must be 0, 1 or 2, got {nn}")
        kw["nn_mode"] = nn
    return kw
