"""Loader and ctypes signatures for libndp_hip.so (the C ABI of include/ndp_hip.h).

There is deliberately NO fallback: if the HIP library is missing or an entry point fails, the
caller gets an exception.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
import ctypes
import os
import shutil
import subprocess

from .layout import CLayerDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libndp_hip.so")
SOURCES = ["ndp_kernels.hip"]
HEADERS = ["ndp_device.h", "ndp_nerfies.inc", "ndp_ed.inc", "ndp_fwd_split.inc", "ndp_bwd_split.inc", "ndp_bwd_fused.inc", "ndp_tick_small.inc", "ndp_nn_matrix.inc", "ndp_generic.inc", os.path.join("..", "..", "include", "ndp_hip.h"), os.path.join("..", "..", "include", "ndp_types.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-pass-failed"]

NDP_MAX_LEVELS = 16
TILE = 64
NHMAX = 16
HROW = 24

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)


class NdpError(RuntimeError):
    pass


class PairGeom(ctypes.Structure):
    _fields_ = [("K", ctypes.c_int), ("S", ctypes.c_int), ("T", ctypes.c_int), ("pad", ctypes.c_int)]


class PairState(ctypes.Structure):
    _fields_ = [("level", ctypes.c_int), ("iter", ctypes.c_int), ("break_counter", ctypes.c_int),
                ("adam_t", ctypes.c_int), ("cur", ctypes.c_int), ("decision", ctypes.c_int),
                ("total_steps", ctypes.c_int), ("total_evals", ctypes.c_int), ("loss", ctypes.c_float),
                ("step_level", ctypes.c_int), ("step_t", ctypes.c_int), ("pad", ctypes.c_int),
                ("loss_prev", ctypes.c_double), ("evals_per_level", ctypes.c_int * NDP_MAX_LEVELS)]


class Engine(ctypes.Structure):
    _fields_ = [("desc", CLayerDesc), ("m", ctypes.c_int), ("k0", ctypes.c_int),
                ("P", ctypes.c_int), ("p_stride", ctypes.c_int),
                ("iters", ctypes.c_int), ("max_break_count", ctypes.c_int), ("early_stop", ctypes.c_int),
                ("B", ctypes.c_int), ("G", ctypes.c_int), ("n_cap", ctypes.c_int), ("t_cap", ctypes.c_int),
                ("break_threshold_ratio", ctypes.c_double),
                ("w_cd", ctypes.c_float), ("trunc", ctypes.c_float),
                ("adam_w1", ctypes.c_float), ("adam_b2", ctypes.c_float), ("adam_w2", ctypes.c_float),
                ("adam_eps", ctypes.c_float), ("w_reg", ctypes.c_float), ("pad_f", ctypes.c_float),
                ("geom", ctypes.c_void_p), ("state", ctypes.c_void_p), ("pts", ctypes.c_void_p),
                ("ldmk_t", ctypes.c_void_p), ("tgt", ctypes.c_void_p), ("params", ctypes.c_void_p),
                ("gpart", ctypes.c_void_p), ("adam_m", ctypes.c_void_p), ("adam_v", ctypes.c_void_p),
                ("act", ctypes.c_void_p), ("heads", ctypes.c_void_p),
                ("d2x", ctypes.c_void_p), ("idx_x", ctypes.c_void_p), ("d2y", ctypes.c_void_p),
                ("idx_y", ctypes.c_void_p), ("adam_tab", ctypes.c_void_p), ("dO", ctypes.c_void_p),
                ("nn_row", ctypes.c_void_p), ("nn_mode", ctypes.c_int), ("gemm_mode", ctypes.c_int),
                ("gmax", ctypes.c_void_p)]


class WarpJob(ctypes.Structure):
    _fields_ = [("params", ctypes.c_void_p), ("x", ctypes.c_void_p), ("x_out", ctypes.c_void_p),
                ("shift_in", ctypes.c_void_p), ("shift_out", ctypes.c_void_p), ("n", ctypes.c_int), ("pad", ctypes.c_int)]


class LoadJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("tgt", ctypes.c_void_p), ("perm_s", ctypes.c_void_p),
                ("perm_t", ctypes.c_void_p), ("ldmk_s", ctypes.c_void_p), ("ldmk_t", ctypes.c_void_p),
                ("params", ctypes.c_void_p), ("means", ctypes.c_void_p),
                ("slot", ctypes.c_int), ("K", ctypes.c_int), ("S", ctypes.c_int), ("T", ctypes.c_int),
                ("n_src", ctypes.c_int), ("n_tgt", ctypes.c_int)]


MAX_WARP_JOBS = 32
TICK_KERNELS = ("k_eng_fwd", "k_eng_nn", "k_eng_loss", "k_eng_bwd2", "k_eng_bwd1", "k_eng_update")   # one tick, launch order
MAX_LOAD_JOBS = 16


def source_id():
    """Hex digest of the sources the library is built from (kernel file, device header, the two ABI headers) and of
    the compile flags.  It is compiled into the library (-DNDP_BUILD_ID) and returned by ndp_build_id(): a library
    whose id differs from the tree's was built from other sources -- lib() rebuilds it or refuses to load it, because
    ndp_engine / ndp_load_job are passed BY VALUE into kernels and a stale layout would corrupt device memory."""
    import hashlib
    h = hashlib.sha256()
    try:
        for d in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, x)) for x in HEADERS]:
            with open(d, "rb") as f:
                h.update(f.read())
    except OSError:
        return None                      # a deployment without csrc/: the prebuilt library's embedded id stands (see _stale)
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def _built_id(path, tag=b"NDP_BUILD_ID"):
    """Build id of an existing library file (the tag string ndp_build_id() returns, found without loading it), or None."""
    import re
    try:
        with open(path, "rb") as f:
            hit = re.search(tag + rb"=([0-9a-f]{16})", f.read())
        return hit.group(1).decode() if hit else None
    except OSError:
        return None


def _stale():
    sid = source_id()
    if sid is None:                      # no sources to compare with: any library that carries an id is taken as built
        return _built_id(LIBPATH) is None
    return not os.path.exists(LIBPATH) or _built_id(LIBPATH) != sid


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> deformationpyramid_amd/lib/libndp_hip.so (in-tree).  Serialised by a file
    lock: N ranks that find a stale library build it once."""
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():
            return LIBPATH
        if source_id() is None:
            raise NdpError(f"{LIBPATH} is missing (or carries no build id) and this deployment has no csrc/ to build it from: "
                           "ship the prebuilt library with the package (its layout is checked against the Python structs by ndp_abi_sizes)")
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            raise NdpError("hipcc not found and libndp_hip.so is missing or stale")
        tmp = f"{LIBPATH}.{os.getpid()}.tmp"          # build aside + atomic rename: nobody ever sees a torn file
        cmd = [hipcc] + HIPCC_FLAGS + [f'-DNDP_BUILD_ID="{source_id()}"', "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIBPATH)
    return LIBPATH


_LIB = None
V = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
DP = ctypes.POINTER(CLayerDesc)

_SIGS = {
    "ndp_level_fwd": [DP, V, I, I, V, I, V, V, V, V, V],
    "ndp_level_bwd": [DP, V, I, I, V, I, V, V, V, V, V, V, I, I, V],
    "ndp_grad_reduce": [V, I, I, I, V, V],
    "ndp_pyramid_fwd": [DP, I, I, V, I, V, I, V, V],
    "ndp_pyramid_fwd_batch": [DP, I, I, I, ctypes.POINTER(WarpJob), I, V],
    "ndp_pyramid_fwd_batch_split": [DP, I, I, I, ctypes.POINTER(WarpJob), I, V],
    "ndp_pyramid_fwd_batch_split_tiles": [DP, I, I, I, ctypes.POINTER(WarpJob), I, I, V],
    "ndp_pair_means": [V, I, V, I, V, V],
    "ndp_nsfp_fwd": [V, V, I, V, V, V, V],
    "ndp_nsfp_bwd": [V, V, I, V, V, V, V, I, I, V],
    "ndp_ed_warp": [V, I, V, V, V, I, V, V, V, V, V],
    "ndp_ed_arap": [V, I, V, V, V, V, I, V, V],
    "ndp_ed_grad": [V, I, V, V, V, V, I, V, V, V, V, V, I, F, V, V],
    "ndp_nerfies_fwd": [V, V, I, c_float_p, V, V, V, V, I, V, V, V, V],
    "ndp_nerfies_bwd": [V, V, I, V, V, V, V, V, V, I, I, V],
    "ndp_chamfer_nn_fwd": [V, I, V, I, V, V, V, V, V],
    "ndp_chamfer_nn_onepass": [V, I, V, I, V, V, V, V, V, V],
    "ndp_chamfer_nn_matrix": [V, I, V, I, V, V, V, V, V, V],
    "ndp_engine_nn_matrix_fits": [I],
    "ndp_engine_nn_onepass_fits": [I],
    "ndp_chamfer_l1_bwd": [V, I, V, I, F, V, V, V, V, V, V, I, V],
    "ndp_flow_metrics": [V, V, V, I, V, V],
    "ndp_landmark_mse_fwd_bwd": [V, V, I, V, V, V],
    "ndp_adam_step": [V, V, V, V, I, F, F, F, F, F, F, V],
    "ndp_engine_run": [ctypes.POINTER(Engine), I, I, V],
    "ndp_engine_run_timed": [ctypes.POINTER(Engine), I, I, V, c_float_p],
    "ndp_engine_run_stages": [ctypes.POINTER(Engine), I, I, I, V],
    "ndp_engine_load": [ctypes.POINTER(Engine), I, ctypes.POINTER(LoadJob), I, V],
}
EXPORTS = ["ndp_version", "ndp_last_error", "ndp_build_id", "ndp_abi_sizes", "ndp_engine_nn_workspace"] + list(_SIGS)


_VARIANT = None


def use_variant(path):
    """Measurement tools only (tools/_modes.py): load `path` -- a timing / experiment build of libndp_hip.so -- instead of the
    product library, before anything else touched it.  Explicit by construction: no environment variable reaches this, the build-id
    check is the caller's to give up, and `variant()` lets bench.py record (and refuse) it."""
    global _VARIANT
    if _LIB is not None:
        raise NdpError("use_variant() after the library was loaded")
    _VARIANT = os.path.abspath(path)


def variant():
    return _VARIANT


def lib(allow_build=True):
    """Load the native library (building it first if the source is newer and hipcc exists)."""
    global _LIB
    if _LIB is None:
        # The library carries the digest of the sources it was built from (ndp_build_id): an absent or stale library is
        # rebuilt (under a file lock, so N ranks build once), and one that cannot be rebuilt is refused -- never loaded.
        override = _VARIANT                          # set by use_variant() only: the package reads no environment variable
        if not override:
            have_hipcc = bool(shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))
            if allow_build and have_hipcc and _stale():
                build()
            if not os.path.exists(LIBPATH):
                raise NdpError(f"{LIBPATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
            if source_id() is not None and _built_id(LIBPATH) != source_id():
                raise NdpError(f"{LIBPATH} was built from other sources (build id {_built_id(LIBPATH)}, tree {source_id()}): "
                               "run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(override or LIBPATH)
        L.ndp_version.restype = I
        L.ndp_last_error.restype = ctypes.c_char_p
        L.ndp_build_id.restype = ctypes.c_char_p
        L.ndp_abi_sizes.argtypes = [c_int_p]
        L.ndp_abi_sizes.restype = I
        L.ndp_engine_nn_workspace.argtypes = [I, I, ctypes.POINTER(ctypes.c_longlong)]
        L.ndp_engine_nn_workspace.restype = I
        sizes = (ctypes.c_int * 6)()
        L.ndp_abi_sizes(sizes)
        mine = [ctypes.sizeof(t) for t in (CLayerDesc, PairGeom, PairState, Engine, WarpJob, LoadJob)]
        if list(sizes) != mine:
            raise NdpError(f"struct layouts differ between {override or LIBPATH} {list(sizes)} and the ctypes mirror {mine}")
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = I
        _LIB = L
    return _LIB


# ------------------------------------------------------------------ host runtime library (no GPU)
HOST_LIBPATH = os.path.join(LIBDIR, "libndp_host.so")
HOST_SOURCE = os.path.join(CSRC, "ndp_host.cpp")
_HOST = None


class DrawOp(ctypes.Structure):
    _fields_ = [("n", ctypes.c_longlong), ("lo", ctypes.c_float), ("hi", ctypes.c_float), ("offset", ctypes.c_longlong)]


HOST_SOURCES = [HOST_SOURCE, os.path.join(CSRC, "ndp_graph.cpp")]      # RNG replay; embedded-deformation graph builder


HOST_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-mfma", "-mavx2", "-fPIC", "-shared"]


def _host_id():
    """Digest of the host library's sources and flags (compiled in as NDP_HOST_BUILD_ID), or None without sources."""
    import hashlib
    h = hashlib.sha256()
    try:
        for src in HOST_SOURCES:
            with open(src, "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    h.update(" ".join(HOST_FLAGS).encode())
    return h.hexdigest()[:16]


def _host_stale():
    hid = _host_id()
    built = _built_id(HOST_LIBPATH, b"NDP_HOST_ID")
    if hid is None:
        return built is None
    return built != hid


def build_host(force=False):
    """g++ -> deformationpyramid_amd/lib/libndp_host.so; rebuilt when the digest of its sources changes (the digest is
    compiled into the library).  Serialised by a file lock, staleness re-checked under the lock: N ranks build once."""
    if force or _host_stale():
        import fcntl
        os.makedirs(LIBDIR, exist_ok=True)
        with open(os.path.join(LIBDIR, ".build_host.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not force and not _host_stale():
                return HOST_LIBPATH
            tmp = f"{HOST_LIBPATH}.{os.getpid()}.tmp"
            subprocess.check_call(["g++"] + HOST_FLAGS + [f'-DNDP_HOST_BUILD_ID="{_host_id()}"', "-o", tmp] + HOST_SOURCES)
            os.replace(tmp, HOST_LIBPATH)
    return HOST_LIBPATH


def host_lib():
    global _HOST
    if _HOST is None:
        if shutil.which("g++") and _host_id() is not None:
            build_host()                                # no-op when the embedded digest matches
        if not os.path.exists(HOST_LIBPATH):
            raise NdpError(f"{HOST_LIBPATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        if _host_stale():                               # cannot be rebuilt here (no g++): refuse it, like the HIP library
            raise NdpError(f"{HOST_LIBPATH} was built from other sources (build id {_built_id(HOST_LIBPATH, b'NDP_HOST_ID')}, "
                           f"tree {_host_id()}): run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(HOST_LIBPATH)
        L.ndp_rng_replay.argtypes = [V, ctypes.c_longlong, ctypes.POINTER(DrawOp), I, I, V, ctypes.c_longlong]
        L.ndp_rng_replay.restype = I
        L.ndp_rng_skip.argtypes = [V, ctypes.c_longlong, ctypes.c_longlong]
        L.ndp_rng_skip.restype = I
        L.ndp_pair_draws.argtypes = [ctypes.POINTER(DrawOp), I, I, I]
        L.ndp_pair_draws.restype = ctypes.c_longlong
        L.ndp_pair_init.argtypes = [V, ctypes.c_longlong, ctypes.POINTER(DrawOp), I, V, I, I, I, V, V, V]
        L.ndp_pair_init.restype = I
        L.ndp_host_build_id.restype = ctypes.c_char_p
        L.ndp_depth_to_mesh.argtypes = [V, I, I, F, V, V, V, c_int_p, c_int_p]
        L.ndp_erode_mesh.argtypes = [I, V, I, I, I, V]
        L.ndp_sample_nodes.argtypes = [V, I, V, F, I, V]
        L.ndp_edges_geodesic.argtypes = [V, I, V, V, I, V, I, I, F, I, I, V, V, V]
        L.ndp_edges_geodesic.restype = V
        L.ndp_geodesic_free.argtypes = [V]
        L.ndp_geodesic_free.restype = None
        L.ndp_node_cleanup.argtypes = [V, I, I, V]
        L.ndp_pixel_anchors.argtypes = [V, V, V, I, I, I, F, V, V]
        for name in ("ndp_depth_to_mesh", "ndp_erode_mesh", "ndp_sample_nodes", "ndp_node_cleanup", "ndp_pixel_anchors"):
            getattr(L, name).restype = I
        _HOST = L
    return _HOST


def make_draw_ops(ops):
    return (DrawOp * len(ops))(*[DrawOp(int(n), float(lo), float(hi), int(off)) for n, lo, hi, off in ops])


def rng_replay(ops, out):
    """Run draw ops on torch's global CPU generator (state exported, advanced natively, re-imported)."""
    import torch
    if not isinstance(ops, ctypes.Array):
        ops = make_draw_ops(ops)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.device.type == "cpu"
    st = torch.get_rng_state()
    rc = host_lib().ndp_rng_replay(ctypes.c_void_p(st.data_ptr()), st.numel(), ops, len(ops), 1,
                                   ctypes.c_void_p(out.data_ptr()), 0)
    if rc != 0:
        raise NdpError("ndp_rng_replay failed")
    torch.set_rng_state(st)


def check(rc, what):
    if rc != 0:
        msg = lib().ndp_last_error().decode(errors="replace")
        raise NdpError(f"{what} failed (rc={rc}): {msg}")


def stream_ptr(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
