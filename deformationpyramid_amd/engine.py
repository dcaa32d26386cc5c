"""Batched NDP optimisation engine -- host plumbing around ndp_engine_run().

B independent pairs are resident in HBM at once ("slots").  One *tick* is one iteration of the
inner loop of optimize_deformation_pyramid (/root/reference/model/registration.py:184-238) for
every unfinished slot: level forward -> 1-NN in both directions -> loss + early-stop decision +
backward -> gradient fold + Adam (+ level hand-over).  Each slot is at its own level/iteration;
the decision is taken on the device, the host only polls the slot states every few ticks.

Memory per slot (n_cap = t_cap = 2048, m = 9): params 1.25 MB, activations 3 MB, Adam 0.28 MB,
gradient partials G x 0.14 MB -- hundreds of slots fit easily in 288 GB of HBM3E.
"""
import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as N
from .layout import LayerDesc
from .ops import adam_scalars, cap


@dataclass
class OptConfig:
    m: int = 9
    k0: int = -8
    iters: int = 500
    lr: float = 0.01
    max_break_count: int = 15
    break_threshold_ratio: float = 0.001
    w_cd: float = 1.0          # weight of the Chamfer term (1 without landmarks; config.w_cd with)
    trunc: float = 1e9         # truncation in squared units (registration.py:212 / config.trunc_cd)
    w_reg: float = 0.0         # nonrigidity BCE weight (registration.py:216-220); needs desc.nonrigidity
    early_stop: bool = True


# Arithmetic of the three level kernels / shape of the nearest-neighbour kernel when the caller does not say (constructor argument >
# these defaults; nothing here reads the environment -- the measurement tools pass what they want, tools/_modes.py):
#   gemm_mode 7 (default since round 3): (mask 1 forward | 2 bwd1 | 4 bwd2) the 128x128 contractions as two-way fp16 splits (three
#                products) on the fp16 MFMA with fp32 accumulation -- closer to a float64 evaluation than the fp32 chain is
#                (tests/test_split_accuracy.py), every engine parity test passes at the same tolerances, but not bitwise the chain.
#                With both backward bits set the two backward layers run as ONE launch (k_eng_bwd_f, round 4: dz1 stays in LDS);
#                + 16: as the two round-3 launches (k_eng_bwd2_8, k_eng_bwd1_8); + 32 (tests): the fused backward also writes dz1;
#                + 8 (tests): the split forward also stores h0, which the split backward recomputes;
#             0: the same contractions on the fp32 MFMA, bitwise the oracle's fma chain (Registration(cfg, gemm_mode=0),
#                bench.py --gemm-mode 0) -- 1/16 of the 16-bit matrix rate;
#   nn matrix : the one-pass NN with the distances on the bf16 matrix pipe and exact re-evaluation (bit-identical results).
DEFAULT_GEMM_MODE = 7
DEFAULT_NN_MATRIX = True


def resolve_modes(B, n_cap, t_cap, gemm_mode=None, nn_mode=None, nn_matrix=None):
    """-> (gemm_mode, nn_mode) an engine of B slots x (n_cap, t_cap) will run with, validated against the LDS limits of the
    one-pass kernels.  An explicit nn_mode that does not fit raises; a default / environment choice degrades (2 -> 0 -> 1).
    nn_matrix (None: DEFAULT_NN_MATRIX) only states a preference for the throughput shape: matrix-pipe kernel where it fits."""
    lib = N.lib()
    if gemm_mode is None:
        gemm_mode = DEFAULT_GEMM_MODE
    gemm_mode = int(gemm_mode)
    if not 0 <= gemm_mode <= 2047:
        raise N.NdpError(f"gemm_mode must be a mask of 1 (forward) | 2 (bwd1) | 4 (bwd2) [| 8: the split forward also stores h0 | 16: "
                         f"the split backward as two launches | 32: the fused backward also writes dz1 -- tests | 64: the Adam step inside the fused backward | 256: persistent small-batch tick | 512: the per-point warp as a launch of its own | 1024: at G = 1 the Adam step of the two 128 x 128 matrices behind the fused backward's tile loop], got {gemm_mode}")
    fits2 = bool(lib.ndp_engine_nn_matrix_fits(n_cap))
    fits0 = bool(lib.ndp_engine_nn_onepass_fits(n_cap))
    if nn_mode is not None:
        nn_mode = int(nn_mode)
        if nn_mode not in (0, 1, 2):
            raise N.NdpError(f"nn_mode must be 0 (one pass, vector pipe), 1 (latency shape) or 2 (one pass, matrix pipe), got {nn_mode}")
        if (nn_mode == 2 and not fits2) or (nn_mode == 0 and not fits0):
            raise N.NdpError(f"nn_mode {nn_mode}: n_cap = {n_cap} sources do not fit the kernel's LDS table "
                             "(ndp_engine_nn_matrix_fits / ndp_engine_nn_onepass_fits); use nn_mode 1 or leave the choice to the engine")
        return gemm_mode, nn_mode
    if B * (t_cap // 256 + 1) < 256:               # few resident pairs: the one-pass kernels have only t_cap/256 workgroups per
        want = 1                                     # pair, the latency shape has (n_cap + t_cap)/64
    else:
        want = 2 if (DEFAULT_NN_MATRIX if nn_matrix is None else nn_matrix) else 0
    if want == 2 and not fits2:
        want = 0
    if want == 0 and not fits0:
        want = 1
    return gemm_mode, want


class Snapshot:
    """Host copy of the [B] pair states of one tick."""
    __slots__ = ("raw", "sz", "level")

    def __init__(self, raw, sz):
        self.raw, self.sz = raw, sz                              # raw: uint8 [B, sz]
        self.level = raw[:, :4].copy().view(np.int32).reshape(-1)   # ndp_pair_state.level is the first field

    def state(self, slot):
        return N.PairState.from_buffer_copy(self.raw[slot].tobytes())

    def __getitem__(self, slot):
        return self.state(slot)


class BatchedEngine:
    def __init__(self, desc: LayerDesc, cfg: OptConfig, B: int, n_cap: int, t_cap: int, device, G=None, nn_mode=None, gemm_mode=None, nn_matrix=None):
        # desc.nonrigidity = True means "every level but the first carries the gate" (nets.py:26); P is then the
        # parameter count of a gated level and level 0 uses a prefix-compatible shorter layout.
        self.lib = N.lib()
        self.desc, self.cfg, self.B = desc, cfg, B
        self.n_cap, self.t_cap = cap(n_cap), cap(t_cap)
        # modes are resolved and validated HERE, once (see resolve_modes): nothing later reads the environment
        self.gemm_mode, self.nn_mode = resolve_modes(B, self.n_cap, self.t_cap, gemm_mode, nn_mode, nn_matrix)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.NdpError("BatchedEngine needs a GPU device; there is no CPU fallback")
        self.P = desc.param_count
        self.p_stride = (self.P + 63) // 64 * 64
        tiles = self.n_cap // N.TILE
        # workgroups per pair in the level kernels: two 4-wave workgroups per CU (fp32 kernels) or, when all three level kernels run
        # on fp16 splits, one 8-wave workgroup per CU -- then also half as many gradient partials to write and to fold
        # (width / depth other than 128 / 3: the generic fp32 kernels, 4-wave workgroups; gemm_mode selects nothing there)
        self.generic = desc.width != 128 or desc.n_hidden != 2
        per_cu = 1 if (self.gemm_mode & 7) == 7 and not self.generic else 2
        self.G = int(G) if G else max(1, min(tiles, -(-256 * per_cu // B)))
        d = self.device
        f32 = dict(device=d, dtype=torch.float32)
        m = cfg.m
        self.pts = torch.zeros(B, 2, self.n_cap, 3, **f32)
        self.ldmk_t = torch.zeros(B, self.n_cap, 3, **f32)
        self.tgt = torch.zeros(B, self.t_cap, 3, **f32)
        self.params = torch.zeros(B, m, self.p_stride, **f32)
        self.gpart = torch.zeros(B, self.G, self.p_stride, **f32)
        self.adam_m = torch.zeros(B, self.p_stride, **f32)
        self.adam_v = torch.zeros(B, self.p_stride, **f32)
        self.act = torch.zeros(B, desc.n_hidden + 1, self.n_cap, desc.width, **f32)    # [B, 3, n_cap, 128] for the shipped 128 / 3
        self.heads = torch.zeros(B, self.n_cap, N.HROW, **f32)
        self.dO = torch.zeros(B, self.n_cap, N.NHMAX, **f32)
        self.d2x = torch.zeros(B, self.n_cap, **f32)
        self.d2y = torch.zeros(B, self.t_cap, **f32)
        self.idx_x = torch.zeros(B, self.n_cap, device=d, dtype=torch.int32)
        self.idx_y = torch.full((B, self.t_cap), -1, device=d, dtype=torch.int32)
        nr = ctypes.c_longlong()
        N.check(self.lib.ndp_engine_nn_workspace(self.n_cap, self.t_cap, ctypes.byref(nr)), "ndp_engine_nn_workspace")
        self.nn_row = torch.zeros(B, max(nr.value, 1), **f32)   # one-pass 1-NN row partials ({d2, idx} per source and target chunk)
        tab = np.zeros((cfg.iters + 1, 2), dtype=np.float32)
        for t in range(1, cfg.iters + 1):
            tab[t] = adam_scalars(t, cfg.lr)
        self.adam_tab = torch.from_numpy(tab).to(d)
        self.state_nbytes = ctypes.sizeof(N.PairState)
        self.state = torch.zeros(2, B, self.state_nbytes, device=d, dtype=torch.uint8)
        self.geom = torch.zeros(B, 4, device=d, dtype=torch.int32)
        self.gmax = torch.zeros(2 * B, device=d, dtype=torch.int32)  # [0, B): max |dO| per pair and tick, the split backward's gradient scale; [B, 2B): tickets of gemm_mode bit 64
        self._geom_h = np.zeros((B, 4), dtype=np.int32)
        self._state_h = torch.zeros(B, self.state_nbytes, dtype=torch.uint8).pin_memory()
        self._snap = [torch.zeros(B, self.state_nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._snap_i = 0
        self.tick = 0
        self._mk_struct()

    def _mk_struct(self):
        c, e = self.cfg, N.Engine()
        e.desc = self.desc.c_struct()
        e.m, e.k0, e.P, e.p_stride = c.m, c.k0, self.P, self.p_stride
        e.iters, e.max_break_count, e.early_stop = c.iters, c.max_break_count, int(bool(c.early_stop))
        e.B, e.G, e.n_cap, e.t_cap = self.B, self.G, self.n_cap, self.t_cap
        e.gemm_mode = self.gemm_mode
        e.break_threshold_ratio = c.break_threshold_ratio
        e.w_cd, e.trunc = c.w_cd, c.trunc
        e.w_reg = c.w_reg if self.desc.nonrigidity else 0.0
        e.adam_w1, e.adam_b2, e.adam_w2, e.adam_eps = 1 - 0.9, 0.999, 1 - 0.999, 1e-8
        e.nn_mode = self.nn_mode
        for name in ("geom", "state", "pts", "ldmk_t", "tgt", "params", "gpart", "adam_m", "adam_v", "act", "heads",
                     "d2x", "idx_x", "d2y", "idx_y", "adam_tab", "dO", "nn_row", "gmax"):
            setattr(e, name, getattr(self, name).data_ptr())
        self.c_engine = e

    # ------------------------------------------------------------------ slot management
    # Nothing here may block the host: a slot is filled by ONE kernel launch (k_eng_load) from device-resident
    # inputs (a pageable H2D copy is synchronous and would drain the tick pipeline behind it).
    def load_jobs(self, jobs):
        """Fill / park slots, one launch per 16 jobs (ndp_engine_load).  Each job is a dict with `slot` and either
        nothing else (park) or: params [m,p_stride] device tensor, K, S, T, src [*,3], tgt [*,3] | None,
        perm_s / perm_t (device int32, first S / T entries used) | None, ldmk_s / ldmk_t [K,3] | None,
        means [8] | None (with n_src / n_tgt > 0 the load call computes them into that tensor first: one launch for the group).
        The tensors must stay alive until the launch has run (the caller holds them)."""
        ptr = lambda t: t.data_ptr() if t is not None else None
        for i0 in range(0, len(jobs), N.MAX_LOAD_JOBS):
            group = jobs[i0:i0 + N.MAX_LOAD_JOBS]
            arr = (N.LoadJob * len(group))()
            for q, j in zip(arr, group):
                q.slot = j["slot"]
                params = j.get("params")
                if params is None:
                    continue
                if params.dtype != torch.float32 or not params.is_cuda or not params.is_contiguous() or \
                        tuple(params.shape) != (self.cfg.m, self.p_stride):
                    raise N.NdpError("load_jobs: params must be a contiguous float32 device tensor [m, p_stride]")
                K, S, T = int(j["K"]), int(j["S"]), int(j.get("T", 0))
                if K + S > self.n_cap or T > self.t_cap or K + S < 1:
                    raise ValueError(f"pair does not fit the engine capacities: n={K + S}/{self.n_cap}, T={T}/{self.t_cap}")
                q.params, q.K, q.S, q.T = params.data_ptr(), K, S, T
                q.src, q.tgt = ptr(j.get("src")), ptr(j.get("tgt"))
                q.perm_s, q.perm_t = ptr(j.get("perm_s")), ptr(j.get("perm_t"))
                q.ldmk_s, q.ldmk_t = ptr(j.get("ldmk_s")), ptr(j.get("ldmk_t"))
                q.means = ptr(j.get("means"))
                q.n_src, q.n_tgt = int(j.get("n_src", 0)), int(j.get("n_tgt", 0))     # > 0: the means are computed by this call
                self._geom_h[q.slot] = (K, S, T, 0)
            N.check(self.lib.ndp_engine_load(ctypes.byref(self.c_engine), self.tick, arr, len(group),
                                             N.stream_ptr(self.device)), "ndp_engine_load")

    def load(self, slot, pts, K, S, ldmk_t, tgt, params):
        """Already centred / sampled inputs: pts [K+S,3] (landmarks first); ldmk_t [K,3]; tgt [T,3];
        params [m, >=P] initial parameters of every level (device or host tensor)."""
        if pts.shape[0] != K + S:
            raise ValueError("pts must hold K landmarks followed by S samples")
        dv = lambda t: None if t is None else t.to(self.device, dtype=torch.float32).contiguous()
        pts, ldmk_t, tgt, params = dv(pts), dv(ldmk_t), dv(tgt), dv(params)
        if tuple(params.shape) != (self.cfg.m, self.p_stride) or not params.is_contiguous():
            full = torch.zeros(self.cfg.m, self.p_stride, device=self.device, dtype=torch.float32)
            full[:, :self.P] = params[:, :self.P]
            params = full
        job = dict(slot=slot, params=params, K=K, S=S, T=0 if tgt is None else tgt.shape[0],
                   src=pts[K:] if S else None, ldmk_s=pts if K else None, ldmk_t=ldmk_t if K else None, tgt=tgt)
        self.load_jobs([job])
        self._keep = (job, pts)                    # keep the inputs alive until the next load

    def park(self, slot):
        """Mark a slot as finished (empty)."""
        self.load_jobs([dict(slot=slot)])

    def park_all(self):
        self.load_jobs([dict(slot=s) for s in range(self.B)])

    def run_ticks(self, n_ticks):
        N.check(self.lib.ndp_engine_run(ctypes.byref(self.c_engine), self.tick, int(n_ticks),
                                        N.stream_ptr(self.device)), "ndp_engine_run")
        self.tick += n_ticks

    def run_stages(self, lo, hi):
        """Stages lo..hi (0 forward, 1 NN, 2 loss, 3 bwd2, 4 bwd1, 5 update) of the CURRENT tick; the tick counter advances
        once stage 5 has run.  Test / measurement aid (ndp_engine_run_stages)."""
        N.check(self.lib.ndp_engine_run_stages(ctypes.byref(self.c_engine), self.tick, int(lo), int(hi),
                                               N.stream_ptr(self.device)), "ndp_engine_run_stages")
        if hi == len(N.TICK_KERNELS) - 1:
            self.tick += 1

    def run_ticks_timed(self, n_ticks):
        """-> per-kernel summed milliseconds, in N.TICK_KERNELS order (HIP events on the launch stream)."""
        ms = (ctypes.c_float * len(N.TICK_KERNELS))()
        N.check(self.lib.ndp_engine_run_timed(ctypes.byref(self.c_engine), self.tick, int(n_ticks),
                                              N.stream_ptr(self.device), ms), "ndp_engine_run_timed")
        self.tick += n_ticks
        return [float(x) for x in ms]

    def read_states(self):
        self._state_h.copy_(self.state[self.tick & 1], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        raw = self._state_h.numpy().tobytes()
        sz = self.state_nbytes
        return [N.PairState.from_buffer_copy(raw[i * sz:(i + 1) * sz]) for i in range(self.B)]

    def snapshot_async(self):
        """Enqueue a device->host copy of the current pair states; returns a handle for wait_snapshot().
        Two pinned buffers alternate, so at most one snapshot may be outstanding besides the newest."""
        buf = self._snap[self._snap_i]
        self._snap_i ^= 1
        buf.copy_(self.state[self.tick & 1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev, buf

    def wait_snapshot(self, handle):
        """-> Snapshot of the B pair states: `.level` is an int32 array (one vectorised look decides which slots have finished);
        `.state(slot)` builds the full PairState of one slot.  (Building B ctypes objects per snapshot made the host the limit of
        the landmark configuration: 0.25 ms per lane step against 0.68 ms of GPU work per chunk for three lanes.)"""
        ev, buf = handle
        ev.synchronize()
        return Snapshot(buf.numpy().copy(), self.state_nbytes)

    def run_until_done(self, chunk=32, max_ticks=None, kernel_ms=None):
        """Advance every loaded slot to the end of its last level; returns the slot states.  kernel_ms (a list of
        len(N.TICK_KERNELS) floats, optional) accumulates the per-kernel device milliseconds of the ticks (HIP events on the
        launch stream, ndp_engine_run_timed) -- what register(timer=...) apportions to the reference's timer keys."""
        limit = max_ticks if max_ticks is not None else self.cfg.m * (self.cfg.iters + 1) + chunk
        done_ticks = 0
        while True:
            if kernel_ms is None:
                self.run_ticks(chunk)
            else:
                for j, v in enumerate(self.run_ticks_timed(chunk)):
                    kernel_ms[j] += v
            done_ticks += chunk
            st = self.read_states()
            if all(s.level >= self.cfg.m for s in st) or done_ticks >= limit:
                return st

    def final_points(self, slot, state):
        """Points of a slot warped through every optimised level (what the next level would read)."""
        g = self._geom_h[slot]
        return self.pts[slot, state.cur, :g[0] + g[1]]
