"""Registration -- drop-in for the NDP path of the reference's model/registration.py.

    model = Registration(config)                       (/root/reference/model/registration.py:27-34)
    model.load_pcds(src, tgt, landmarks=None)          (:93-103)
    warped, iter_cnt, timer = model.register(visualize=False, timer=None)   (:106-123, :126-262)
    model.src_pcd                                      un-centred source on the device (eval_nolearned.py:94)

plus one extension the reference has no counterpart for, because its loop is one pair at a time:

    results = model.register_batch([(src, tgt[, landmarks]), ...], slots=64)

which keeps `slots` pairs resident on the GPU and advances them together (engine.py).

Host work stays in torch (tensor plumbing, the CPU RNG replay of the reference's init and
randperm calls); the optimisation itself runs in libndp_hip.so.  There is no CPU fallback:
config.device must be a GPU.
"""
import queue
import threading
import time

import numpy as np
import torch

from . import _native as N
from . import ops
from .engine import BatchedEngine, OptConfig
from .layout import LayerDesc
from .nets import init_pyramid_store


class _Prepared:
    """Everything one pair needs on the device before it enters an engine slot."""
    __slots__ = ("src_centered", "tgt_mean", "pts", "K", "S", "ldmk_t", "tgt_sample", "desc", "store", "result",
                 "state", "src_pcd")


class Registration:
    def __init__(self, config):
        self.tgt_pcd = None
        self.src_pcd = None
        self.landmarks = None
        self.config = config
        self.device = config.device
        self.deformation_model = config.deformation_model
        self._engines = {}
        self.last_state = None

    # ------------------------------------------------------------------ reference surface
    def load_pcds(self, src, tgt, landmarks=None):
        if isinstance(src, np.ndarray):
            src = torch.from_numpy(src)
            tgt = torch.from_numpy(tgt)
        self.src_pcd = src.to(self.device)
        self.tgt_pcd = tgt.to(self.device)
        self.landmarks = landmarks

    def register(self, **kwargs):
        if self.deformation_model == "NDP":
            return self.optimize_deformation_pyramid(**kwargs)
        # Sinkhorn / ED / NSFP / Nerfies are comparison baselines outside this path
        raise KeyError(self.deformation_model)

    def optimize_deformation_pyramid(self, visualize=False, timer=None):
        if visualize:
            raise NotImplementedError("mayavi visualisation is outside the hot path")
        t0 = time.time()
        prep = self._prepare(self.src_pcd, self.tgt_pcd, self.landmarks)
        self.src_pcd = prep.src_pcd
        eng = self._engine(1, prep)
        eng.load(0, prep.pts, prep.K, prep.S, prep.ldmk_t, prep.tgt_sample, prep.store)
        st = eng.run_until_done(chunk=32)[0]
        warped = self._finish(eng, 0, prep, st)
        self.last_state = st
        iter_cnt = {lvl: int(st.evals_per_level[lvl]) for lvl in range(self.config.m)}
        if timer is not None:
            torch.cuda.synchronize(self._dev())
            timer.tictoc("ndp_engine", time.time() - t0)
        return warped, iter_cnt, timer

    # ------------------------------------------------------------------ batched extension
    def register_batch(self, pairs, slots=64, chunk=8, prefetch=True):
        """pairs: sequence of (src, tgt) or (src, tgt, (ldmk_s, ldmk_t)).  Pairs are prepared in order
        (so the CPU RNG stream is consumed exactly as by sequential register() calls) and optimised
        `slots` at a time, finished slots being refilled.  With prefetch=True the host-side preparation
        (RNG replay of the init, centring, sampling) runs in a producer thread ahead of the GPU.
        Returns [(warped, iter_cnt)] in input order."""
        pairs = list(pairs)
        if not pairs:
            return []
        todo = queue.Queue(maxsize=max(2 * slots, 8))
        preps = [None] * len(pairs)
        dev = self._dev()
        side = torch.cuda.Stream(dev) if prefetch else None
        fin_stream = torch.cuda.Stream(dev)                      # final all-point warps overlap the ticking engine

        def produce():
            try:
                for i, item in enumerate(pairs):
                    src, tgt = item[0], item[1]
                    ldmk = item[2] if len(item) > 2 else None
                    if isinstance(src, np.ndarray):
                        src, tgt = torch.from_numpy(src), torch.from_numpy(tgt)
                    if side is not None:
                        with torch.cuda.stream(side):
                            p = self._prepare(src.to(dev), tgt.to(dev), ldmk)
                            ev = torch.cuda.Event()
                            ev.record(side)
                    else:
                        p, ev = self._prepare(src.to(dev), tgt.to(dev), ldmk), None
                    todo.put((i, p, ev))
                todo.put(None)
            except BaseException as e:       # surface producer failures in the consumer
                todo.put(e)

        if prefetch:
            th = threading.Thread(target=produce, daemon=True)
            th.start()
        else:
            # unbounded in-line preparation
            todo = queue.Queue()
            produce()

        def next_prepared():
            item = todo.get()
            if item is None:
                return None
            if isinstance(item, BaseException):
                raise item
            i, p, ev = item
            if ev is not None:
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                for name in ("src_centered", "tgt_mean", "pts", "ldmk_t", "tgt_sample", "src_pcd", "store"):
                    t = getattr(p, name)
                    if t is not None:
                        t.record_stream(cur)         # allocated on the producer's stream, consumed here
                        t.record_stream(fin_stream)
            else:
                for name in ("src_centered", "tgt_mean"):
                    getattr(p, name).record_stream(fin_stream)
            preps[i] = p
            return i, p

        first = next_prepared()
        B = min(slots, len(pairs))
        eng = self._engine(B, first[1], n_hint=self.config.samples + (first[1].K if first[1].K else 0))
        for slot in range(B):
            eng.park(slot)
        # Pipelined control loop: the states of chunk k are read back while chunk k+1 runs, so the GPU
        # never waits for the host; a slot that finishes in chunk k is refilled before chunk k+2.
        fin_done = {}                                            # slot -> event: its parameters have been consumed
        main = torch.cuda.current_stream(dev)
        active, free, exhausted = {}, list(range(B)), False      # active: slot -> (pair index, first valid snapshot)
        nxt = first
        m = self.config.m
        seq = 0                                                  # snapshots taken so far
        pending = None
        while True:
            while free and not exhausted:
                if nxt is None:
                    nxt = next_prepared()
                    if nxt is None:
                        exhausted = True
                        break
                i, p = nxt
                nxt = None
                slot = free.pop()
                if slot in fin_done:
                    main.wait_event(fin_done.pop(slot))          # the previous tenant's final warp read these params
                eng.load(slot, p.pts, p.K, p.S, p.ldmk_t, p.tgt_sample, p.store)
                active[slot] = (i, seq)                          # snapshots >= seq see this pair in the slot
            if not active and pending is None:
                break
            handle = None
            if active:
                eng.run_ticks(chunk)
                handle = (eng.snapshot_async(), seq)
                seq += 1
            if pending is not None:
                (h, hseq) = pending
                states = eng.wait_snapshot(h)
                for slot in list(active):
                    i, valid_from = active[slot]
                    st = states[slot]
                    if hseq < valid_from or st.level < m:
                        continue
                    del active[slot]
                    p = preps[i]
                    # the snapshot proves every tick that touched this slot has completed: the final warp
                    # needs no dependency on the main stream, only the slot's refill must wait for it
                    with torch.cuda.stream(fin_stream):
                        frozen = eng.params[slot].clone()            # 1.25 MB device copy: the slot is free again
                        ev = torch.cuda.Event()
                        ev.record(fin_stream)
                        p.result = self._finish(eng, slot, p, st, store=frozen)
                    fin_done[slot] = ev
                    p.result.record_stream(main)
                    p.state = st
                    eng.park(slot)
                    free.append(slot)
            pending = handle
        main.wait_stream(fin_stream)
        self.last_states = [p.state for p in preps]
        return [(p.result, {lvl: int(p.state.evals_per_level[lvl]) for lvl in range(m)}) for p in preps]

    # ------------------------------------------------------------------ internals
    def _dev(self):
        d = self.device
        if isinstance(d, int):
            return torch.device("cuda", d)
        d = torch.device(d)
        if d.type != "cuda":
            raise N.NdpError("deformationpyramid_amd runs the NDP path on the GPU only (config.device is CPU); "
                             "there is no CPU fallback")
        return d

    def _opt_config(self, has_ldmk):
        c = self.config
        if has_ldmk:
            w_cd, trunc = float(c.w_cd), float(c.trunc_cd)                       # registration.py:189-197
        else:
            w_cd, trunc = 1.0, 1e9                                               # registration.py:212
        return OptConfig(m=c.m, k0=c.k0, iters=c.iters, lr=c.lr, max_break_count=c.max_break_count,
                         break_threshold_ratio=c.break_threshold_ratio, w_cd=w_cd, trunc=trunc,
                         w_reg=float(c.w_reg), early_stop=True)

    def _prepare(self, src_pcd, tgt_pcd, landmarks):
        c = self.config
        dev = self._dev()
        if c.depth != 3 or c.width != 128:
            raise N.NdpError("the HIP kernels are specialised for depth=3, width=128 (NDP.yaml / LNDP.yaml)")
        p = _Prepared()
        # registration.py:133-140 -- all m levels are initialised up front on the CPU generator
        gate = c.w_reg > 0                                                          # registration.py:138
        p.desc = LayerDesc(width=c.width, n_hidden=c.depth - 1, motion=c.motion_type, rotfmt=c.rotation_format,
                           nonrigidity=gate)                                        # engine: "levels > 0 gated"
        level0 = LayerDesc(width=c.width, n_hidden=c.depth - 1, motion=c.motion_type, rotfmt=c.rotation_format)
        stride = (p.desc.param_count + 63) // 64 * 64
        host = self._pinned_store(c.m, stride)                                     # reused pinned staging buffer
        init_pyramid_store([level0] + [p.desc] * (c.m - 1), c.depth, stride, out=host)   # nets.py:26
        p.store = host.to(dev, non_blocking=True)                                  # async upload on the current stream
        self._pin_busy.append((host, torch.cuda.Event()))
        self._pin_busy[-1][1].record(torch.cuda.current_stream(dev))
        src_pcd = src_pcd.to(dev).float()
        tgt_pcd = tgt_pcd.to(dev).float()
        p.src_pcd = src_pcd
        src_mean = src_pcd.mean(dim=0, keepdim=True)                              # :150-153
        p.tgt_mean = tgt_pcd.mean(dim=0, keepdim=True)
        p.src_centered = (src_pcd - src_mean).contiguous()
        tgt_c = tgt_pcd - p.tgt_mean
        perm_s = torch.randperm(src_pcd.shape[0])                                 # :156-159 (CPU RNG)
        perm_t = torch.randperm(tgt_pcd.shape[0])
        s_sample = p.src_centered[perm_s[: c.samples].to(dev)]
        t_sample = tgt_c[perm_t[: c.samples].to(dev)]
        if landmarks is not None:
            ls = landmarks[0].to(dev).float() - src_mean                          # :162-164
            p.ldmk_t = (landmarks[1].to(dev).float() - p.tgt_mean).contiguous()
            p.K = ls.shape[0]
            if c.w_cd > 0:
                p.S = s_sample.shape[0]
                p.pts = torch.cat([ls, s_sample]).contiguous()                    # :190
                p.tgt_sample = t_sample.contiguous()
            else:
                p.S = 0
                p.pts = ls.contiguous()
                p.tgt_sample = None
        else:
            p.K, p.S, p.ldmk_t = 0, s_sample.shape[0], None
            p.pts = s_sample.contiguous()
            p.tgt_sample = t_sample.contiguous()
        p.result = p.state = None
        return p

    def _pinned_store(self, m, stride):
        """A pinned [m, stride] host buffer whose previous upload has completed (small ring, allocated once:
        cudaHostAlloc costs milliseconds)."""
        if not hasattr(self, "_pin_free"):
            self._pin_free, self._pin_busy = [], []
        while self._pin_busy and self._pin_busy[0][1].query():
            self._pin_free.append(self._pin_busy.pop(0)[0])
        for i, buf in enumerate(self._pin_free):
            if buf.shape == (m, stride):
                return self._pin_free.pop(i)
        if len(self._pin_busy) >= 64:                                               # bound the ring: wait for the oldest
            host, ev = self._pin_busy.pop(0)
            ev.synchronize()
            if host.shape == (m, stride):
                return host
        return torch.empty(m, stride, dtype=torch.float32).pin_memory()

    def _engine(self, B, like, n_hint=0):
        n_cap = ops.cap(max(like.K + like.S, n_hint))
        t_cap = ops.cap(max(0 if like.tgt_sample is None else like.tgt_sample.shape[0],
                            self.config.samples if like.S else 0))
        cfg = self._opt_config(like.K > 0)
        desc = like.desc
        key = (B, n_cap, t_cap, desc, tuple(sorted(vars(cfg).items())))
        if key not in self._engines:
            self._engines.clear()                      # one resident engine at a time
            self._engines[key] = BatchedEngine(desc, cfg, B, n_cap, t_cap, self._dev())
        return self._engines[key]

    def _finish(self, eng, slot, prep, st, store=None):
        """registration.py:253-262: warp ALL source points through the optimised pyramid, add tgt_mean."""
        c = self.config
        if store is None:
            store = eng.params[slot]                                              # [m, p_stride] on device
        warped = ops.pyramid_fwd(prep.desc, c.m, c.k0, store, prep.src_centered)
        return warped + prep.tgt_mean
