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
import ctypes
import gc
import queue
import threading

import numpy as np
import torch

from . import _native as N
from . import ops
from .engine import BatchedEngine, OptConfig
from .layout import LayerDesc
from .nets import _draw_ops, _native_rng_ok, init_pyramid_store


class _Prepared:
    """Everything one pair needs on the device before it enters an engine slot: the raw clouds, their means, the
    freshly initialised pyramid and the sampling permutations (one pinned upload), optional landmarks."""
    __slots__ = ("src_pcd", "tgt_pcd", "means", "buf", "store", "perm_s", "perm_t", "K", "S", "T", "ldmk_s", "ldmk_t",
                 "desc", "result", "state", "index")

    def tensors(self):
        return [t for t in (self.src_pcd, self.tgt_pcd, self.means, self.buf, self.ldmk_s, self.ldmk_t) if t is not None]

    def load_job(self, slot):
        return dict(slot=slot, params=self.store, K=self.K, S=self.S, T=self.T, src=self.src_pcd, tgt=self.tgt_pcd,
                    perm_s=self.perm_s, perm_t=self.perm_t, ldmk_s=self.ldmk_s, ldmk_t=self.ldmk_t, means=self.means,
                    n_src=self.src_pcd.shape[0], n_tgt=self.tgt_pcd.shape[0])     # (the means are computed by the load call)

    def warp_job(self, store):
        return (store, self.src_pcd, self.means, self.means[4:])

    def release(self):
        """Drop the device staging once the final warp has been enqueued (the allocator keeps the memory alive for
        the streams the tensors were recorded on)."""
        self.buf = self.store = self.perm_s = self.perm_t = self.tgt_pcd = self.ldmk_s = self.ldmk_t = None
        self.src_pcd = self.means = None             # (the warp job that read them is enqueued; a per-pair device copy of the source
                                                     #  and a 512-byte allocation per pair otherwise live until the batch call returns)


class _PinRing:
    """Pinned float32 staging buffers of ONE preparing thread, reused once their upload has completed (allocated once:
    hipHostMalloc costs milliseconds), plus that thread's integer scratch for the permutation replay."""

    def __init__(self):
        self.free, self.busy, self._scratch = [], [], None

    def take(self, numel):
        while self.busy and self.busy[0][1].query():
            self.free.append(self.busy.pop(0)[0])
        for i, buf in enumerate(self.free):
            if buf.numel() == numel:
                return self.free.pop(i)
        if len(self.busy) >= 64:                                                    # bound the ring: wait for the oldest
            host, ev = self.busy.pop(0)
            ev.synchronize()
            if host.numel() == numel:
                return host
        return torch.empty(numel, dtype=torch.float32).pin_memory()

    def uploaded(self, host, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.busy.append((host, ev))

    def scratch(self, n):
        if self._scratch is None or self._scratch.numel() < n:
            self._scratch = torch.empty(max(n, 8192), dtype=torch.int32)
        return self._scratch


class _BatchCtx:
    """What the lanes of one register_batch call share."""
    __slots__ = ("reg", "preps", "next_prepared", "fin_stream", "main", "chunk", "m", "exhausted", "handed_out", "total_slots", "sink", "states", "fin_batches")

    def __init__(self, reg, preps, next_prepared, fin_stream, main, chunk, m):
        self.reg, self.preps, self.next_prepared = reg, preps, next_prepared
        self.fin_stream, self.main, self.chunk, self.m = fin_stream, main, chunk, m
        self.exhausted = False
        self.handed_out, self.total_slots = 0, 0          # pairs given to lanes so far / slots of all lanes (set once they exist)
        self.sink = None
        self.fin_batches = 0                              # final-warp batches handed to the sink so far
        self.states = [None] * len(preps)                 # final pair states by index (register_batch -> last_states)



_NOT_READY = object()


class _Lane:
    """One engine on one stream.  Pipelined control: the states of chunk k are read back while chunk k+1
    runs, so the GPU never waits for the host; a slot that finishes in chunk k is refilled before chunk k+2.
    Refills of one round go up in ONE launch (k_eng_load), the final all-point warps of the pairs that
    finished in one chunk in ONE launch (k_pyramid_fwd) on the shared side stream."""

    def __init__(self, ctx, eng, stream):
        self.ctx, self.eng, self.stream = ctx, eng, stream
        self.fin_done = {}                               # slot -> event: its parameters have been consumed
        self.active, self.free = {}, list(range(eng.B))  # active: slot -> (pair index, first valid snapshot)
        self.seq, self.pending, self.done = 0, None, False

    def step(self, first=None):
        ctx, eng = self.ctx, self.eng
        jobs = []
        # Refill policy.  While the batch ramps up (fewer pairs handed out than there are slots) a lane takes what the producer
        # has ready and ticks: filling every slot first kept the GPU idle for slots x 0.45 ms at the start of each batch, lane
        # after lane.  After that it waits for the producer: in a GPU-bound run the queue is never empty, and in a producer-bound
        # one (the landmark config) ticking half-empty engines only costs launches that slow the producer down (measured both ways).
        # It never waits for more than `quota` pairs per step, so the pairs already resident keep ticking.
        ramp = ctx.handed_out < ctx.total_slots
        quota = max(4, eng.B // 16)
        while self.free and not ctx.exhausted and (ramp or len(jobs) < quota):
            idle = not self.active and self.pending is None and not jobs
            nxt = first if first is not None else ctx.next_prepared(self.stream, block=idle or not ramp)
            first = None
            if nxt is None:
                ctx.exhausted = True
                break
            if nxt is _NOT_READY:
                break
            ctx.handed_out += 1
            i, p = nxt
            slot = self.free.pop()
            if slot in self.fin_done:
                self.stream.wait_event(self.fin_done.pop(slot))   # the previous tenant's parameters have been copied out for its final warp
            jobs.append(p.load_job(slot))
            self.active[slot] = (i, self.seq)            # snapshots >= seq see this pair in the slot
        if jobs:
            eng.load_jobs(jobs)
        if not self.active and self.pending is None:
            self.done = True
            return
        handle = None
        if self.active:
            eng.run_ticks(ctx.chunk)
            handle = (eng.snapshot_async(), self.seq)
            self.seq += 1
        if self.pending is not None:
            (h, hseq) = self.pending
            snap = eng.wait_snapshot(h)
            done = []
            for slot in np.nonzero(snap.level >= ctx.m)[0].tolist():   # finished (or parked) slots only
                if slot not in self.active:
                    continue
                i, valid_from = self.active[slot]
                if hseq < valid_from:
                    continue
                st = snap.state(slot)
                del self.active[slot]
                ctx.preps[i].state = st
                ctx.states[i] = st
                done.append((slot, ctx.preps[i]))
                self.free.append(slot)
            if done:
                # the snapshot proves every tick that touched these slots has completed: the final warp
                # needs no dependency on the lane's stream, only the slots' refill must wait for it
                with torch.cuda.stream(ctx.fin_stream):
                    ev = torch.cuda.Event()
                    outs = ctx.reg._finish(eng, done, freeze=True, frozen=lambda: ev.record(ctx.fin_stream))
                for (slot, p), out in zip(done, outs):
                    self.fin_done[slot] = ev
                    out.record_stream(ctx.main)
                    p.result = out
                    p.release()
                if ctx.sink is not None:
                    with torch.cuda.stream(ctx.fin_stream):           # whatever the sink enqueues is ordered behind the final warp that
                        for slot, p in done:                          # produced `warped` (it is NOT complete on the lane's stream)
                            ctx.sink(p.index, p.result, p.state)
                            p.result = None
                            ctx.preps[p.index] = None                 # a long stream holds the resident pairs only (its states: ctx.states)
                    # nobody waits on the final-warp stream in a sink stream (the sink's work is ordered on it): like the producers'
                    # side streams it is synchronised now and then, or the HIP runtime's per-command state of it grows with the stream
                    ctx.fin_batches += 1
                    if ctx.fin_batches % 64 == 0:
                        ctx.fin_stream.synchronize()
        self.pending = handle


class Registration:
    def __init__(self, config, gemm_mode=None, nn_mode=None, nn_matrix=None):
        """config: the reference's attribute-accessible config (NDP.yaml / LNDP.yaml keys).  gemm_mode / nn_mode (extension, both
        default to the engine's choice, see engine.resolve_modes): arithmetic of the level kernels' 128x128 contractions (0: fp32
        MFMA, bitwise the oracle's chain; 7: two-way fp16 splits on the fp16 MFMA), a forced shape of the nearest-neighbour kernel
        (nn_mode) or just the preference for its matrix-pipe variant where the engine picks the throughput shape (nn_matrix)."""
        self.gemm_mode, self.nn_mode, self.nn_matrix = gemm_mode, nn_mode, nn_matrix
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

    def load_raw_pcds_from_depth(self, source_depth_path, tgt_depth_path, K, landmarks=None):
        """Deformation graph + raw clouds of the embedded-deformation (N-ICP) baseline (registration.py:38-90)."""
        from .ed import load_raw_pcds_from_depth
        load_raw_pcds_from_depth(self, source_depth_path, tgt_depth_path, K, landmarks)

    def register(self, **kwargs):
        if self.deformation_model == "NDP":
            return self.optimize_deformation_pyramid(**kwargs)
        if self.deformation_model == "NSFP":                      # registration.py:112-113 -> (warped, None)
            from .nsfp import optimize_neural_SFlow
            kwargs.pop("timer", None)
            return optimize_neural_SFlow(self, **kwargs)
        if self.deformation_model == "Nerfies":                   # registration.py:118-119 -> (warped, None)
            from .nerfies import optimize_Nerfies
            kwargs.pop("timer", None)
            return optimize_Nerfies(self, **kwargs)
        if self.deformation_model == "ED":                        # registration.py:112-113 -> (warped sampled cloud, valid_id)
            from .ed import optimize_Embeded_deformation
            kwargs.pop("timer", None)
            return optimize_Embeded_deformation(self, **kwargs)
        # Sinkhorn is a comparison baseline outside this path
        raise KeyError(self.deformation_model)

    def optimize_deformation_pyramid(self, visualize=False, timer=None):
        if visualize:
            raise NotImplementedError("mayavi visualisation is outside the hot path")
        prep = self._prepare(self.src_pcd, self.tgt_pcd, self.landmarks)
        self.src_pcd = prep.src_pcd
        eng = self._engine(1, prep)
        eng.load_jobs([prep.load_job(0)])
        kernel_ms = [0.0] * len(N.TICK_KERNELS) if timer is not None else None
        st = eng.run_until_done(chunk=32, kernel_ms=kernel_ms)[0]
        warped = self._finish(eng, [(0, prep)])[0]
        self.last_state = st
        iter_cnt = {lvl: int(st.evals_per_level[lvl]) for lvl in range(self.config.m)}
        if timer is not None:
            self._fill_timer(timer, kernel_ms, st, prep.K > 0)
        return warped, iter_cnt, timer

    @staticmethod
    def _fill_timer(timer, kernel_ms, st, has_ldmk):
        """The reference tics `lvl_warp` and `Chamfer` once per loss evaluation (only without landmarks) and `backprop` once per
        Adam step (registration.py:207-213, 234-238), and eval_nolearned.py:139-146 prints those keys.  The iteration never
        returns to the host here, so the pair's DEVICE time per kernel (HIP events around every launch, ndp_engine_run_timed) is
        apportioned: lvl_warp = level forward, Chamfer = nearest neighbours + loss / early-stop decision / dL/dx', backprop =
        the two backward kernels + gradient fold and Adam -- recorded as one tictoc per evaluation / step, so that `calls` and
        `average` mean what they mean upstream."""
        ms = dict(zip(N.TICK_KERNELS, kernel_ms))
        groups = [("backprop", (ms["k_eng_bwd2"] + ms["k_eng_bwd1"] + ms["k_eng_update"]) * 1e-3, int(st.total_steps))]
        if not has_ldmk:
            groups = [("lvl_warp", ms["k_eng_fwd"] * 1e-3, int(st.total_evals)),
                      ("Chamfer", (ms["k_eng_nn"] + ms["k_eng_loss"]) * 1e-3, int(st.total_evals))] + groups
        for key, total, calls in groups:
            for _ in range(calls):
                timer.tictoc(key, total / calls)

    # ------------------------------------------------------------------ batched extension
    def register_batch(self, pairs, slots=64, chunk=8, prefetch=True, engines=1, workers=3, sink=None):
        """pairs: sequence of (src, tgt) or (src, tgt, (ldmk_s, ldmk_t)).  Pairs are prepared in order
        (so the CPU RNG stream is consumed exactly as by sequential register() calls) and optimised
        `slots` at a time per engine, finished slots being refilled.  With prefetch=True the host-side preparation
        (RNG replay of the init and of the sampling permutations) runs ahead of the GPU in `workers` producer threads fed by
        one generator-stepping thread (see below: bit-identical to the sequential order whatever the thread count).
        engines > 1 keeps that many independent engines ticking on their own HIP streams: their launches interleave on
        the GPU, so the VALU-bound and latency-bound kernels of one overlap the MFMA-bound kernels of another.
        Returns [(warped, iter_cnt)] in input order.
        sink(i, warped, state): called (on the calling thread, in completion order) for every finished pair INSTEAD of keeping its
        result -- a long stream of pairs then holds no more than the resident ones; the call returns None.  The sink runs with the
        final-warp stream current: GPU work it enqueues on `warped` is ordered behind the kernel that writes it; a reference it keeps
        is safe to use once the call has returned."""
        pairs = list(pairs)
        if not pairs:
            return []
        # one engine configuration serves the whole batch: capacities from the LARGEST landmark set (landmarks are known
        # before any preparation), and the objective (w_cd / trunc_cd, registration.py:189-212) must be the same for all
        ks = [int(item[2][0].shape[0]) if len(item) > 2 and item[2] is not None else 0 for item in pairs]
        if min(ks) == 0 and max(ks) > 0:
            raise ValueError("register_batch: pairs with and without landmarks use different objectives "
                             "(registration.py:189-212); register them in separate batches")
        k_max = max(ks)
        engines = max(1, min(int(engines), len(pairs)))
        preps = [None] * len(pairs)
        dev = self._dev()
        fin_stream = self._stream("fin", dev)                    # final all-point warps overlap the ticking engines
        # Host-side preparation (the RNG replay of the reference's init and of its two randperm calls, registration.py:133-159)
        # runs ahead of the GPU.  The draw counts per pair are known up front, so ONE stepper thread walks torch's CPU generator
        # from pair to pair (regenerations only) and hands each pair the generator state it starts from; `workers` threads replay
        # their pairs from those snapshots natively (GIL released) on their own side streams -- bit-identical to sequential
        # register() calls, in any completion order.  Pair i is prepared by worker i % W and delivered in index order.
        W = max(1, int(workers)) if prefetch else 0
        native = prefetch and _native_rng_ok()
        if not native:
            W = min(W, 1)                                        # the torch-call replay consumes the global generator: one thread
        depth_q = max(2 * slots * engines // max(W, 1), 4)
        out_q = [queue.Queue(maxsize=depth_q) for _ in range(max(W, 1))]
        in_q = [queue.Queue(maxsize=8) for _ in range(W)] if native else []
        stop = threading.Event()

        def put(q, item):
            """Bounded put that gives up when the consumer has failed (so a thread never stays blocked)."""
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def as_tensors(item):
            src, tgt = item[0], item[1]
            ldmk = item[2] if len(item) > 2 else None
            if isinstance(src, np.ndarray):
                src, tgt = torch.from_numpy(src), torch.from_numpy(tgt)
            return src, tgt, ldmk

        def prepare_on(side, ring, item, rng_state):
            src, tgt, ldmk = as_tensors(item)
            with torch.cuda.stream(side):
                p = self._prepare(src.to(dev), tgt.to(dev), ldmk, rng_state=rng_state, ring=ring)
                ev = torch.cuda.Event()
                ev.record(side)
            return p, ev

        def stepper():
            """Walks the generator over all pairs; pair i starts from the snapshot taken before its draws."""
            try:
                L = N.host_lib()
                st = torch.get_rng_state()
                for i, item in enumerate(pairs):
                    if stop.is_set():
                        return
                    n_src, n_tgt = int(item[0].shape[0]), int(item[1].shape[0])
                    if not put(in_q[i % W], (i, item, st.clone())):
                        return
                    if L.ndp_rng_skip(ctypes.c_void_p(st.data_ptr()), st.numel(), self._pair_draws(n_src, n_tgt)) != 0:
                        raise N.NdpError("ndp_rng_skip failed")
                torch.set_rng_state(st)                           # where sequential register() calls would have left it
                for q in in_q:
                    put(q, None)
            except BaseException as e:                            # surface the failure in the consumer
                for q in out_q:
                    put(q, e)

        def worker(w):
            try:
                side, ring = self._stream(("side", w), dev), self._pin_ring(w)
                while not stop.is_set():
                    try:
                        task = in_q[w].get(timeout=0.1)
                    except queue.Empty:
                        continue
                    if task is None:
                        put(out_q[w], None)
                        return
                    i, item, st = task
                    p, ev = prepare_on(side, ring, item, st)
                    # The host never waits on this stream otherwise (lanes wait on its events), and the HIP runtime keeps per-command
                    # state of a stream until the host synchronises it: a long stream of pairs grew the resident set by ~4.5 KB per
                    # pair (tools/stream_memory.py: 4.6 -> 1.2 KB with this).  The producer runs ahead of the GPU: the wait is idle time.
                    if (i // W) % 64 == 63:
                        side.synchronize()
                    if not put(out_q[w], (i, p, ev)):
                        return
            except BaseException as e:
                put(out_q[w], e)

        def produce_sequential():
            """One thread on the global generator (no native replay available)."""
            try:
                side, ring = self._stream(("side", 0), dev), self._pin_ring(0)
                for i, item in enumerate(pairs):
                    if stop.is_set():
                        return
                    p, ev = prepare_on(side, ring, item, None)
                    if i % 64 == 63:
                        side.synchronize()                        # (see worker())
                    if not put(out_q[0], (i, p, ev)):
                        return
                put(out_q[0], None)
            except BaseException as e:
                put(out_q[0], e)

        threads = []
        if native:
            threads = [threading.Thread(target=stepper, daemon=True)] + [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(W)]
        elif prefetch:
            threads = [threading.Thread(target=produce_sequential, daemon=True)]
        for th in threads:
            th.start()
        cursor = [0]                                             # index of the next pair to hand out

        def next_prepared(stream, block=True):
            """Next pair (in index order) made visible to `stream` (a lane's stream) and to fin_stream.
            block=False: _NOT_READY when its worker has not finished it yet."""
            i = cursor[0]
            if i >= len(pairs):
                return None
            if not prefetch:                                     # in-line preparation on the caller's stream
                src, tgt, ldmk = as_tensors(pairs[i])
                item = (i, self._prepare(src.to(dev), tgt.to(dev), ldmk), None)
            else:
                try:
                    item = out_q[i % max(W, 1)].get(block=block)
                except queue.Empty:
                    return _NOT_READY
            if item is None:
                return None
            if isinstance(item, BaseException):
                raise item
            i, p, ev = item
            cursor[0] = i + 1
            if ev is not None:
                stream.wait_event(ev)
                fin_stream.wait_event(ev)
            for t in p.tensors():                                # allocated on the producer's stream, consumed on these two
                t.record_stream(stream)
                t.record_stream(fin_stream)
            preps[i] = p
            p.index = i
            return i, p

        m = self.config.m
        main = torch.cuda.current_stream(dev)
        ctx = _BatchCtx(self, preps, next_prepared, fin_stream, main, chunk, m)
        ctx.sink = sink

        # the cyclic collector's FULL passes over thousands of live pair objects stalled every lane for 50-85 ms a few times per
        # batch (rocprofv3 trace of the bench); nothing in the loop builds reference cycles worth collecting before it ends.  Only
        # the oldest generation is held back for the duration of the call -- young collections (cheap, what other threads of the
        # application may rely on) keep running.
        gc_thr = gc.get_threshold()
        gc.set_threshold(gc_thr[0], gc_thr[1], 1 << 30)
        try:
            first = next_prepared(main)
            B = min(slots, -(-len(pairs) // engines))
            ctx.total_slots = B * engines
            lanes = []
            for e in range(engines):
                stream = main if engines == 1 else self._stream(("lane", e), dev)
                eng = self._engine(B, first[1], n_hint=(self.config.samples if first[1].S else 0) + k_max, lane=e)
                with torch.cuda.stream(stream):
                    stream.wait_stream(main)
                    eng.park_all()
                lanes.append(_Lane(ctx, eng, stream))
            if engines > 1:                                          # the first pair was made visible to `main` only
                for t in first[1].tensors():
                    t.record_stream(lanes[0].stream)
                lanes[0].stream.wait_stream(main)
            while not all(lane.done for lane in lanes):
                for lane in lanes:
                    if lane.done:
                        continue
                    with torch.cuda.stream(lane.stream):
                        lane.step(first)
                    first = None
        except BaseException:
            # a failing lane must not leave the producer blocked on the bounded queue, holding device tensors and
            # pinned buffers: stop it, drain what it queued, join it, and let every stream finish what was enqueued
            stop.set()
            for q in out_q + in_q:
                while True:
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        break
            for th in threads:
                th.join()
            torch.cuda.synchronize(dev)
            ctx.preps = ctx.next_prepared = None
            raise
        finally:
            gc.set_threshold(*gc_thr)
        for th in threads:
            th.join()
        for lane in lanes:
            main.wait_stream(lane.stream)
        main.wait_stream(fin_stream)
        self.last_states = ctx.states
        results = None if sink is not None else [(p.result, {lvl: int(p.state.evals_per_level[lvl]) for lvl in range(m)}) for p in preps]
        ctx.preps = ctx.next_prepared = None                     # nothing of this call stays reachable but the results
        return results

    # ------------------------------------------------------------------ internals
    def _stream(self, key, dev):
        if not hasattr(self, "_streams"):
            self._streams = {}
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(dev)
        return self._streams[key]

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

    def _init_ops(self, descs, depth, stride):
        """ctypes draw ops of one pair's pyramid initialisation (cached per configuration)."""
        key = (tuple(descs), depth, stride)
        cache = self.__dict__.setdefault("_ops_cache", {})
        if key not in cache:
            cache[key] = N.make_draw_ops(_draw_ops(descs, depth, stride))
        return cache[key]

    def _pair_descs(self):
        c = self.config
        gate = c.w_reg > 0                                                          # registration.py:138
        desc = LayerDesc(width=c.width, n_hidden=c.depth - 1, motion=c.motion_type, rotfmt=c.rotation_format, nonrigidity=gate)
        level0 = LayerDesc(width=c.width, n_hidden=c.depth - 1, motion=c.motion_type, rotfmt=c.rotation_format)
        return desc, [level0] + [desc] * (c.m - 1)                                  # nets.py:26: level 0 never carries the gate

    def _pair_draws(self, n_src, n_tgt):
        """Raw generator draws one pair consumes: pyramid init + randperm(n_src) + randperm(n_tgt)."""
        desc, descs = self._pair_descs()
        ops_ = self._init_ops(descs, self.config.depth, (desc.param_count + 63) // 64 * 64)
        return int(N.host_lib().ndp_pair_draws(ops_, len(ops_), int(n_src), int(n_tgt)))

    def _pin_ring(self, key):
        rings = self.__dict__.setdefault("_pin_rings", {})
        if key not in rings:
            rings[key] = _PinRing()
        return rings[key]

    def _prepare(self, src_pcd, tgt_pcd, landmarks, rng_state=None, ring=None):
        """Host side of registration.py:133-164 for one pair: RNG replay of the pyramid initialisation and of the two
        sampling permutations into ONE pinned buffer, one asynchronous upload, one launch for the two cloud means.
        Centring, sampling and the slot fill itself happen on the device (k_eng_load).
        rng_state: a snapshot of torch's CPU generator state this pair starts from (the batched producer's workers; the global
        generator is then left alone), None: consume the global generator like the reference does.  ring: pinned staging ring."""
        c = self.config
        # (upstream fails deep inside knn_points / mean() on such inputs; say what is wrong instead)
        for name, cloud in (("source", src_pcd), ("target", tgt_pcd)):
            if not torch.is_tensor(cloud) or cloud.ndim != 2 or cloud.shape[1] != 3 or cloud.shape[0] < 1:
                raise ValueError(f"{name} cloud must be a [N, 3] tensor with N >= 1, got "
                                 f"{tuple(cloud.shape) if torch.is_tensor(cloud) else type(cloud).__name__}")
        if landmarks is not None:
            ls, lt = landmarks
            if ls.ndim != 2 or ls.shape[1] != 3 or tuple(ls.shape) != tuple(lt.shape) or ls.shape[0] < 1:
                raise ValueError(f"landmarks must be two [K, 3] tensors with the same K >= 1, got {tuple(ls.shape)} and {tuple(lt.shape)}")
        dev = self._dev()
        if not (1 <= c.width <= 256 and 1 <= c.depth <= 4):        # (128 / 3: the MFMA kernels; the rest: csrc/ndp_generic.inc)
            raise N.NdpError(f"width must be 1..256 and depth 1..4, got width={c.width}, depth={c.depth}")
        p = _Prepared()
        # registration.py:133-140 -- all m levels are initialised up front on the CPU generator
        p.desc, descs = self._pair_descs()                                          # engine: "levels > 0 gated"
        stride = (p.desc.param_count + 63) // 64 * 64
        n_par = c.m * stride
        samples = int(c.samples)
        ring = ring if ring is not None else self._pin_ring("main")
        host = ring.take(n_par + 2 * samples)                                      # reused pinned staging buffer
        src_pcd = src_pcd.to(dev, non_blocking=True).float().contiguous()
        tgt_pcd = tgt_pcd.to(dev, non_blocking=True).float().contiguous()
        p.src_pcd, p.tgt_pcd = src_pcd, tgt_pcd
        n_src, n_tgt = src_pcd.shape[0], tgt_pcd.shape[0]
        ns, nt = min(samples, n_src), min(samples, n_tgt)
        hi = host[n_par:].view(torch.int32)
        if _native_rng_ok():
            # init draws + both permutation prefixes in ONE native call from a generator-state snapshot (GIL released)
            own = rng_state is None
            st = torch.get_rng_state() if own else rng_state
            ops_ = self._init_ops(descs, c.depth, stride)
            store = host[:n_par].view(c.m, stride)
            for i, d in enumerate(descs):
                store[i, d.param_count:] = 0.0
            scratch = ring.scratch(max(n_src, n_tgt))
            L = N.host_lib()
            rc = L.ndp_pair_init(ctypes.c_void_p(st.data_ptr()), st.numel(), ops_, len(ops_), ctypes.c_void_p(host.data_ptr()),
                                 n_src, n_tgt, samples, ctypes.c_void_p(hi.data_ptr()), ctypes.c_void_p(hi[samples:].data_ptr()),
                                 ctypes.c_void_p(scratch.data_ptr()))
            if rc != 0:
                raise N.NdpError("ndp_pair_init failed")
            if own:                                                                 # leave the global generator where torch would
                if L.ndp_rng_skip(ctypes.c_void_p(st.data_ptr()), st.numel(), L.ndp_pair_draws(ops_, len(ops_), n_src, n_tgt)) != 0:
                    raise N.NdpError("ndp_rng_skip failed")
                torch.set_rng_state(st)
        else:
            if rng_state is not None:
                raise N.NdpError("a generator-state snapshot needs the native RNG replay (libndp_host.so)")
            init_pyramid_store(descs, c.depth, stride, out=host[:n_par].view(c.m, stride))   # nets.py:26
            perm_s = torch.randperm(n_src)                                          # :156-159 (CPU RNG)
            perm_t = torch.randperm(n_tgt)
            hi[:ns] = perm_s[:ns]
            hi[samples:samples + nt] = perm_t[:nt]
        p.buf = host.to(dev, non_blocking=True)                                    # async upload on the current stream
        ring.uploaded(host, torch.cuda.current_stream(dev))
        p.store = p.buf[:n_par].view(c.m, stride)
        di = p.buf[n_par:].view(torch.int32)
        p.perm_s, p.perm_t = di[:ns], di[samples:samples + nt]
        p.means = torch.empty(8, device=dev, dtype=torch.float32)                 # :150-153: filled by the slot's load call (ONE launch per
                                                                                   # group of up to 16 pairs instead of one per pair)
        p.ldmk_s = p.ldmk_t = None
        if landmarks is not None:
            p.ldmk_s = landmarks[0].to(dev).float().contiguous()                  # :162-164 (centred on the device)
            p.ldmk_t = landmarks[1].to(dev).float().contiguous()
            p.K = p.ldmk_s.shape[0]
            if c.w_cd > 0:
                p.S, p.T = ns, nt                                                  # :190
            else:
                p.S, p.T = 0, 0
        else:
            p.K, p.S, p.T = 0, ns, nt
        p.result = p.state = None
        return p

    def _engine(self, B, like, n_hint=0, lane=0):
        n_cap = ops.cap(max(like.K + like.S, n_hint))
        t_cap = ops.cap(max(like.T, self.config.samples if like.S else 0))
        cfg = self._opt_config(like.K > 0)
        desc = like.desc
        key = (B, n_cap, t_cap, desc, tuple(sorted(vars(cfg).items())), self.gemm_mode, self.nn_mode, self.nn_matrix)
        if self._engines.get("key") != key:
            self._engines.clear()                      # one resident engine configuration at a time
            self._engines["key"] = key
        if lane not in self._engines:
            self._engines[lane] = BatchedEngine(desc, cfg, B, n_cap, t_cap, self._dev(), gemm_mode=self.gemm_mode, nn_mode=self.nn_mode, nn_matrix=self.nn_matrix)
        return self._engines[lane]

    def _finish(self, eng, done, freeze=False, frozen=None):
        """registration.py:253-262 for every (slot, prepared pair) of `done`, in one launch: ALL source points through
        the optimised pyramid (centring by the source mean and adding the target mean happen in the kernel).
        freeze=True first snapshots the slots' parameters -- ONE gathering copy for all of them (round 6; until then a clone per
        pair) -- so that the slots can be refilled at once: `frozen()` is called between the copy and the warp launch (the batched
        lanes record the event their refills wait for there: a refill waits for the copy, not for the warp).  The batched path
        (freeze) also takes the warp's throughput shape, eight tiles per workgroup."""
        c = self.config
        split = bool(eng.gemm_mode & 1)
        if freeze:
            stores = torch.stack([eng.params[slot] for slot, _ in done])          # [len(done), m, p_stride]
            if frozen is not None:
                frozen()
            jobs = [prep.warp_job(stores[k]) for k, (_, prep) in enumerate(done)]
        else:
            jobs = [prep.warp_job(eng.params[slot]) for slot, prep in done]
        # the final warp runs in the engine's arithmetic: fp16-split contractions with gemm_mode & 1, the fp32 MFMA otherwise
        return ops.pyramid_fwd_batch(done[0][1].desc, c.m, c.k0, jobs, device=self._dev(), split=split,
                                     tiles=8 if (freeze and split) else None)
