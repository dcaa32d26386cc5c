#!/usr/bin/env python3
"""bench.py -- pairs/sec (+ ms/iteration) of NDP registration on synthetic 8192-point pairs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` (no launcher, WORLD_SIZE unset) starts the N ranks itself: it re-executes this file under
`torch.distributed.run --nproc-per-node N` on 127.0.0.1, one rank per GPU, backend nccl (= RCCL); rank 0 prints the line.

One *step* registers `--pairs-per-step` (default 32 x `--slots`) synthetic 8192-point pairs per GPU,
`--slots` (default 256) per engine of them resident on the device at any time, with the shipped
NDP.yaml settings (SE3 / axis-angle, m = 9 levels, 2000 samples per cloud, lr 0.01, early stop on):
per pair the full Registration.register() work -- pyramid init, centring, sampling, the level/Adam
loop on the device, and the final warp of all 8192 source points.  The point clouds are resident in
HBM before the timed region.  Pairs shard over ranks (weak scaling: the same number of pairs per
GPU); the only collective is the final all-reduce of the aggregate (RCCL).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement / DESIGN.md).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before HIP initialises: one hardware queue per busy stream (see the package __init__)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deformationpyramid_amd import _native as N                  # noqa: E402
from deformationpyramid_amd.config import load_config            # noqa: E402
from deformationpyramid_amd.loss import compute_flow_metrics     # noqa: E402
from deformationpyramid_amd.parallel import job_summary, pin_rank_to_gpu_numa   # noqa: E402
from deformationpyramid_amd.registration import Registration     # noqa: E402
from deformationpyramid_amd.config import Config                 # noqa: E402
from deformationpyramid_amd.synthetic import surface_pair, synthetic_landmarks, synthetic_pair      # noqa: E402

FP32_PEAK_TFLOPS = 157.3          # MI355X fp32 matrix = vector peak (MI355X_MICROARCH.md)
BF16_PEAK_TFLOPS = 2500.0         # dense 16-bit (bf16 = fp16) MFMA peak (same guide)
SPLIT_PRODUCTS = 3                # fp16 partial products per fp32-equivalent product in the split kernels (hi.hi, hi.lo, lo.hi); rounds 2-3: six bf16 ones
FLOP_FWD_PT = 68608
FLOP_BWD_PT = 135680
FLOP_NN_PAIR = 8                  # per (source, target) distance evaluation; one pass serves both directions


def algorithmic_flops(S, T, P):
    """F(S,T) of one pair-iteration (SURVEY.md section 8d / BASELINE.md section 3)."""
    return (FLOP_FWD_PT + FLOP_BWD_PT) * S + FLOP_NN_PAIR * S * T + 12 * P


def kernel_profile(model, pairs, slots, n_ticks=24):
    """Per-kernel average launch durations (HIP events on the launch stream) over ticks in which
    every slot is active at level 0.  Returns dict kernel -> ms per launch, plus the engine."""
    preps = [model._prepare(it[0], it[1], it[2] if len(it) > 2 else None) for it in pairs[:slots]]
    eng = model._engine(len(preps), preps[0])
    for b, p in enumerate(preps):
        eng.load_jobs([p.load_job(b)])
    eng.run_ticks(4)                                  # warm-up ticks
    ms = eng.run_ticks_timed(n_ticks)
    st = eng.read_states()
    active = sum(1 for s in st if s.level == 0)
    from deformationpyramid_amd._native import TICK_KERNELS
    return {k: v / n_ticks for k, v in zip(TICK_KERNELS, ms)}, eng, preps, active


def _git_blob_hash(path):
    """`git hash-object` of a file without git (the GPU box has no .git): sha1("blob <len>\\0" + content)."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_traffic(kernels, pairs):
    """HBM bytes per launch of the kernel(s) behind one tick stage, from the rocprofv3 PMC passes kept under profiles/
    (FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as is; collected by tools/pmc_traffic.sh at 128 pairs per launch
    and scaled linearly to this launch's pair count).  -> (bytes or None, source description or None): the figure is NOT
    measured in this run -- `traffic_source` names the committed profile it comes from and that file's git blob hash."""
    try:
        names = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc.json") and "_hbm_traffic" in p and "_mode" not in p and "2launch" not in p)
        rnd = names[-1][:3] if names else ""                                     # newest round's files only
        best = None
        for name in names:
            if not name.startswith(rnd):
                continue
            path = os.path.join(ROOT, "profiles", name)
            rec = json.load(open(path))
            if all(k in rec for k in kernels):
                # the collection whose launch geometry is closest to this one (the 256-pair engines run G = 1: the backward carries the
                # Adam tail there and not at 128 pairs), scaled linearly to the active pair count
                d = max(abs(rec[k]["pairs_per_launch"] - pairs) for k in kernels)
                if best is None or d < best[0]:
                    best = (d, name, path, rec)
        if best is not None:
            _, name, path, rec = best
            tot = sum(rec[k]["hbm_bytes_per_launch"] * pairs / rec[k]["pairs_per_launch"] for k in kernels)
            return tot, {"file": "profiles/" + name, "git_blob": _git_blob_hash(path), "kernels": list(kernels),
                         "pairs_per_launch_of_the_collection": rec[kernels[0]]["pairs_per_launch"],
                         "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_traffic.sh), FETCH_SIZE doubled"}
    except (OSError, KeyError, ValueError, IndexError):
        pass
    return None, None


def adam_in_tail(gemm_mode, G):
    """G = 1 engines (round 6): the fused backward steps the two 128 x 128 matrices behind its tile loop (csrc/ndp_bwd_fused.inc:
    bf_adam_in_tail), k_eng_update_rest the other parameters; gemm_mode bit 1024 keeps the whole step in k_eng_update."""
    return G == 1 and (gemm_mode & 7) == 7 and not gemm_mode & (16 | 64 | 256 | 1024)


def stage_kernels(gemm_mode, nn_mode, G=0):
    """The launches behind each of the six tick stages (N.TICK_KERNELS order) for an engine's modes.  (The per-point warp of the split
    forward runs behind that kernel's tile loop since round 6; gemm_mode bit 512 keeps it a launch of its own.)"""
    fused = bwd_fused(gemm_mode)
    return {"k_eng_fwd": (["k_eng_fwd8", "k_eng_warp"] if gemm_mode & 512 else ["k_eng_fwd8"]) if gemm_mode & 1 else ["k_eng_fwd"],
            "k_eng_nn": [{0: "k_eng_nn", 1: "k_eng_nn_lat" if gemm_mode & 128 else "k_eng_nn_lat8", 2: "k_eng_nn_mx" if gemm_mode & 128 else "k_eng_nn_mx8"}[nn_mode]],
            "k_eng_loss": ["k_eng_loss"],
            "k_eng_bwd2": ["k_eng_bwd_f"] if fused else (["k_eng_bwd2_8"] if gemm_mode & 4 else ["k_eng_bwd2"]),
            "k_eng_bwd1": [] if fused else (["k_eng_bwd1_8"] if gemm_mode & 2 else ["k_eng_bwd1"]),
            "k_eng_update": ["k_eng_update_rest"] if adam_in_tail(gemm_mode, G) else ["k_eng_update"]}


def bwd_fused(gemm_mode):
    """Both backward layers of the split arithmetic in ONE launch (k_eng_bwd_f, round 4) -- stage 4 of the tick then launches nothing."""
    return (gemm_mode & 7) == 7 and not gemm_mode & 16


def roofline_report(model, pairs, B, config):
    """Roofline of the dominant kernel of one engine's tick (HIP events on the launch stream, all slots active at level 0),
    priced against the peak of the pipe that kernel's contractions run on: the fp32 MFMA / vector peak for the bitwise kernels
    and the nearest-neighbour kernels, the dense 16-bit MFMA peak / 3 for the split level kernels (three fp16 products per
    fp32-equivalent product: algorithmic FLOP stay SURVEY 8(d)'s)."""
    prof, eng, preps, active = kernel_profile(model, pairs, B)
    S, T, n = preps[0].S, preps[0].T, preps[0].S + preps[0].K       # n: points through the MLP (landmarks + samples)
    P = eng.P
    names = stage_kernels(eng.gemm_mode, eng.nn_mode, eng.G)
    fused = bwd_fused(eng.gemm_mode)
    tail = adam_in_tail(eng.gemm_mode, eng.G)
    if fused:                      # the empty stage's slot holds two event records back to back (~5 us), not a kernel: not part of the tick
        prof = {k: v for k, v in prof.items() if k != "k_eng_bwd1"}
    dom = max(prof, key=prof.get)
    nh = preps[0].desc.n_heads                                       # head rows: 6 SE3, 7 Sim3 (+768 FLOP/pt)
    # backward split by layer: bwd2 = heads (dWh + dh2) + dW2 + dh1, bwd1 = dW1 + dh0 + dW0 (768 MAC)
    flops = {"k_eng_fwd": (FLOP_FWD_PT + 256 * (nh - 6)) * n, "k_eng_bwd2": 2 * (2 * 16384 + 256 * nh) * n,
             "k_eng_bwd1": 2 * (2 * 16384 + 768) * n, "k_eng_nn": FLOP_NN_PAIR * S * T,
             "k_eng_update": 12 * P, "k_eng_loss": 4 * (S + T)}
    if fused:
        flops["k_eng_bwd2"] += flops.pop("k_eng_bwd1")
    split = {"k_eng_fwd": eng.gemm_mode & 1, "k_eng_bwd1": eng.gemm_mode & 2, "k_eng_bwd2": eng.gemm_mode & 4}
    peak_of = {k: (BF16_PEAK_TFLOPS / SPLIT_PRODUCTS if split.get(k) else FP32_PEAK_TFLOPS) for k in prof}
    ach = flops[dom] * active / (prof[dom] * 1e-3) / 1e12
    tick_ms = sum(prof.values())
    traffic, src = pmc_traffic(names[dom], active) if config == "A" else (None, None)
    roof = {"bound": "mfma", "kernel": "+".join(names[dom]), "achieved": ach, "peak": peak_of[dom], "unit": "TFLOP/s",
            "frac": ach / peak_of[dom], "traffic": traffic, "traffic_source": src,
            "peak_is": ("dense fp16 MFMA peak / 3 (three fp16 partial products per fp32-equivalent product; rounds 2-3 priced six bf16 ones against / 6)" if split.get(dom)
                        else "fp32 MFMA = fp32 vector peak"),
            "avg_launch_ms": prof[dom], "pairs_per_launch": active, "algorithmic_flop_per_pair_launch": flops[dom]}
    if tail and dom == "k_eng_bwd2":
        # The dominant launch also carries the Adam step of 2 x 128 x 128 parameters per pair (p, m, v read and written: an HBM-bound
        # tail of ~25 us that used to be most of k_eng_update).  `frac` above prices the WHOLE launch against the MFMA peak; for the
        # contractions alone the same tick is timed once more with the step kept in k_eng_update (gemm_mode | 1024, bitwise the same state).
        roof["adam_tail"] = {"bytes_per_pair": 6 * 4 * 2 * 128 * 128, "note": "k_eng_bwd_f steps W1 and W2 behind its tile loop at G = 1; "
                             "k_eng_update_rest steps the other parameters"}
        try:
            m2 = Registration(model.config, gemm_mode=eng.gemm_mode | 1024, nn_mode=model.nn_mode, nn_matrix=model.nn_matrix)
            prof2, eng2, _, active2 = kernel_profile(m2, pairs, B)
            a2 = flops[dom] * active2 / (prof2[dom] * 1e-3) / 1e12
            roof["adam_tail"].update({"contractions_only_ms": prof2[dom], "contractions_only_frac": a2 / peak_of[dom],
                                      "update_stage_ms_without_tail": prof2["k_eng_update"], "tick_ms_without_tail": sum(v for k, v in prof2.items() if k != "k_eng_bwd1")})
            del eng2, m2
        except Exception as ex:                        # (a measurement aid: never fails the line)
            roof["adam_tail"]["error"] = str(ex)
    if traffic:                                   # the other ceiling, for the record: HBM bytes/s of the same kernel vs 8 TB/s
        roof["hbm_tbps"] = traffic / (prof[dom] * 1e-3) / 1e12
        roof["hbm_frac"] = roof["hbm_tbps"] / 8.0
    per_kernel = {"+".join(names[k]): {"ms": prof[k], "achieved_tflops": flops[k] * active / (prof[k] * 1e-3) / 1e12,
                                        "frac_of_its_peak": flops[k] * active / (prof[k] * 1e-3) / 1e12 / peak_of[k]}
                  for k in ("k_eng_fwd", "k_eng_nn", "k_eng_bwd2", "k_eng_bwd1") if k in prof}
    # the tick's own roofline: every stage at the peak of the pipe it runs on (contractions: FLOP / MFMA or vector peak; the update:
    # its algorithmic bytes -- G gradient partials + parameters, two moments read and written -- at 8 TB/s), summed, against the
    # measured tick.  The per-kernel table decides nothing here: a tie between two kernels cannot flip this fraction.
    ideal = {k: flops[k] * active / (peak_of[k] * 1e12) * 1e3 for k in ("k_eng_fwd", "k_eng_nn", "k_eng_bwd2", "k_eng_bwd1") if k in prof}
    ideal["k_eng_update"] = ((6 if tail else eng.G + 7) * 4 * P) * active / 8e12 * 1e3   # (tail: the matrices' gradient never leaves the chip)
    ideal_ms = sum(ideal.values())
    return {"roofline": roof, "kernels_ms_per_tick": {"+".join(names[k]): v for k, v in prof.items()}, "kernel_rooflines": per_kernel,
            "tick": {"ms": tick_ms, "achieved_tflops": (algorithmic_flops(n, 0, P) + FLOP_NN_PAIR * S * T) * active / (tick_ms * 1e-3) / 1e12,
                     "ideal_ms": ideal_ms, "frac": ideal_ms / tick_ms,
                     "ideal_ms_by_stage": {"+".join(names[k]): v for k, v in ideal.items()},
                     "ideal_is": "sum over stages of algorithmic FLOP / peak of the stage's pipe (split level kernels: 2500 / 3 TFLOP/s; "
                                 "nearest neighbours: 157.3), update (wherever it runs): algorithmic bytes / 8 TB/s; loss / decision stage: 0"},
            "engine_modes": {"gemm_mode": eng.gemm_mode, "nn_mode": eng.nn_mode, "G": eng.G}}


ARITH_TEXT = {0: "fp32 MFMA, bitwise the oracle's fma chain",
              7: "128x128 contractions of the level kernels as two-way fp16 splits (2^k x = hi + lo, three partial products into one fp32 "
                 "accumulator; activations and weights scaled by 2^6, gradient operands by a power of two per pair) on the fp16 MFMA: "
                 "as close to float64 as the fp32 chain (0.5-2x its RMS error per tensor, tests/test_split_accuracy.py), not bitwise the chain"}
# `dtype` of the line: fp32 storage, fp32 accumulation and fp32-level accuracy in both arithmetics; the split one says how its products are formed
DTYPE_TEXT = {0: "f32", 7: "f32 (fp16x2-split contractions, fp32 accumulate)"}
NN_TEXT = {0: "one pass, distances on the vector pipe", 1: "latency shape (two passes, 64-query workgroups)",
           2: "one pass, distances on the bf16 matrix pipe + exact re-evaluation (bit-identical results), 512 targets per 8-wave workgroup"}


def latency_profile(cfg, pairs, repeats=5, gemm_mode=None):
    """Batch 1 -- what the reference API is (one pair per register() call, /root/reference/eval_nolearned.py:89-93):
    wall time of Registration.register() on single 8192-pt pairs, and the per-kernel split of one tick at B = 1."""
    model = Registration(cfg, gemm_mode=gemm_mode)
    torch.manual_seed(0)
    dev = model._dev()
    walls, iters = [], []
    for r in range(repeats + 1):
        src, tgt = pairs[r % len(pairs)]
        model.load_pcds(src, tgt)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        model.register()
        torch.cuda.synchronize(dev)
        if r:                                          # the first call builds the engine
            walls.append(time.perf_counter() - t0)
            iters.append(model.last_state.total_steps)
    prep = model._prepare(pairs[0][0], pairs[0][1], None)
    eng = model._engine(1, prep)
    eng.load_jobs([prep.load_job(0)])
    eng.run_ticks(4)
    n_ticks = 24
    ms = eng.run_ticks_timed(n_ticks)
    from deformationpyramid_amd._native import TICK_KERNELS
    med = sorted(range(len(walls)), key=lambda i: walls[i])[len(walls) // 2]
    return {"ms_per_pair": 1e3 * walls[med], "adam_iters": int(iters[med]), "ms_per_iter": 1e3 * walls[med] / max(iters[med], 1),
            "ms_per_pair_all": [round(1e3 * w, 2) for w in walls], "workgroups_per_level_kernel": eng.G,
            "kernels_ms_per_tick": {k: v / n_ticks for k, v in zip(TICK_KERNELS, ms)},
            "tick_ms": sum(ms) / n_ticks, "gemm_mode": eng.gemm_mode, "nn_mode": eng.nn_mode}


def _cpu_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    return model, physical, logical


def _cgroup_quota():
    """The container's CPU quota in cores (cgroup v2 cpu.max: "max" or "<quota> <period>" microseconds; v1: cfs_quota / cfs_period), or None."""
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q[0] == "max" else float(q[0]) / float(q[1])
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _cpu_pair(cfg, src, tgt, k, threads, kinds=("port-c", "port-torch")):
    """One full register() of the reference's per-pair path on the host, by each of the parity-pinned ports -> {kind: (seconds, Adam steps)}."""
    from oracle import ndp_oracle as O
    from oracle import ndp_torch_ref as T
    from deformationpyramid_amd.nets import Deformation_Pyramid
    torch.manual_seed(k)
    pyr = Deformation_Pyramid(depth=cfg.depth, width=cfg.width, device="cpu", k0=cfg.k0, m=cfg.m,
                              rotation_format=cfg.rotation_format, motion=cfg.motion_type)
    d = pyr.descs[0]
    cd = O.make_desc(d.width, d.n_hidden, d.motion, d.rotfmt, d.nonrigidity, d.mlp_scale)
    src_c = src.cpu() - src.cpu().mean(0, keepdim=True)
    tgt_c = tgt.cpu() - tgt.cpu().mean(0, keepdim=True)
    s = src_c[torch.randperm(src_c.shape[0])[: cfg.samples]].contiguous()
    t = tgt_c[torch.randperm(tgt_c.shape[0])[: cfg.samples]].contiguous()
    out = {}
    if "port-c" in kinds:
        pa = np.concatenate([pyr.store[i, :d.param_count].numpy() for i in range(cfg.m)])
        t0 = time.perf_counter()
        r = O.optimize([cd] * cfg.m, pa, s.numpy(), 0, s.shape[0], None, t.numpy(), k0=cfg.k0, iters=cfg.iters,
                       max_break_count=cfg.max_break_count, ratio=cfg.break_threshold_ratio, lr=cfg.lr, nthreads=threads)
        O.pyramid_fwd([cd] * cfg.m, cfg.k0, r["params_all"], src_c.numpy(), nthreads=threads)
        out["port-c"] = (time.perf_counter() - t0, int(r["steps"]))
    if "port-torch" in kinds:
        t0 = time.perf_counter()
        levels, _, _, steps = T.optimize(pyr.store[:, :d.param_count], s, t, m=cfg.m, k0=cfg.k0, iters=cfg.iters, lr=cfg.lr,
                                         max_break_count=cfg.max_break_count, ratio=cfg.break_threshold_ratio)
        T.pyramid_forward(levels, src_c, cfg.k0)
        out["port-torch"] = (time.perf_counter() - t0, int(steps))
    return out


def cpu_worker(idx, nproc, threads, kind, n_pairs):
    """`bench.py --cpu-worker idx nproc threads kind pairs`: one of the nproc concurrent processes of the whole-box CPU baseline --
    pinned to its own slice of `threads` cores, 1 warm-up pair, then n_pairs full pairs of its own (synthetic_pair(1000 + ...)); prints
    one JSON line with its per-pair seconds.  The processes start together and run the same amount of work, so their timed pairs overlap."""
    cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
    try:
        cpus = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, set(cpus[idx * threads:(idx + 1) * threads]) or set(cpus))
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(threads)
    from oracle import ndp_oracle as O
    O.lib()
    ts, its = [], []
    for q in range(n_pairs + 1):
        src, tgt, _, _ = synthetic_pair(1000 + idx * 16 + q)
        dt, steps = _cpu_pair(cfg, src, tgt, 1000 + idx * 16 + q, threads, kinds=(kind,))[kind]
        if q:
            ts.append(dt); its.append(steps)
    print(json.dumps({"idx": idx, "s_per_pair": ts, "adam_iters": its}))


def cpu_baseline_whole_box(kind, threads, physical, n_pairs=1):
    """SURVEY 8(d)'s optional figure: the WHOLE host -- one process per `threads`-core slice (physical // threads of them), each on its
    own pairs, all at once.  -> dict or None."""
    import subprocess
    nproc = max(1, physical // threads)
    if nproc < 2:
        return None
    quota = _cgroup_quota()
    if quota is not None and quota < 0.75 * nproc * threads:
        # (measured on the GPU boxes of round 4: quota 16 cores -- four 32-thread processes then share what one of them already could not use)
        return {"skipped": f"the container's CPU quota is {quota:g} cores: {nproc} x {threads} threads cannot run side by side",
                "cgroup_cpu_quota_cores": quota, "processes": nproc, "threads_per_process": threads}
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), str(nproc), str(threads), kind, str(n_pairs)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for i in range(nproc)]
    recs = []
    for pr in procs:
        out, _ = pr.communicate(timeout=900)
        for ln in out.splitlines():
            if ln.startswith("{"):
                recs.append(json.loads(ln))
    if len(recs) != nproc:
        return None
    rate = sum(len(r["s_per_pair"]) / sum(r["s_per_pair"]) for r in recs)
    return {"value": rate, "cgroup_cpu_quota_cores": quota, "unit": "pairs/s", "cores": nproc * threads, "processes": nproc, "threads_per_process": threads, "kind": kind,
            "sample": f"{nproc} concurrent processes x {threads} threads on disjoint core slices, each 1 warm-up + {n_pairs} full 8192-pt pairs of its own",
            "s_per_pair": [[round(x, 3) for x in r["s_per_pair"]] for r in sorted(recs, key=lambda r: r["idx"])]}


def cpu_baseline(cfg, pairs):
    """The reference's per-pair path on this box's host cores, two ways, both parity-pinned test infrastructure:
      port-c     oracle/ndp_oracle.c -- the bit-faithful scalar C restatement, OpenMP over the points;
      port-torch oracle/ndp_torch_ref.py -- the same loop on plain torch-CPU ops (BLAS-backed linear layers, autograd,
                 torch.optim.Adam): what the reference's own CPU path amounts to.
    Fixed policy (no calibration): threads = min(physical cores, 32, floor(the container's CPU quota)) -- OpenMP over 2000 points
    and 128-wide GEMMs stop scaling at 32, and threads beyond the cgroup quota are CFS-throttled onto the quota's worth of time
    (round 4 ran 32 threads under a 16-core quota: neither a 32-core nor a clean 16-core figure; that configuration is kept as the
    labelled secondary `oversubscribed`) -- the process restricted to that many cores, 1 warm-up pair, then 3 pairs (the bench's pairs 1..3,
    NDP.yaml unchanged, full register() work incl. the 8192-pt final warp); value = pairs / total seconds of the faster
    port, spread = (max - min) / median of its per-pair times.  `whole_box`: the faster port again as one process per
    32-core slice, all slices at once (what the whole host delivers on independent pairs)."""
    from oracle import ndp_oracle as O
    model_name, physical, logical = _cpu_info()
    quota = _cgroup_quota()
    threads_unlimited = max(1, min(physical, 32))
    threads = max(1, min(threads_unlimited, int(quota))) if quota else threads_unlimited
    aff = None
    try:
        aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(sorted(aff)[:threads]))
    except (AttributeError, OSError):
        aff = None
    old_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    O.lib()
    runs = {"port-c": [], "port-torch": []}
    iters = {"port-c": [], "port-torch": []}
    try:
        for k, (src, tgt) in enumerate(pairs[:4]):
            res = _cpu_pair(cfg, src, tgt, k, threads)
            if k:                                          # pair 0 is the warm-up
                for kind, (dt, steps) in res.items():
                    runs[kind].append(dt); iters[kind].append(steps)
        over = None
        if threads_unlimited > threads:                    # the round-4 configuration, for continuity: more threads than the quota feeds
            best0 = min(runs, key=lambda k: sum(runs[k]))
            if aff is not None:
                os.sched_setaffinity(0, set(sorted(aff)[:threads_unlimited]))
            torch.set_num_threads(threads_unlimited)
            ts = [_cpu_pair(cfg, src, tgt, k, threads_unlimited, kinds=(best0,))[best0][0] for k, (src, tgt) in list(enumerate(pairs[:3]))[1:]]
            over = {"pairs_per_s": len(ts) / sum(ts), "threads": threads_unlimited, "kind": best0, "s_per_pair": [round(x, 3) for x in ts],
                    "note": f"{threads_unlimited} threads under a {quota:g}-core cgroup quota: throttled, NOT a {threads_unlimited}-core figure"}
    finally:
        torch.set_num_threads(old_threads)
        if aff is not None:
            os.sched_setaffinity(0, aff)
    rep = {}
    for kind, ts in runs.items():
        med = sorted(ts)[len(ts) // 2]
        rep[kind] = {"pairs_per_s": len(ts) / sum(ts), "s_per_pair": [round(x, 3) for x in ts],
                     "ms_per_iter": 1e3 * sum(ts) / max(sum(iters[kind]), 1), "adam_iters": iters[kind],
                     "spread": (max(ts) - min(ts)) / med}
    best = max(rep, key=lambda k: rep[k]["pairs_per_s"])
    whole = None
    try:
        whole = cpu_baseline_whole_box(best, threads, physical)
    except Exception as exc:                               # the whole-box leg is a second opinion: never fail the line for it
        whole = {"error": repr(exc)}
    return {"value": rep[best]["pairs_per_s"], "unit": "pairs/s", "cores": threads, "kind": best,
            "sample": f"1 warm-up + {len(runs[best])} full 8192-pt pairs, NDP.yaml unchanged, register() work incl. the final warp; "
                      f"{threads} threads on {threads} of {physical} physical cores ({logical} logical"
                      + (f"; container CPU quota {quota:g} cores" if quota else "") + f"), {model_name}",
            "ms_per_iter": rep[best]["ms_per_iter"], "cpu_model": model_name, "physical_cores": physical, "cgroup_cpu_quota_cores": quota,
            "ports": rep, "oversubscribed": over, "whole_box": whole}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py <same args>`
    (one process per GPU, rendezvous on 127.0.0.1 at a free port) and exit with its status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // n))))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    if len(sys.argv) >= 7 and sys.argv[1] == "--cpu-worker":
        cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slots", type=int, default=256, help="pairs resident per engine (256: one level-kernel workgroup per pair and CU -- the per-workgroup "
                                                        "weight prologue over 32 tiles instead of 16; 128 until round 4)")
    ap.add_argument("--pairs-per-step", type=int, default=0, help="pairs registered per step per GPU (default 32 x slots)")
    ap.add_argument("--chunk", type=int, default=0, help="ticks between host polls (0: 4, or 8 for the landmark configuration E whose ticks are "
                                                          "four times shorter -- the per-chunk host work of three lanes must fit under one chunk of GPU work)")
    ap.add_argument("--engines", type=int, default=3, help="independent engines (HIP streams) per GPU, `slots` pairs each (3 x 128 until round 3, 2 x 256 in "
                                                             "round 4; 3 x 256 since round 5: +2.7 % over 2 x 256 in three alternating 4-step runs each, "
                                                             "profiles/r05_engines_sweep_long.txt)")
    ap.add_argument("--config", default="A", choices=list("ABCDE"),
                    help="SURVEY 8(d) workload: A NDP.yaml faithful (the headline line); B fixed work (= --fixed-work); C samples = 8192; "
                         "D Sim3/euler, 6000 samples of 24 856-pt clouds (shape transfer); E LNDP.yaml, 500 landmarks, m = 10")
    ap.add_argument("--fixed-work", action="store_true",
                    help="SURVEY 8(d) config B: early stop off, 50 iterations x 9 levels = 450 Adam steps per pair")
    ap.add_argument("--gemm-mode", type=int, default=-1, choices=list(range(-1, 8)),
                    help="-1 (default): the engine's default arithmetic (engine.DEFAULT_GEMM_MODE); 0: level kernels on the fp32 MFMA, "
                         "bitwise the oracle's fma chain; mask 1 forward | 2 bwd1 | 4 bwd2 (7 = all): their 128x128 contractions as "
                         "two-way fp16 splits on the fp16 MFMA (fp32-level accuracy, not bitwise)")
    ap.add_argument("--nn-mode", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="-1 (default): the engine chooses (one-pass kernel at throughput sizes -- on the matrix pipe when "
                         "engine.DEFAULT_NN_MATRIX -- latency shape for a few pairs); 0 one-pass on the vector pipe; 1 latency shape; "
                         "2 one-pass with the distances on the bf16 matrix pipe and exact re-evaluation (bit-identical results)")
    ap.add_argument("--drain-between-steps", action="store_true", help="one register_batch call per step (every step fills and drains the slots: "
                                                                        "rounds 1-4) instead of the steps as one stream of pairs")
    ap.add_argument("--no-alt", action="store_true", help="skip the second measurement in the other arithmetic configuration")
    ap.add_argument("--alt-steps", type=int, default=2, help="timed steps of the second measurement (1 warm-up step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 register() latency measurement")
    ap.add_argument("--allow-variant", action="store_true", help="run on an experiment build of the library (deformationpyramid_amd._native."
                                                                  "use_variant); the line records it as `library_variant`")
    args = ap.parse_args()
    if N.variant() and not args.allow_variant:
        sys.exit(f"bench.py: an experiment build of the library is selected ({N.variant()}); pass --allow-variant to time it")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NDP_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with ONE GPU (all ranks share cuda:0, the
    # aggregate goes over gloo on the CPU); the real multi-GPU run uses RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("NDP_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node has {torch.cuda.device_count()} "
                 "(NDP_BENCH_BACKEND=gloo rehearses the multi-rank path with every rank on cuda:0)")
    # NDP_BENCH_DIST=1 under a launcher with ONE rank still initialises the process group, so that the RCCL
    # all-reduce of the aggregate runs on a 1-GPU box too (tests/test_bench_gpu.py)
    use_dist = world > 1 or (os.environ.get("NDP_BENCH_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = max(world, 1)
    if use_dist:
        import torch.distributed as dist
        if dist.get_world_size() != max(args.gpus, 1):
            sys.exit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # one process per GPU: keep this rank (main thread, pair producer, torch's few host threads) on the CPUs of its GPU's NUMA
    # node, in a slice of its own (8 ranks x 2-3 busy threads otherwise migrate across sockets)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(n_gpus)))
    placement = pin_rank_to_gpu_numa(local_rank if backend != "gloo" else int(os.environ.get("LOCAL_RANK", "0")), local_world) if n_gpus > 1 else None
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // n_gpus)))   # host plumbing only; more threads hurt

    cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=local_rank)
    if args.fixed_work:
        args.config = "B"
    B = args.slots
    workload = ("synthetic 8192-pt pair, NDP.yaml (SE3/axis-angle, m=9, samples=2000, iters<=500, early stop on), "
                "full register(): init + level/Adam loop + 8192-pt final warp")
    make_pair = lambda i: synthetic_pair(i)
    if args.config == "B":
        cfg.iters, cfg.max_break_count = 50, 10 ** 9
        workload = ("synthetic 8192-pt pair, NDP.yaml with iters=50 and the early stop off (fixed work: 450 Adam iterations per "
                    "pair), full register(): init + level/Adam loop + 8192-pt final warp")
    elif args.config == "C":
        cfg.samples = 8192
        B = min(B, 32)
        workload = ("synthetic 8192-pt pair, NDP.yaml with samples=8192 (every point a Chamfer sample: S=8192, T~6144), "
                    "full register()")
    elif args.config == "D":
        cfg = Config(cfg, motion_type="Sim3", rotation_format="euler", samples=6000)
        B = min(B, 32)
        make_pair = lambda i: surface_pair(i, n_total=2 * 24856, partial=False)
        workload = ("shape-transfer shape: Sim3/euler, 6000 Chamfer samples of 24 856-pt surface clouds (the vertex count of "
                    "sim3_demo/AlienSoldier.ply), m=9, early stop on, full register() incl. the 24 856-pt final warp")
    elif args.config == "E":
        cfg = load_config(os.path.join(ROOT, "config", "LNDP.yaml"), device=local_rank)
        workload = ("LNDP.yaml (supervised path): 500 synthetic landmark correspondences per 8192-pt pair (noise 0.005), m=10, "
                    "w_cd=0: landmark MSE only, full register() incl. the 8192-pt final warp")
    if args.chunk <= 0:
        args.chunk = 8 if args.config == "E" else 4
    NP = args.pairs_per_step or (32 * B if args.config in "ABE" else 8 * B)
    # inputs resident in HBM before the timed region
    pairs, gts = [], []
    for i in range(NP):
        src, tgt, flow_gt, overlap = make_pair(rank * NP + i)
        if args.config == "E":
            ls, lt = synthetic_landmarks(rank * NP + i, src, flow_gt, k=500)
            pairs.append((src.to(dev), tgt.to(dev), (ls.to(dev), lt.to(dev))))
        else:
            pairs.append((src.to(dev), tgt.to(dev)))
        gts.append((flow_gt, overlap))

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    host_cpu = {}

    def timed_run(model, steps, warmup):
        """`warmup` untimed + exactly `steps` timed register_batch passes over the resident pairs, bracketed by barrier + synchronize."""
        torch.manual_seed(rank)
        for _ in range(warmup):
            model.register_batch(pairs, slots=B, chunk=args.chunk, engines=args.engines)
        barrier()
        t0 = time.perf_counter()
        c0 = time.process_time()                                 # CPU seconds of ALL threads of this rank (main, stepper, producers)
        if args.drain_between_steps:
            n_steps = n_evals = 0
            last = None
            for _ in range(steps):
                last = model.register_batch(pairs, slots=B, chunk=args.chunk, engines=args.engines)
                n_steps += sum(s.total_steps for s in model.last_states)
                n_evals += sum(s.total_evals for s in model.last_states)
        else:
            # the K steps as ONE stream of K x pairs_per_step pairs: a slot that frees up at the end of step k takes the first pair of
            # step k + 1 (a service fed by a queue never drains between batches; round 4 filled and drained 512 slots per step).  The
            # step stays the unit of work and of reporting: exactly `steps` passes over the resident pairs, each pair's full
            # register() -- the results of the last pass are kept for the accuracy block, the others handed to a sink and dropped
            last = [None] * len(pairs)
            first_of_last = (steps - 1) * len(pairs)

            def sink(i, warped, state):
                if i >= first_of_last:
                    last[i - first_of_last] = (warped, None)
            model.register_batch(pairs * steps, slots=B, chunk=args.chunk, engines=args.engines, sink=sink)
            n_steps = sum(s.total_steps for s in model.last_states)
            n_evals = sum(s.total_evals for s in model.last_states)
        barrier()
        host_cpu["s"] = time.process_time() - c0
        return time.perf_counter() - t0, n_steps, n_evals, last

    def accuracy_sums(last):
        keys = msum = None
        for (warped, _), item, (flow_gt, overlap) in zip(last, pairs, gts):
            mtr = compute_flow_metrics(warped - item[0], flow_gt.to(dev), overlap.to(dev))
            keys = list(mtr.keys())
            v = np.array([mtr[k] for k in keys], dtype=np.float64)
            msum = v if msum is None else msum + v
        return keys, msum

    model = Registration(cfg, gemm_mode=None if args.gemm_mode < 0 else args.gemm_mode, nn_mode=None if args.nn_mode < 0 else args.nn_mode)
    elapsed, steps_total, evals_total, last = timed_run(model, args.steps, args.warmup)
    host_cpu_main = host_cpu["s"]
    eng0 = model._engines[0]
    main_modes = (eng0.gemm_mode, eng0.nn_mode)
    keys, msum = accuracy_sums(last)                          # accuracy of the last step's pairs (not timed)

    # the single collective (RCCL): one SUM all-reduce carries the sums, every rank's pair count and every rank's elapsed time
    job = job_summary(args.steps * NP, elapsed, [float(steps_total), float(evals_total)] + list(msum) + [float(NP)],
                      torch.device("cpu") if backend == "gloo" else dev)
    n_pairs, elapsed = job["pairs"], job["elapsed"]
    sums = job["sums"].numpy()
    n_steps, n_evals = sums[0], sums[1]
    metrics = {k: float(v / sums[-1]) for k, v in zip(keys, sums[2:-1])}

    out = {
        "metric": "point-cloud pairs/sec (8192-pt NDP registration)",
        "value": job["value"],
        "unit": "pairs/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_TEXT.get(main_modes[0], "f32"),
        "data": "synthetic",
        "config": {"workload": workload, "survey_8d_config": args.config,
                   "contraction_arithmetic": ARITH_TEXT.get(main_modes[0], f"mask {main_modes[0]} (1 fwd | 2 bwd1 | 4 bwd2) on fp16 splits, the rest on the fp32 MFMA"),
                   "nn_kernel": NN_TEXT[main_modes[1]], "gemm_mode": main_modes[0], "nn_mode": main_modes[1],
                   "pairs_per_step_per_gpu": NP, "steps_as_one_stream": not args.drain_between_steps, "resident_slots_per_gpu": B * args.engines, "engines_per_gpu": args.engines, "parallelism": f"pair-parallel x{n_gpus}, no data-path collective",
                   "backend": ("none" if not use_dist else ("rccl" if backend == "nccl" else backend)),
                   "seeds": "rank r registers synthetic_pair(r*pairs_per_step + i), i < pairs_per_step; torch.manual_seed(r) "
                            "feeds the pyramid init and the sampling permutations"},
        "ms_per_iter": 1e3 * elapsed * n_gpus / max(n_steps, 1.0),
        "adam_iters_per_pair": n_steps / n_pairs,
        "loss_evals_per_pair": n_evals / n_pairs,
        "accuracy": metrics,
        # what one rank costs the HOST (rank 0's process CPU time over its timed region: main thread, engine steppers, pair
        # producers): the 8-GPU readiness figure -- ranks x host_cores_busy must fit the node's cores / the container's quota
        "host_cpu_s_per_pair": host_cpu_main / (args.steps * NP),
        "host_cores_busy": host_cpu_main / max(job["elapsed_per_rank"][0] if job["elapsed_per_rank"] else elapsed, 1e-9),
        "cgroup_cpu_quota_cores": _cgroup_quota(),
        "ndp_build_id": N.lib().ndp_build_id().decode(), "library_variant": N.variant(),
        "world_size": job["world_size"],
        "ranks": {"pairs_per_s_min": job["rank_pairs_per_s_min"], "pairs_per_s_max": job["rank_pairs_per_s_max"],
                  "elapsed_s": [round(t, 4) for t in job["elapsed_per_rank"]], "allreduce_ms": job["allreduce_ms"],
                  "cpu_placement_rank0": placement},
    }

    single = rank == 0 and n_gpus == 1
    if single and not args.no_roofline:
        out.update(roofline_report(model, pairs, B, args.config))
    if single and not args.no_alt and args.config == "A" and main_modes[0] in (0, 7):
        # the same workload in the OTHER arithmetic configuration, on a shorter step count: both numbers in one line
        alt_kw = dict(gemm_mode=7, nn_matrix=True) if main_modes[0] == 0 else dict(gemm_mode=0, nn_matrix=False)
        alt_model = Registration(cfg, **alt_kw)
        a_elapsed, a_steps, a_evals, a_last = timed_run(alt_model, args.alt_steps, 1)
        a_eng = alt_model._engines[0]
        a_keys, a_msum = accuracy_sums(a_last)
        alt = {"name": "split (fp16x2 level kernels + matrix-pipe NN)" if main_modes[0] == 0 else "bitwise (fp32-MFMA level kernels + vector-pipe NN)",
               "value": args.alt_steps * NP / a_elapsed, "unit": "pairs/s", "steps": args.alt_steps, "warmup": 1,
               "ms_per_step": 1e3 * a_elapsed / args.alt_steps, "ms_per_iter": 1e3 * a_elapsed / max(a_steps, 1),
               "adam_iters_per_pair": a_steps / (args.alt_steps * NP), "dtype": DTYPE_TEXT.get(a_eng.gemm_mode, "f32"),
               "contraction_arithmetic": ARITH_TEXT[a_eng.gemm_mode], "nn_kernel": NN_TEXT[a_eng.nn_mode],
               "gemm_mode": a_eng.gemm_mode, "nn_mode": a_eng.nn_mode,
               "accuracy": {k: float(v / NP) for k, v in zip(a_keys, a_msum)}}
        if not args.no_roofline:
            alt.update(roofline_report(alt_model, pairs, B, args.config))
        out["alt"] = alt
        if main_modes[0] == 0:
            out["optin"] = alt                                  # (the name VERDICT r02 asked for while the split path was opt-in)
        del alt_model
    if single and not args.no_roofline and args.config == "A":
        # accuracy on pairs NDP actually solves: 8 partial-overlap SURFACE pairs under EIGHT process seeds (as many as the reference's fixture) (the reference seeds once and
        # registers pair after pair, eval_nolearned.py:22; its own seed-to-seed distribution is tests/golden/F10c: 8 seeds x 8 pairs,
        # full-EPE 6.42 +- 0.11 (s.e. of the 8-seed mean), seed means 6.06 .. 6.93; zero flow: EPE 13.4, AccS 0.8 %) -- not timed
        sp = [surface_pair(p) for p in range(8)]
        dp = [(a.to(dev), b.to(dev)) for a, b, _, _ in sp]
        per_seed = []
        for seed in range(8):
            torch.manual_seed(seed)
            res = model.register_batch(dp, slots=8, engines=1)
            acc = None
            for (w, _), (a, _, fg, ov) in zip(res, sp):
                mtr = compute_flow_metrics(w - a.to(dev), fg.to(dev), ov.to(dev))
                v = np.array(list(mtr.values()), dtype=np.float64)
                acc = v if acc is None else acc + v
            per_seed.append(acc / 8)
        per_seed = np.array(per_seed)
        mk = list(mtr.keys())
        out["accuracy_surface_pairs"] = dict({k: float(per_seed[:, j].mean()) for j, k in enumerate(mk)},
                                             seeds=8, pairs=8,
                                             seed_min={k: float(per_seed[:, mk.index(k)].min()) for k in ("full-epe", "full-AccS", "full-AccR")},
                                             seed_max={k: float(per_seed[:, mk.index(k)].max()) for k in ("full-epe", "full-AccS", "full-AccR")},
                                             reference={"full-epe": 6.42, "full-AccS": 33.7, "full-AccR": 62.4, "seeds": 8,
                                                        "seed_mean_se": {"full-epe": 0.11, "full-AccS": 1.3, "full-AccR": 1.2},
                                                        "seed_min": {"full-epe": 6.06, "full-AccS": 27.0, "full-AccR": 57.7},
                                                        "seed_max": {"full-epe": 6.93, "full-AccS": 37.8, "full-AccR": 66.6},
                                                        "source": "tests/golden/F10c_surface_benchmark_seeds.npz (the reference run in the build container)"},
                                             zero_flow={"full-epe": 13.36, "full-AccS": 0.83})
    if single and not args.no_latency and args.config == "A":
        out["latency"] = latency_profile(cfg, pairs, gemm_mode=main_modes[0])
        if main_modes[0] in (0, 7):                            # and the other arithmetic at batch 1, fewer repeats
            out["latency"]["alt"] = latency_profile(cfg, pairs, repeats=3, gemm_mode=7 - main_modes[0])
    if single and not args.no_cpu_baseline and args.config == "A":
        out["cpu_baseline"] = cpu_baseline(cfg, pairs)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
