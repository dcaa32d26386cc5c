"""Per-phase shader-cycle breakdown of the level kernels (experiment build, -DNDP_PHASE_TIMING).

    python tools/phase_timing.py [B] [ticks] [extra hipcc flags...]

Builds a timing variant of the library next to the product one (never shipped), runs B pairs for `ticks`
ticks at level 0 and prints, per phase, the cycles thread 0 of a workgroup spends per tile (mean over
workgroups and tiles).  s_memtime ticks are shader cycles."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
extra = sys.argv[3:]
lib = os.path.join(ROOT, "gpurun_out", "libndp_phase.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
src = os.path.join(ROOT, "deformationpyramid_amd", "csrc", "ndp_kernels.hip")
if os.environ.get("NDP_PT_LIB"):          # a timing build made elsewhere (e.g. of another revision of the sources): use it as it is
    lib = os.path.abspath(os.environ["NDP_PT_LIB"])
else:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                       "-DNDP_PHASE_TIMING"] + extra + ["-o", lib, src])
os.environ["NDP_HIP_LIB"] = lib
import torch
from deformationpyramid_amd import _native as N
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair

dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _modes import from_env
model = Registration(cfg, **from_env())
if os.environ.get("NDP_PT_G"):            # level-kernel workgroups per pair forced (e.g. B = 128 with G = 1: half of the CUs idle)
    import deformationpyramid_amd.registration as _R
    _BE, _G = _R.BatchedEngine, int(os.environ["NDP_PT_G"])
    _R.BatchedEngine = lambda *a, **k: _BE(*a, G=_G, **k)
preps = [model._prepare(*[t.to(dev) for t in synthetic_pair(i)[:2]], None) for i in range(B)]
eng = model._engine(B, preps[0])
eng.load_jobs([p.load_job(b) for b, p in enumerate(preps)][:16])
for i in range(16, B, 16):
    eng.load_jobs([p.load_job(b) for b, p in enumerate(preps)][i:i + 16])
# timing-only variants with wrong results: ONE stage alone, over and over, the pair states untouched (NDP_PT_STAGE = 0 forward .. 4 bwd1)
STAGE = int(os.environ.get("NDP_PT_STAGE", "0" if os.environ.get("NDP_PT_FWD_ONLY") == "1" else "-1"))
FWD_ONLY = STAGE >= 0
if not FWD_ONLY:
    eng.run_ticks(4)
torch.cuda.synchronize()
L = N.lib()
buf = (ctypes.c_ulonglong * 96)()
L.ndp_debug_phase_read(buf, 1)
if FWD_ONLY:
    eng.run_stages(0, STAGE)                                 # the stages before it once, with whatever this build computes
    torch.cuda.synchronize()
    L.ndp_debug_phase_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ticks):
        eng.run_stages(STAGE, STAGE)
    e1.record()
    torch.cuda.synchronize()
    ms = [0] * 6
    ms[STAGE] = e0.elapsed_time(e1)
else:
    ms = eng.run_ticks_timed(ticks)
L.ndp_debug_phase_read(buf, 1)
tiles = B * (eng.n_cap // 64) * ticks
if False:
    pass
names = {0: "bwd2 load+lds", 1: "bwd2 barrier", 2: "bwd2 mfma(outer+gemm)+db", 3: "bwd2 barrier", 4: "bwd2 epilogue",
         5: "bwd2 barrier", 6: "bwd2 store", 7: "bwd2 barrier",
         24: "bwd1 load+lds", 25: "bwd1 barrier", 26: "bwd1 mfma(outer+gemm)", 27: "bwd1 barrier", 28: "bwd1 dz0 epilogue+db0",
         29: "bwd1 barrier", 30: "bwd1 dW0 (mfma16)", 31: "bwd1 barrier",
         12: "fwd weights -> registers (once per workgroup)", 15: "fwd encodings of a chunk of tiles", 18: "fwd a chunk's tiles (outer stamp)",
         14: "fwd layer0", 16: "fwd layer1+h0 store", 17: "fwd barrier",
         19: "fwd barrier", 23: "fwd h2 store", 20: "fwd heads (mfma16)", 21: "fwd barrier"}
print("per-tick ms [fwd nn loss bwd2 bwd1 upd]:", [round(x / ticks, 4) for x in ms])
if buf[95]:            # workgroup (0, 0) of the instrumented kernels: shader cycles over 100 MHz ticks (one stage alone: that kernel's clock)
    print(f"shader clock while the instrumented kernels ran: {buf[94] / buf[95] * 0.1:.2f} GHz  ({buf[94]} shader cycles / {buf[95]} ticks of 10 ns)")
for grp, lo in (("bwd2", 0), ("fwd", 12), ("bwd1", 24)):
    tot = sum(buf[i] for i in range(lo, lo + 12))
    print(f"{grp}: {tot / tiles:.0f} cycles per tile (thread 0 wall)")
    for i in range(lo, lo + 12):
        if buf[i]:
            print(f"   {str(names.get(i, i)):28s} {buf[i] / tiles:9.0f}  {100.0 * buf[i] / tot:5.1f} %")

lnames = {36: "loss-grad state+row fold+own term", 37: "loss-grad count pass", 38: "loss-grad scan", 39: "loss-grad fill pass",
          40: "loss-grad sort+accumulate", 41: "loss-grad head bwd + dO",
          48: "loss-dec state", 49: "loss-dec sx fold", 50: "loss-dec sx block sum", 51: "loss-dec sy", 52: "loss-dec decision"}
wg_grad = B * ((eng.n_cap + 255) // 256) * ticks
wg_dec = B * ticks
for lo, n, tag in ((36, wg_grad, "loss gradient workgroup"), (48, wg_dec, "loss decision workgroup")):
    tot = sum(buf[i] for i in range(lo, lo + 12))
    print(f"{tag}: {tot / n:.0f} cycles (thread 0 wall)")
    for i in range(lo, lo + 12):
        if buf[i]:
            print(f"   {lnames.get(i, i):36s} {buf[i] / n:9.0f}  {100.0 * buf[i] / tot:5.1f} %")

if eng.nn_mode == 2:
    nmn = ["setup: targets + B operands, sources -> LDS, |x|max", "A operands of a 128-source block (x4 per wave)", "distances: 8 tiles x (8 MFMA + row / column minima)",
           "rows: transposition, best class, exact evaluation", "barrier (waves done with their blocks)", "columns: fold table, exact re-scan of the winning block"]
    mx8 = not (eng.gemm_mode & 128)                          # the 8-wave shape: 512 targets per workgroup
    wgs = B * ((eng.t_cap + (511 if mx8 else 255)) // (512 if mx8 else 256)) * ticks
    tot = sum(buf[64 + i] for i in range(6))
    print(f"nn_mx{'8' if mx8 else ''}: {tot / wgs:.0f} cycles per workgroup (all sources x {512 if mx8 else 256} targets; thread 0 wall)")
    for i, nm in enumerate(nmn):
        print(f"   {nm:58s} {buf[64 + i] / wgs:9.0f}  {100.0 * buf[64 + i] / max(tot, 1):5.1f} %")
FUSED = (eng.gemm_mode & 7) == 7 and not (eng.gemm_mode & 16)
if FUSED:
    # default: three stamps per tile, one after each barrier (any stamp INSIDE a barrier interval pins the schedule of this
    # register-capped kernel and its timing build spills); -DNDP_PHASE_TIMING_FINE adds the inner ones (distorted: read with care)
    nmf = ["stage 4 (wgrad1, swapped dgrad1, [dW0|db0], next tile's requests) + wait for the h2 rows + top barrier", "(fine) stage 1",
           "-", "(fine) stage 2", "stages 1-2 (small rows; dWh, dz2 chain + split) + wait for the h1 image + barrier", "(fine) h0 recompute, wgrad2",
           "(fine) dgrad2 + mask + dz1 split", "stage 3 (h0 recompute + split, wgrad2, dgrad2 + mask + dz1 split) + barrier", "(fine) wgrad1",
           "(fine) dgrad1", "prologue: weight slices, staging (once per workgroup)", "tail: partial stores, bias sums (once per workgroup)"]
    tot = sum(buf[24 + i] for i in range(12))
    print(f"bwd_f (fused): {tot / tiles:.0f} cycles per tile (thread 0 wall)")
    for i, nm in enumerate(nmf):
        print(f"   {nm:66s} {buf[24 + i] / tiles:9.0f}  {100.0 * buf[24 + i] / max(tot, 1):5.1f} %")
if eng.gemm_mode & 4 and not FUSED:
    nm2 = ["top barrier (incl. wait for the requested rows)", "h1 split + h2 tile + dO rows -> LDS", "barrier", "dz2 chain (VALU) + split -> planes", "barrier",
           "dWh (fp32 MFMA 16x16x4)", "wgrad (24 MFMA 32x32x16)", "dgrad (48 MFMA 16x16x32) + mask + dz1 store", "tail: dW store, bias sums (once)"]
    tot = sum(buf[i] for i in range(12))
    print(f"bwd2_8: {tot / tiles:.0f} cycles per tile (thread 0 wall)")
    for i, nm in enumerate(nm2):
        print(f"   {nm:52s} {buf[i] / tiles:9.0f}  {100.0 * buf[i] / max(tot, 1):5.1f} %")
if eng.gemm_mode & 2 and not FUSED:
    n_arr = max(buf[24 + 8], 1)
    print("bwd1_8: mean lateness of wave w at the top-of-tile barrier relative to wave 0 (cycles):", [round(buf[24 + w] / n_arr) for w in range(8)])
if False:
    nm1 = ["top barrier (incl. wait for the requested rows)", "dz1 / h0 split -> planes", "barrier", "wgrad (48 MFMA 32x32x16)", "dgrad (96 MFMA 16x16x32) + mask + dW0 fma",
           "tail: dW store, bias sums, dW0 fold (once)"]
    tot = sum(buf[24 + i] for i in range(12))
    print(f"bwd1_8: {tot / tiles:.0f} cycles per tile (thread 0 wall)")
    for i, nm in enumerate(nm1):
        print(f"   {nm:52s} {buf[24 + i] / tiles:9.0f}  {100.0 * buf[24 + i] / max(tot, 1):5.1f} %")
if getattr(eng, 'fwd_as', False):
    f9 = ["stage W1 / W0 (once per workgroup)", "x, sincos, layer-0 operands", "layer 0 (12 MFMA) + h0 store + split", "layer 1 (192 MFMA) + h1 store",
          "phase barrier", "stage W2 / Wh (once per workgroup)", "h1 read back + split", "layer 2 + heads (240 MFMA) + h2 store", "-", "-", "-"]
    tot = sum(buf[12 + i] for i in range(11))
    print(f"fwd_as (activation-stationary): {tot / tiles:.0f} cycles per 64 points (wave 0 wall; a wave's 32-point group = 1/4 of the workgroup's work per 64 points... see below)")
    for i, nm in enumerate(f9):
        print(f"   {nm:48s} {buf[12 + i] / tiles:9.0f}  {100.0 * buf[12 + i] / max(tot, 1):5.1f} %")
elif eng.gemm_mode & 1:
    f8 = ["-", "layer 0 (two MFMAs per group) + split + planes", "barrier", "-", "layer 1 MFMA + epilogue", "barrier", "-",
          "layer 2 MFMA + epilogue", "barrier", "heads (waves 0..3)", "weights -> registers (once per workgroup; 4-wave experiment only)"]
    if "-DNDP_EXPERIMENT_FWD_4W" in extra:
        f8 = ["-", "P0: layer 0, epilogue (L0, g0) [+ heads of the previous tile's g1]", "barriers (six per tile)", "P1: MFMAs (L1, g0) | epilogue (L0, g1)",
              "P2: MFMAs (L1, g1) | epilogue (L1, g0)", "P3: MFMAs (L2, g0) | epilogue (L1, g1)", "P4: MFMAs (L2, g1) | epilogue (L2, g0)",
              "P5: epilogue (L2, g1), heads of g0", "-", "-", "weights -> registers (once per workgroup)"]
    if "-DNDP_EXPERIMENT_FWD_LP" in extra:
        f8 = ["-", "H1: L1 waves MFMAs (L1, s) | L2 waves heads (s-3)", "barrier", "H2, L1 waves: epilogue (L1, s)",
              "H2: L1 waves encode s+2 | L2 waves MFMAs (L2, s-1)", "barrier", "H2, L1 waves: layer 0 of s+1", "H1, L2 waves: epilogue (L2, s-2)", "-", "-",
              "weights -> registers, first encodings (once per workgroup)"]
    tot = sum(buf[12 + i] for i in range(11))
    print(f"fwd8 (fp16 splits): {tot / tiles:.0f} cycles per tile (thread 0 wall)")
    for i, nm in enumerate(f8):
        print(f"   {nm:66s} {buf[12 + i] / tiles:9.0f}  {100.0 * buf[12 + i] / max(tot, 1):5.1f} %")
