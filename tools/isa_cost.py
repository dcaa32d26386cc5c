#!/usr/bin/env python3
"""Static issue-cost estimate of a kernel's barrier intervals from its ISA (-save-temps .s of ndp_kernels.hip).

usage: tools/isa_cost.py <kernel> [extra hipcc flags...]

On gfx950 the vector ALU and the matrix pipe of a SIMD do NOT run side by side -- neither from two waves nor inside one
(tools/experiments/micro/coexec.hip: a wave of MFMAs and a wave of FMAs on one SIMD take the SUM of their solo times) -- so a
kernel's floor per SIMD is  sum(VALU issue time) + sum(MFMA time)  over its waves.  The weights below are the measured issue
costs per instruction and SIMD (tools/experiments/micro/valu_rates.hip, two waves per SIMD, ns at the clock the part ran at):
  1.2  v_mul/add/sub_f32, v_mov, v_and/or, v_add_u32, v_fmamk          2.0  conversions, min/max/med3, cmp, cndmask, shifts,
  2.55 v_fma_f32 (three distinct sources)   1.85 v_fmac_f32                 lshl_add, perm, bfe, packed fp16 / fp32 mul-add
  7.6  v_mfma 16x16x32 f16 (16 cycles)      15.2 v_mfma 32x32x16 f16
Loops are counted once (the dz2 chain's head-row loop runs nh times), so read the numbers as a lower bound per tile."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kern = sys.argv[1]
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-save-temps",
                       '-DNDP_BUILD_ID="x"'] + sys.argv[2:] + ["-o", os.path.join(tmp, "lib.so"), os.path.join(ROOT, "deformationpyramid_amd/csrc/ndp_kernels.hip")],
                      cwd=tmp, stderr=subprocess.DEVNULL)
text = open(os.path.join(tmp, "ndp_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
a = next(i for i, l in enumerate(text) if l.startswith(kern + ":"))
b = next(i for i in range(a, len(text)) if "s_endpgm" in text[i])
body = text[a:b]
FAST = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_fmamk_f32",
        "v_fmaak_f32", "v_mov_b64", "v_accvgpr")
def cost(op):
    if "mfma" in op:
        return 15.2 if "32x32" in op else 7.6, "mfma"
    if not op.startswith("v_"):
        return 0.0, "ds" if op.startswith("ds_") else ("vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "other")
    if op.startswith("v_fma_f32"):
        return 2.55, "valu"
    if op.startswith("v_fmac_f32"):
        return 1.85, "valu"
    if op.startswith("v_pk_fma_f32"):
        return 2.27, "valu"
    if any(op.startswith(f) for f in FAST):
        return 1.2, "valu"
    return 2.0, "valu"
bars = [i for i, l in enumerate(body) if "s_barrier" in l]
edges = [0] + bars + [len(body)]
print(f"{kern}: {len(body)} lines, {len(bars)} barriers")
print(f"{'interval':>14s} {'VALU n':>7s} {'VALU ns':>8s} {'MFMA n':>7s} {'MFMA ns':>8s} {'LDS n':>6s} {'VMEM n':>7s}   heaviest vector instructions")
for lo, hi in zip(edges[:-1], edges[1:]):
    n = collections.Counter(); ns = collections.Counter(); ops = collections.Counter()
    for l in body[lo:hi]:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        op = l.split()[0]
        c, k = cost(op)
        n[k] += 1; ns[k] += c
        if k == "valu":
            ops[re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", op)] += c
    if n["valu"] + n["mfma"] + n["ds"] < 8:
        continue
    top = ", ".join(f"{k} {v:.0f}" for k, v in ops.most_common(6))
    print(f"{lo:6d}-{hi:6d} {n['valu']:7d} {ns['valu']:8.0f} {n['mfma']:7d} {ns['mfma']:8.0f} {n['ds']:6d} {n['vmem']:7d}   {top}")
