#!/usr/bin/env python3
"""Generator of coexec2.hip: what a vector / LDS instruction costs when it is hand-placed BETWEEN independent MFMAs.

VERDICT r04 item 1: the regime MI355X_MICROARCH.md measured -- ONE wave per SIMD (256-thread workgroup, one per CU),
`v_mfma_f32_32x32x16_f16` rotating over four independent accumulators, k fillers between consecutive MFMAs -- and the
same with `16x16x32`, with two waves per SIMD, with `s_setprio 1` on the younger half, and with the vector work in the
partner wave instead of in the gaps.  Every stream is ONE inline-asm block per loop body, so the order is the order
written here.  Output: shader cycles per MFMA (s_memtime around the loop, mean over the waves that ran MFMAs).

    python tools/experiments/micro/coexec2_gen.py > tools/experiments/micro/coexec2.hip
    hipcc --offload-arch=gfx950 -O3 -o coexec2 tools/experiments/micro/coexec2.hip && ./coexec2
"""
import sys

NACC = 4          # independent accumulators
UNROLL = 8        # MFMAs per loop body
NF = 8            # rotating filler destinations (inline asm takes at most 30 operands)
NS = 4            # filler sources (never written)

# operand numbering of the asm block
#  %0..%3 accumulators, %4 a, %5 b, %6..%13 filler destinations, %14..%17 sources, %18 lds address, %19..%22 lds destinations
def acc(i): return f"%{i}"
A, B = "%4", "%5"
def fd(i): return f"%{6 + i % NF}"
def fs(i): return f"%{6 + NF + i % NS}"
LDSA = f"%{6 + NF + NS}"
def ld(i): return f"%{7 + NF + NS + i % 4}"


def filler(kind, n):
    """n-th filler instruction of the stream (n counts over the whole body)"""
    if kind == "med3":
        return f"v_med3_f32 {fd(n)}, {fs(n)}, {fs(n + 1)}, {fs(n + 2)}"
    if kind == "cvt":
        return f"v_cvt_pk_f16_f32 {fd(n)}, {fs(n)}, {fs(n + 1)}"
    if kind == "sub":
        return f"v_sub_f32 {fd(n)}, {fs(n)}, {fs(n + 1)}"
    if kind == "fma":
        return f"v_fma_f32 {fd(n)}, {fs(n)}, {fs(n + 1)}, {fs(n + 2)}"
    if kind == "mix":   # the split's own sequence: clamp, pack-convert, convert back, subtract, pack-convert
        return [f"v_med3_f32 {fd(n)}, {fs(n)}, {fs(n + 1)}, {fs(n + 2)}",
                f"v_cvt_pk_f16_f32 {fd(n)}, {fs(n)}, {fs(n + 1)}",
                f"v_cvt_f32_f16 {fd(n)}, {fs(n)}",
                f"v_sub_f32 {fd(n)}, {fs(n)}, {fs(n + 1)}",
                f"v_cvt_pk_f16_f32 {fd(n)}, {fs(n + 2)}, {fs(n + 3)}"][n % 5]
    if kind == "fmix":  # the split by v_fma_mix: hi half, hi half, lo half (reads the hi pair), lo half
        return [f"v_fma_mixlo_f16 {fd(n)}, {fs(n)}, {fs(n + 1)}, 0",
                f"v_fma_mixhi_f16 {fd(n)}, {fs(n + 2)}, {fs(n + 1)}, 0",
                f"v_fma_mixlo_f16 {fd(n)}, {fs(n)}, {fs(n + 1)}, -{fs(n + 3)} op_sel_hi:[0,0,1]",
                f"v_fma_mixhi_f16 {fd(n)}, {fs(n + 2)}, {fs(n + 1)}, -{fs(n + 3)} op_sel:[0,0,1] op_sel_hi:[0,0,1]"][n % 4]
    if kind == "lds":
        return f"ds_read_b128 {ld(n)}, {LDSA} offset:{(n % 8) * 1024}"
    if kind == "mixl":  # four of the split's instructions, then one LDS read
        if n % 5 == 4:
            return f"ds_read_b128 {ld(n)}, {LDSA} offset:{(n % 8) * 1024}"
        return filler("mix", n)
    raise ValueError(kind)


def body(shape, kind, k, mfma=True):
    op = {"32": "v_mfma_f32_32x32x16_f16", "16": "v_mfma_f32_16x16x32_f16"}[shape]
    lines, n = [], 0
    for u in range(UNROLL):
        if mfma:
            a = acc(u % NACC)
            lines.append(f"{op} {a}, {A}, {B}, {a}")
        for _ in range(k):
            lines.append(filler(kind, n))
            n += 1
    if kind in ("lds", "mixl") and k:
        lines.append("s_waitcnt lgkmcnt(0)")
    return "\\n\\t".join(lines)


OPERANDS = (': "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a), "+v"(b), '
            + ", ".join(f'"+v"(f[{i}])' for i in range(NF)) + ", "
            + ", ".join(f'"+v"(s[{i}])' for i in range(NS)) + ", "
            + '"+v"(la), "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3)')

# wave configurations: threads, setprio on waves 4-7, cross (waves 0-3 MFMA only, waves 4-7 fillers only until told to stop)
CONFIGS = {"w1": (256, False, False), "w2": (512, False, False), "w2p": (512, True, False), "w2x": (512, False, True),
           "w2xp": (512, True, True)}


def kernel(name, cfg, shape, kind, k):
    threads, prio, cross = CONFIGS[cfg]
    acct = "f32x16" if shape == "32" else "f32x4"
    out = [f'extern "C" __global__ void __launch_bounds__({threads}) {name}(unsigned long long *cyc, float *out, int nit) {{',
           "    extern __shared__ float lds[];",
           "    PROLOGUE(%s)" % acct]
    if prio:
        out.append('    if (threadIdx.x >= 256) asm volatile("s_setprio 1");')
    if not cross:
        out += ["    __syncthreads();",
                "    unsigned long long t0 = __builtin_amdgcn_s_memtime();",
                "    for (int it = 0; it < nit; ++it)",
                f'        asm volatile("{body(shape, kind, k)}" {OPERANDS});',
                "    unsigned long long t1 = __builtin_amdgcn_s_memtime();",
                "    cyc[2 * (blockIdx.x * 8 + (threadIdx.x >> 6))] = t1 - t0;",
                "    cyc[2 * (blockIdx.x * 8 + (threadIdx.x >> 6)) + 1] = (unsigned long long)nit * %d;" % UNROLL]
    else:
        out += ["    volatile int *flag = (volatile int *)lds;",
                "    if (threadIdx.x == 0) *flag = 0;",
                "    __syncthreads();",
                "    unsigned long long t0 = __builtin_amdgcn_s_memtime(), cnt = 0;",
                "    if (threadIdx.x < 256) {",
                "        for (int it = 0; it < nit; ++it)",
                f'            asm volatile("{body(shape, kind, 0)}" {OPERANDS});',
                "        cnt = (unsigned long long)nit * %d;" % UNROLL,
                "        __builtin_amdgcn_s_waitcnt(0);",
                "        if ((threadIdx.x & 63) == 0) atomicAdd((int *)lds, 1);",
                "    } else {",
                "        while (*flag < 4) {",
                f'            asm volatile("{body(shape, kind, k, mfma=False)}" {OPERANDS});',
                "            cnt += %d;" % (UNROLL * k),
                "        }",
                "    }",
                "    unsigned long long t1 = __builtin_amdgcn_s_memtime();",
                "    cyc[2 * (blockIdx.x * 8 + (threadIdx.x >> 6))] = t1 - t0;",
                "    cyc[2 * (blockIdx.x * 8 + (threadIdx.x >> 6)) + 1] = cnt;"]
    out += ["    EPILOGUE", "}"]
    return "\n".join(out)


HEADER = r'''// GENERATED by coexec2_gen.py -- do not edit.  What a vector / LDS instruction costs when it is hand-placed BETWEEN independent MFMAs
// on gfx950: one wave per SIMD (w1), two waves per SIMD with the same stream (w2), the younger half at s_setprio 1 (w2p), and the
// fillers in the partner wave instead of in the gaps (w2x / w2xp).
//   hipcc --offload-arch=gfx950 -O3 -o coexec2 coexec2.hip && ./coexec2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Var { const char *cfg, *shape, *kind; int k, threads, cross; void (*fn)(unsigned long long *, float *, int); };
#define PROLOGUE(ACCT) \
    ACCT c0, c1, c2, c3; \
    for (int e = 0; e < (int)(sizeof(ACCT) / 4); ++e) { c0[e] = 0.f; c1[e] = 1.f; c2[e] = 2.f; c3[e] = 3.f; } \
    h16x8 a, b; \
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); } \
    float f[8], s[4]; \
    for (int e = 0; e < 8; ++e) f[e] = 0.f; \
    for (int e = 0; e < 4; ++e) s[e] = 0.25f * e + 1e-3f * threadIdx.x; \
    unsigned la = (threadIdx.x & 63) * 16 + 64; \
    f32x4 l0 = {0.f, 0.f, 0.f, 0.f}, l1 = l0, l2 = l0, l3 = l0; \
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) lds[e] = 1.f;
#define EPILOGUE \
    { float r = 0.f; \
      for (int e = 0; e < (int)(sizeof(c0) / 4); ++e) r += c0[e] + c1[e] + c2[e] + c3[e]; \
      for (int e = 0; e < 8; ++e) r += f[e]; \
      for (int e = 0; e < 4; ++e) r += l0[e] + l1[e] + l2[e] + l3[e]; \
      out[blockIdx.x * blockDim.x + threadIdx.x] = r; }
'''

MAIN = r'''
int main(int argc, char **argv) {
    const char *only = argc > 1 ? argv[1] : nullptr;
    const int NB = 256, nit = 1500;
    unsigned long long *cyc; float *out;
    hipMalloc(&cyc, NB * 8 * 2 * 8); hipMalloc(&out, NB * 512 * 4);
    std::vector<unsigned long long> h(NB * 8 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# cfg shape kind k | cycles per MFMA (mean / min / max over MFMA waves) | fillers per MFMA issued by the partner (cross only) | wall us | MFMA-wave GHz\n");
    for (const Var &v : vars) {
        if (only && !strstr(v.cfg, only) && !strstr(v.kind, only)) continue;      // argv[1]: a configuration (w1, w2p ..) or a filler kind
        const size_t ldsb = 100 * 1024;                       // more than half a CU's LDS: one workgroup per CU
        hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        hipLaunchKernelGGL(v.fn, dim3(NB), dim3(v.threads), ldsb, 0, cyc, out, nit);
        hipMemset(cyc, 0, NB * 8 * 2 * 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(v.fn, dim3(NB), dim3(v.threads), ldsb, 0, cyc, out, nit);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0, mn = 1e30, mx = 0, part = 0; int n = 0, np = 0;
        const int wpb = v.threads / 64;
        for (int b = 0; b < NB; ++b)
            for (int w = 0; w < wpb; ++w) {
                const double c = (double)h[2 * (b * 8 + w)], m = (double)h[2 * (b * 8 + w) + 1];
                if (v.cross && w >= 4) { part += m; ++np; continue; }
                const double per = c / m;
                sum += per; mn = std::min(mn, per); mx = std::max(mx, per); ++n;
            }
        const double mean = sum / n;
        const double nm = (double)nit * 8;
        printf("%-5s %s %-5s k=%2d | %7.2f %7.2f %7.2f | %6.2f | %8.1f | %.2f\n", v.cfg, v.shape, v.kind, v.k, mean, mn, mx,
               v.cross ? part / np / nm : 0.0, 1000.f * ms, mean * nm / (1000.0 * ms) / 1000.0);
        fflush(stdout);
    }
    return 0;
}
'''


def main():
    variants = []
    for shape, kmax in (("32", 12), ("16", 6)):
        for kind in ("med3", "cvt", "sub", "fma", "mix", "fmix"):
            for k in range(0, kmax + 1):
                if k == 0 and kind != "mix":
                    continue
                variants.append(("w1", shape, kind, k))
        for k in (1, 2):
            variants.append(("w1", shape, "lds", k))
        for k in (5, 10) if shape == "32" else (5,):
            variants.append(("w1", shape, "mixl", k))
        for cfg in ("w2", "w2p"):
            for kind in ("mix", "fma", "fmix"):
                for k in range(0, kmax + 1, 1 if shape == "16" else 2):
                    if k == 0 and kind != "mix":
                        continue
                    variants.append((cfg, shape, kind, k))
        for cfg in ("w2x", "w2xp"):
            for kind in ("mix", "fma"):
                variants.append((cfg, shape, kind, 8))
    print(HEADER)
    names = []
    for cfg, shape, kind, k in variants:
        name = f"k_{cfg}_{shape}_{kind}_{k}"
        names.append((name, cfg, shape, kind, k))
        print(kernel(name, cfg, shape, kind, k))
    print("static const Var vars[] = {")
    for name, cfg, shape, kind, k in names:
        threads, _, cross = CONFIGS[cfg]
        print(f'    {{"{cfg}", "{shape}", "{kind}", {k}, {threads}, {int(cross)}, {name}}},')
    print("};")
    print(MAIN)


if __name__ == "__main__":
    main()
