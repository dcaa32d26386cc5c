// Do the matrix pipe and the vector pipe of a SIMD run side by side when the work comes from two DIFFERENT waves?  And inside one wave?
//   hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip && ./coexec
// mode 0: every wave runs N dependent-chain MFMAs (16x16x32 f16);  mode 1: every wave runs M fp32 FMAs (4 independent chains);
// mode 2: waves 0-3 of a workgroup run the MFMAs, waves 4-7 the FMAs (one of each per SIMD);  mode 3: every wave runs both, interleaved
// in program order (3 MFMA, then 12 FMA);  mode 4: every wave runs both, one after the other.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NIT 4096
__device__ __forceinline__ void mfma_block(f32x4 &d, h16x8 a, h16x8 b) {
#pragma unroll
    for (int i = 0; i < 3; ++i) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
}
__device__ __forceinline__ void fma_block(float (&x)[4], float y) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = fmaf(x[c], y, 1.0f);
}
extern "C" __global__ void __launch_bounds__(512) k(float *out, int mode) {
    const int wv = threadIdx.x >> 6;
    h16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    float x[4] = {1.f, 2.f, 3.f, 4.f};
    const float y = 0.999f + 1e-6f * threadIdx.x;
    const bool do_m = mode == 0 || (mode == 2 && wv < 4), do_f = mode == 1 || (mode == 2 && wv >= 4);
    if (do_m) for (int it = 0; it < NIT; ++it) { mfma_block(d, a, b); asm volatile("" : "+v"(d)); }
    if (do_f) for (int it = 0; it < NIT; ++it) { fma_block(x, y); asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); }
    if (mode == 3) for (int it = 0; it < NIT; ++it) { mfma_block(d, a, b); fma_block(x, y); asm volatile("" : "+v"(d), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); }
    if (mode == 5 || mode == 6) {                                  // 5: waves 0-3 run four INDEPENDENT MFMA chains, waves 4-7 the FMAs; 6: the MFMA waves alone
        if (wv < 4) {
            f32x4 d1 = d, d2 = d, d3 = d;
            for (int it = 0; it < NIT / 4; ++it) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0);
                    d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0);
                }
                asm volatile("" : "+v"(d), "+v"(d1), "+v"(d2), "+v"(d3));
            }
            d += d1 + d2 + d3;
        } else if (mode == 5) {
            for (int it = 0; it < NIT; ++it) { fma_block(x, y); asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); }
        }
    }
    if (mode == 7 && wv >= 4) for (int it = 0; it < NIT; ++it) { fma_block(x, y); asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); }   // the FMA waves alone
    if (mode == 8 && wv < 4) for (int it = 0; it < NIT; ++it) { mfma_block(d, a, b); asm volatile("" : "+v"(d)); }                                          // the (dependent-chain) MFMA waves alone
    if (mode == 4) {
        for (int it = 0; it < NIT; ++it) { mfma_block(d, a, b); asm volatile("" : "+v"(d)); }
        for (int it = 0; it < NIT; ++it) { fma_block(x, y); asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); }
    }
    out[blockIdx.x * 512 + threadIdx.x] = d[0] + d[1] + d[2] + d[3] + x[0] + x[1] + x[2] + x[3];
}
int main() {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 9; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.1f us per launch (one 512-thread workgroup per CU: 2 waves per SIMD; %d x [3 MFMA 16x16x32 | 12 FMA] per wave)\n", mode, 1000.f * ms / 5, NIT);
    }
    return 0;
}
