// Two waves on one SIMD: wave A issues only 32x32x16 bf16 MFMAs, wave B only vector-ALU (or LDS) instructions.  How much does each
// slow the other down?   hipcc --offload-arch=gfx950 -O3 probe_mfma_sibling.hip -o probe_mfma_sibling && ./probe_mfma_sibling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0 v_fma_f32, 1 v_cvt_pk_bf16_f32, 2 v_pk_add_f32, 3 ds_write_b64, 4 v_and_b32, 5 v_sub_f32
template <int KIND, bool MF_ON, bool V_ON>
__global__ void __launch_bounds__(512, 1) k_probe(unsigned long long *out, float *sink, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    const int wv = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wv < 4) {
        if (MF_ON)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            }
    } else if (V_ON) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 48; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k % 12]) : "v"(v[(k + 5) % 12]));
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[k % 12]) : "v"(v[(k + 5) % 12]));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&v[2 * (k % 6)]) : "v"(*(double *)&v[2 * ((k + 1) % 6)]));
                if (KIND == 3) asm volatile("ds_write_b64 %0, %1" :: "v"((threadIdx.x & 255) * 8 + (k & 7) * 2048), "v"(*(double *)&v[2 * (k % 6)]) : "memory");
                if (KIND == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[k % 12]) : "v"(v[(k + 5) % 12]));
                if (KIND == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[k % 12]) : "v"(v[(k + 5) % 12]));
            }
            if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int KIND, bool MF_ON, bool V_ON>
static void run(const char *name, int grid) {
    const int iters = 1000;
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, grid * 8 * 8); (void)hipMalloc(&sink, 4);
    hipLaunchKernelGGL((k_probe<KIND, MF_ON, V_ON>), dim3(grid), dim3(512), 0, 0, d, sink, iters);
    hipLaunchKernelGGL((k_probe<KIND, MF_ON, V_ON>), dim3(grid), dim3(512), 0, 0, d, sink, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 8);
    (void)hipMemcpy(h.data(), d, grid * 8 * 8, hipMemcpyDeviceToHost);
    double cm = 0, cv = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv) += h[g * 8 + w];
    cm /= grid * 4; cv /= grid * 4;
    printf("%-20s mfma wave %d, other wave %d: %6.1f ticks per MFMA, %6.2f ticks per %s\n", name, (int)MF_ON, (int)V_ON, MF_ON ? cm / (iters * 8.0) : 0.0,
           V_ON ? cv / (iters * 48.0) : 0.0, name);
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    const int grid = 256;
    run<0, true, false>("v_fma_f32", grid);
    run<0, false, true>("v_fma_f32", grid); run<0, true, true>("v_fma_f32", grid);
    run<1, false, true>("v_cvt_pk_bf16_f32", grid); run<1, true, true>("v_cvt_pk_bf16_f32", grid);
    run<2, false, true>("v_pk_add_f32", grid); run<2, true, true>("v_pk_add_f32", grid);
    run<4, false, true>("v_and_b32", grid); run<4, true, true>("v_and_b32", grid);
    run<5, false, true>("v_sub_f32", grid); run<5, true, true>("v_sub_f32", grid);
    run<3, false, true>("ds_write_b64", grid); run<3, true, true>("ds_write_b64", grid);
    return 0;
}
