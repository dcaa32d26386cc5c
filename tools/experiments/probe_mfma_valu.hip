// Does a wave's own vector-ALU work run in the shadow of its MFMAs?  One wave per SIMD, a chain of 32x32x16 bf16 MFMAs with K
// independent VALU instructions (and optionally one LDS read) behind each: counter ticks per MFMA as K grows.
//   hipcc --offload-arch=gfx950 -O3 probe_mfma_valu.hip -o probe_mfma_valu && ./probe_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int KIND, bool MF>
__global__ void __launch_bounds__(256, 1) k_probe(unsigned long long *out, float *sink, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x + i;
    f32x4 l4 = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k % 12]) : "v"(v[(k + 5) % 12]));
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[k % 12]) : "v"(v[(k + 5) % 12]));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&v[2 * (k % 6)]) : "v"(*(double *)&v[2 * ((k + 1) % 6)]));
            }
            if (KIND == 3) { asm volatile("ds_read_b128 %0, %1" : "=v"(l4) : "v"((threadIdx.x & 255) * 16)); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = l4[0] + l4[3];
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K, int KIND, bool MF>
static void run(const char *name, int grid) {
    const int iters = 2000;
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, grid * 4 * 8); (void)hipMalloc(&sink, 4);
    hipLaunchKernelGGL((k_probe<K, KIND, MF>), dim3(grid), dim3(256), 0, 0, d, sink, iters);
    hipLaunchKernelGGL((k_probe<K, KIND, MF>), dim3(grid), dim3(256), 0, 0, d, sink, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    (void)hipMemcpy(h.data(), d, grid * 4 * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto x : h) cyc += x;
    cyc /= h.size();
    printf("%-28s K=%2d mfma=%d grid %4d: %6.1f ticks per (MFMA + K ops)\n", name, K, (int)MF, grid, cyc / (iters * 8.0));
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    for (int grid : {1, 256}) {
        run<0, 0, true>("mfma only", grid);
        run<2, 0, true>("v_fma_f32", grid); run<4, 0, true>("v_fma_f32", grid); run<6, 0, true>("v_fma_f32", grid);
        run<8, 0, true>("v_fma_f32", grid); run<12, 0, true>("v_fma_f32", grid); run<16, 0, true>("v_fma_f32", grid);
        run<8, 0, false>("v_fma_f32 alone", grid); run<16, 0, false>("v_fma_f32 alone", grid);
        run<4, 1, true>("v_cvt_pk_bf16_f32", grid); run<8, 1, true>("v_cvt_pk_bf16_f32", grid); run<8, 1, false>("v_cvt_pk_bf16_f32 alone", grid);
        run<4, 2, true>("v_pk_add_f32", grid); run<8, 2, true>("v_pk_add_f32", grid); run<8, 2, false>("v_pk_add_f32 alone", grid);
        run<0, 3, true>("ds_read_b128", grid); run<0, 3, false>("ds_read_b128 alone", grid);
    }
    return 0;
}
