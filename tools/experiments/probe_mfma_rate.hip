// Issue rate of the bf16 MFMAs on gfx950: cycles per instruction for independent and dependent chains, one or two waves per SIMD,
// one workgroup or the whole chip busy.   hipcc --offload-arch=gfx950 -O3 probe_mfma_rate.hip -o probe_mfma_rate && ./probe_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool BIG>
__global__ void __launch_bounds__(1024) k_rate(unsigned long long *out, float *sink, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
    f32x16 acc32[4];
    f32x4 acc16[4];
    for (int c = 0; c < 4; ++c) { for (int r = 0; r < 16; ++r) acc32[c][r] = 0.f; for (int r = 0; r < 4; ++r) acc16[c][r] = 0.f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (BIG) acc32[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[u % CHAINS], 0, 0, 0);
            else acc16[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[u % CHAINS], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) { for (int r = 0; r < 16; ++r) s += acc32[c][r]; for (int r = 0; r < 4; ++r) s += acc16[c][r]; }
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS, bool BIG>
static void run(const char *name, int grid, int block, int iters) {
    unsigned long long *d; float *sink;
    hipMalloc(&d, grid * 16 * 8); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<CHAINS, BIG>), dim3(grid), dim3(block), 0, 0, d, sink, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<CHAINS, BIG>), dim3(grid), dim3(block), 0, 0, d, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 16);
    hipMemcpy(h.data(), d, grid * 16 * 8, hipMemcpyDeviceToHost);
    double cyc = 0; int n = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < block / 64; ++w) { cyc += h[g * 16 + w]; ++n; }
    cyc /= n;
    const double per = cyc / (iters * 8.0), flop = (BIG ? 32768.0 : 16384.0) * iters * 8.0 * (block / 64) * grid;
    printf("%-34s grid %5d block %4d: %6.1f counter ticks / MFMA / wave, %.3f ms, %.1f TFLOP/s, counter %.2f GHz\n", name, grid, block, per, ms, flop / ms * 1e-9,
           cyc / ms * 1e-6);
    hipFree(d); hipFree(sink);
}

int main() {
    const int it = 4000;
    for (int grid : {1, 256, 1024}) {
        run<4, true>("32x32x16 4 chains", grid, 256, it);
        run<2, true>("32x32x16 2 chains", grid, 256, it);
        run<1, true>("32x32x16 1 chain", grid, 256, it);
        run<4, true>("32x32x16 4 chains, 2 waves/SIMD", grid, 512, it);
        run<1, true>("32x32x16 1 chain, 2 waves/SIMD", grid, 512, it);
        run<4, false>("16x16x32 4 chains", grid, 256, it);
        run<1, false>("16x16x32 1 chain", grid, 256, it);
        run<4, false>("16x16x32 4 chains, 2 waves/SIMD", grid, 512, it);
        run<1, false>("16x16x32 1 chain, 2 waves/SIMD", grid, 512, it);
    }
    return 0;
}
