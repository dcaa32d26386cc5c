// How fast does a CU get a 64-point tile's activations (2 x 32 KB) out to HBM, by the SHAPE of the store instructions?  One 512-thread
// workgroup per CU streams `tiles` tiles of 64 KB each; nothing else runs (an upper bound for what the level forward's epilogue can do).
//   pattern 0: what the round-4 forward issues: h1 image as dwordx2 (a wave-instruction covers 16 rows x 32 contiguous bytes) and h2 rows
//              as dwordx4 (16 rows x 64 bytes), per point group, from the accumulator layout
//   pattern 1: both halves as dwordx4 from the accumulator layout (16 rows x 64 bytes per instruction)
//   pattern 2: fully coalesced dwordx4 (a wave-instruction covers 1 KiB contiguous): what a copy out of the LDS planes can issue
//   pattern 3: pattern 2 with `nt` stores
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void __launch_bounds__(512) k(unsigned char *buf, int tiles, int pattern, unsigned long long *cyc) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l15 = lane & 15, lk = lane >> 4;
    unsigned char *base = buf + (size_t)blockIdx.x * tiles * 65536;
    const u32x4 v4 = {(unsigned)t, 1u, 2u, 3u};
    const u32x2 v2 = {(unsigned)t, 7u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < tiles; ++i) {
        unsigned char *tb = base + (size_t)i * 65536;
        if (pattern == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {                            // image: row p = 16 g + l15 (256 B), this wave's 32 bytes at 32 wv + 8 lk, two planes
                unsigned char *d = tb + (16 * g + l15) * 256 + 32 * wv + 8 * lk;
                *reinterpret_cast<u32x2 *>(d) = v2;
                *reinterpret_cast<u32x2 *>(d + 16384) = v2;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)                              // fp32 rows: row p (512 B), this wave's 64 bytes at 64 wv + 16 lk
                *reinterpret_cast<u32x4 *>(tb + 32768 + (16 * g + l15) * 512 + 64 * wv + 16 * lk) = v4;
        } else if (pattern == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<u32x4 *>(tb + 32768 * h + (16 * g + l15) * 512 + 64 * wv + 16 * lk) = v4;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {                            // wave wv: 8 KiB of the tile, 1 KiB per instruction
                u32x4 *d = reinterpret_cast<u32x4 *>(tb + 8192 * wv + 1024 * q + 16 * lane);
                if (pattern == 3) __builtin_nontemporal_store(v4, d);
                else *d = v4;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int NB = 256, tiles = 64;
    unsigned char *buf; unsigned long long *cyc;
    (void)hipMalloc(&buf, (size_t)NB * tiles * 65536); (void)hipMalloc(&cyc, NB * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int pattern = 0; pattern < 4; ++pattern) {
            hipLaunchKernelGGL(k, dim3(NB), dim3(512), 0, 0, buf, tiles, pattern, cyc);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(NB), dim3(512), 0, 0, buf, tiles, pattern, cyc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
            double s = 0; for (int b = 0; b < NB; ++b) s += (double)h[b];
            printf("pattern %d: %.1f us, %.2f TB/s, %.0f cycles per 64 KB tile and CU = %.1f B per cycle and CU\n", pattern, 1000.f * ms,
                   (double)NB * tiles * 65536 / (ms * 1e-3) / 1e12, s / NB / tiles, 65536.0 / (s / NB / tiles));
        }
    return 0;
}
