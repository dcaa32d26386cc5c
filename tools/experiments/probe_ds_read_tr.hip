// EXPERIMENT: what does ds_read_b64_tr_b16 return?  LDS holds element i = i (16-bit); every lane passes an address and gets four
// 16-bit elements back.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/experiments/probe_ds_read_tr.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k_tr(short *out, int mode) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    int off = 0;                                       // element offset of this lane's address
    if (mode == 1) off = lane;                         // consecutive elements
    if (mode == 2) off = 4 * lane;                     // consecutive 8-byte groups
    if (mode == 3) off = 64 * (lane >> 4);             // one base per 16-lane group, same within the group
    if (mode == 4) off = 64 * (lane >> 4) + (lane & 15);
    if (mode == 5) off = 16 * (lane & 15) + 1024 * (lane >> 4);   // lane = row of a [16][16] block, groups 1024 apart
    if (mode == 6) off = 136 * (lane & 15) + 4 * (lane >> 4);     // row-major plane with stride 136: lane = row, group = 4-column block
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + off));
    for (int j = 0; j < 4; ++j) out[4 * lane + j] = v[j];
}
int main() {
    short *d, h[256];
    hipMalloc(&d, sizeof h);
    for (int mode = 0; mode <= 6; ++mode) {
        hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  l%02d: %5d %5d %5d %5d", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
