// EXPERIMENT: the bf16-split distance contraction of ndp_nn_matrix.inc on one 32 x 32 tile, against the exact distances.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ void split3(float x, __bf16 &hi, __bf16 &mid, __bf16 &lo) {
    hi = (__bf16)x; const float r1 = x - (float)hi; mid = (__bf16)r1; lo = (__bf16)(r1 - (float)mid);
}
__global__ void k(const float *xs, const float *ys, float *out) {
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    float x[3] = {xs[3 * l31], xs[3 * l31 + 1], xs[3 * l31 + 2]}, y[3] = {ys[3 * l31], ys[3 * l31 + 1], ys[3 * l31 + 2]};
    const float nx = fmaf(x[2], x[2], fmaf(x[1], x[1], x[0] * x[0])), ny = fmaf(y[2], y[2], fmaf(y[1], y[1], y[0] * y[0]));
    __bf16 sa[32], sb[32];
    for (int c = 0; c < 3; ++c) {
        __bf16 p, m, l; split3(x[c], p, m, l);
        sa[6 * c] = p; sa[6 * c + 1] = p; sa[6 * c + 2] = m; sa[6 * c + 3] = p; sa[6 * c + 4] = l; sa[6 * c + 5] = m;
        split3(-2.0f * y[c], p, m, l);
        sb[6 * c] = p; sb[6 * c + 1] = m; sb[6 * c + 2] = p; sb[6 * c + 3] = l; sb[6 * c + 4] = p; sb[6 * c + 5] = m;
    }
    { __bf16 p, m, l; split3(nx, p, m, l); sa[18] = p; sa[19] = m; sa[20] = l; sa[21] = sa[22] = sa[23] = (__bf16)1.0f;
      split3(ny, p, m, l); sb[18] = sb[19] = sb[20] = (__bf16)1.0f; sb[21] = p; sb[22] = m; sb[23] = l; }
    for (int q = 24; q < 32; ++q) { sa[q] = (__bf16)0.f; sb[q] = (__bf16)0.f; }
    bf16x8 A0, A1, B0, B1;
    for (int e = 0; e < 8; ++e) { A0[e] = h ? sa[8 + e] : sa[e]; A1[e] = h ? sa[24 + e] : sa[16 + e]; B0[e] = h ? sb[8 + e] : sb[e]; B1[e] = h ? sb[24 + e] : sb[16 + e]; }
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[r];     // [source row][target col]
}
int main() {
    float hx[96], hy[96], ho[1024], *dx, *dy, *dout;
    srand(1); for (int i = 0; i < 96; ++i) { hx[i] = rand() / (float)RAND_MAX - 0.5f; hy[i] = rand() / (float)RAND_MAX - 0.5f; }
    hipMalloc(&dx, sizeof hx); hipMalloc(&dy, sizeof hy); hipMalloc(&dout, sizeof ho);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dy, hy, sizeof hy, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    double worst = 0; int wi = 0, wj = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double d = 0; for (int c = 0; c < 3; ++c) { double t = (double)hx[3 * i + c] - hy[3 * j + c]; d += t * t; }
        double e = fabs(ho[i * 32 + j] - d); if (e > worst) { worst = e; wi = i; wj = j; }
    }
    printf("worst |approx - exact| = %.3e at (%d, %d): approx %.6f\n", worst, wi, wj, ho[wi * 32 + wj]);
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) { double d = 0; for (int c = 0; c < 3; ++c) { double t = (double)hx[3*i+c] - hy[3*j+c]; d += t*t; } printf("  %.5f/%.5f", ho[i*32+j], d); } printf("\n"); }
    return 0;
}
