// v_fma_mixlo_f16 as the split's conversion -- scale, convert (and pack: mixhi writes the upper half) in ONE instruction, the fp16 hi part
// read back as a source of the lo part's instruction -- with MODE.FP16_OVFL = 1 for the saturation: does it give the bits of the
// reference sequence (multiply, v_med3, convert, convert back, subtract, convert)?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mixprobe mixprobe.hip && ./mixprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
__global__ void k(const float *x, unsigned *ref, unsigned *got, int n, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[i];
    // reference: x' = 2^-6 a bounded to [0, 65504]; hi = fp16(x'), lo = fp16(x' - hi)
    const float xs = __builtin_amdgcn_fmed3f(a * 0.015625f, 0.f, 65504.0f);
    const _Float16 h = (_Float16)xs, l = (_Float16)(xs - (float)h);
    ref[i] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
    const float y = fmaxf(a, 0.f), s = 0.015625f;
    unsigned hi = 0, lo = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(y), "v"(s));
    // lo = fp16(y * s - hi): src2 read as the fp16 low half of `hi`, negated
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(y), "v"(s), "v"(hi));
    got[i] = (hi & 0xffffu) | (lo << 16);
}
int main() {
    std::vector<float> x;
    for (int e = -60; e <= 40; ++e)
        for (int m = 0; m < 64; ++m) { float v = ldexpf(1.0f + m / 64.0f + 1e-4f * m, e); x.push_back(v); x.push_back(-v); }
    x.push_back(0.f); x.push_back(65504.f * 64); x.push_back(65520.f * 64); x.push_back(1e30f); x.push_back(3.4e38f);
    srand(1);
    for (int i = 0; i < 200000; ++i) x.push_back(ldexpf((float)rand() / RAND_MAX, (rand() % 40) - 20));
    const int n = x.size();
    float *dx; unsigned *dr, *dg;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dr, n * 4); (void)hipMalloc(&dg, n * 4);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ++ov) {
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dr, dg, n, ov);
        std::vector<unsigned> r(n), g(n);
        (void)hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(g.data(), dg, n * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i)
            if (r[i] != g[i]) {
                if (bad < 12) printf("  ovfl %d x = %.9g (2^-6 x = %.9g): ref %08x got %08x\n", ov, x[i], x[i] / 64, r[i], g[i]);
                ++bad;
            }
        printf("FP16_OVFL = %d: %d of %d differ\n", ov, bad, n);
    }
    return 0;
}
