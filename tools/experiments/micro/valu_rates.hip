// Issue cost of the vector instructions the level kernels are made of (gfx950, wave64): cycles per instruction and SIMD with two waves
// per SIMD, eight independent chains per wave.   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define NIT 2048
#define OP8(STR)                                                                                                   \
    for (int it = 0; it < NIT; ++it)                                                                               \
        asm volatile(STR(%0) "\n" STR(%1) "\n" STR(%2) "\n" STR(%3) "\n" STR(%4) "\n" STR(%5) "\n" STR(%6) "\n" STR(%7) \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "v"(z));
#define S_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9"
#define S_MUL(r) "v_mul_f32 " #r ", " #r ", %8"
#define S_ADD(r) "v_add_f32 " #r ", " #r ", %8"
#define S_MED3(r) "v_med3_f32 " #r ", " #r ", %8, %9"
#define S_MAX(r) "v_max_f32 " #r ", " #r ", %8"
#define S_CVT16(r) "v_cvt_f16_f32 " #r ", " #r
#define S_CVT32(r) "v_cvt_f32_f16 " #r ", " #r
#define S_CVTPK(r) "v_cvt_pk_f16_f32 " #r ", " #r ", %8"
#define S_CND(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc"
#define S_CMP(r) "v_cmp_lt_f32 vcc, " #r ", %8"
#define S_MOV(r) "v_mov_b32 " #r ", %8"
#define S_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 1, %8"
#define S_ADDU(r) "v_add_u32 " #r ", " #r ", %8"
#define S_OR(r) "v_or_b32 " #r ", " #r ", %8"
#define S_SDWA(r) "v_cvt_f32_f16_sdwa " #r ", " #r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define S_FMAC(r) "v_fmac_f32 " #r ", %8, %9"
#define S_MIN(r) "v_min_f32 " #r ", " #r ", %8"
#define S_SUB(r) "v_sub_f32 " #r ", " #r ", %8"
#define S_AND(r) "v_and_b32 " #r ", " #r ", %8"
#define S_SHL(r) "v_lshlrev_b32 " #r ", 1, " #r
#define S_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9"
#define S_PKRTZ(r) "v_cvt_pkrtz_f16_f32 " #r ", " #r ", %8"
#define S_PKMULH(r) "v_pk_mul_f16 " #r ", " #r ", %8"
#define S_PKFMAH(r) "v_pk_fma_f16 " #r ", " #r ", %8, %9"
#define S_CMPCND(r) "v_cmp_lt_f32 vcc, " #r ", %8\nv_cndmask_b32 " #r ", " #r ", %9, vcc"
#define S_CNDS(r) "v_cndmask_b32 " #r ", " #r ", %8, s[20:21]"
#define S_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9"
#define S_MULE64(r) "v_mul_f32_e64 " #r ", " #r ", %8"
#define S_FMAK(r) "v_fmamk_f32 " #r ", " #r ", 0x3f000000, %8"
#define S_MULLIT(r) "v_mul_f32 " #r ", 0x3c800000, " #r
#define S_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 5"
typedef float f2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void __launch_bounds__(512) k(float *out, int op) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + 0.001f * (threadIdx.x + i);
    float y = 0.9999f, z = 0.5f;
    asm volatile("" : "+v"(y), "+v"(z));
    switch (op) {
    case 0: OP8(S_FMA) break;
    case 1: OP8(S_MUL) break;
    case 2: OP8(S_ADD) break;
    case 3: OP8(S_MED3) break;
    case 4: OP8(S_MAX) break;
    case 5: OP8(S_CVT16) break;
    case 6: OP8(S_CVT32) break;
    case 7: OP8(S_CVTPK) break;
    case 8: OP8(S_CND) break;
    case 9: OP8(S_CMP) break;
    case 10: OP8(S_MOV) break;
    case 11: OP8(S_LSHLADD) break;
    case 12: OP8(S_ADDU) break;
    case 13: OP8(S_OR) break;
    case 14: OP8(S_SDWA) break;
    case 16: OP8(S_FMAC) break;
    case 17: OP8(S_MIN) break;
    case 18: OP8(S_SUB) break;
    case 19: OP8(S_AND) break;
    case 20: OP8(S_SHL) break;
    case 21: OP8(S_PERM) break;
    case 22: OP8(S_PKRTZ) break;
    case 23: OP8(S_PKMULH) break;
    case 24: OP8(S_PKFMAH) break;
    case 25: OP8(S_CMPCND) break;
    case 26: asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21"); OP8(S_CNDS) break;
    case 27: OP8(S_MAX3) break;
    case 28: OP8(S_MULE64) break;
    case 29: OP8(S_FMAK) break;
    case 30: OP8(S_MULLIT) break;
    case 31: OP8(S_BFE) break;
    case 15: {                                                    // v_pk_fma_f32 on four independent register pairs
        f2 p[4], yy = {y, y}, zz = {z, z};
        for (int i = 0; i < 4; ++i) p[i] = f2{x[2 * i], x[2 * i + 1]};
        for (int it = 0; it < NIT; ++it) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(yy), "v"(zz));
        }
        for (int i = 0; i < 4; ++i) { x[2 * i] = p[i][0]; x[2 * i + 1] = p[i][1]; }
    } break;
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_med3_f32", "v_max_f32", "v_cvt_f16_f32", "v_cvt_f32_f16", "v_cvt_pk_f16_f32", "v_cndmask_b32",
                           "v_cmp_lt_f32", "v_mov_b32", "v_lshl_add_u32", "v_add_u32", "v_or_b32", "v_cvt_f32_f16_sdwa", "v_pk_fma_f32",
                           "v_fmac_f32", "v_min_f32", "v_sub_f32", "v_and_b32", "v_lshlrev_b32", "v_perm_b32", "v_cvt_pkrtz_f16_f32", "v_pk_mul_f16", "v_pk_fma_f16",
                           "v_cmp + v_cndmask (x2)", "v_cndmask (sgpr pair)", "v_max3_f32", "v_mul_f32_e64", "v_fmamk_f32", "v_mul_f32 (literal)", "v_bfe_u32"};
    for (int op = 0; op < 32; ++op) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, op);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, op);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = 1000.0 * ms / 5, per = us * 1e-6 / (2.0 * NIT * 8);       // seconds per instruction and SIMD (two waves per SIMD)
        printf("%-20s %7.1f us  %5.2f ns per instruction and SIMD = %4.2f cycles at 2.4 GHz\n", names[op], us, per * 1e9, per * 2.4e9);
    }
    return 0;
}
