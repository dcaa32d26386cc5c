// ndp_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the NDP per-pair optimisation
// hot path, and the C ABI declared in include/ndp_hip.h.
//
// Design:
//   * One workgroup = 256 threads = 4 wave64.  A tile is 64 points.  Wave w owns output columns
//     [32w, 32w+32) of every 128-wide layer.
//   * The 128x128 weight matrices are WEIGHT-STATIONARY IN REGISTERS: each lane holds its 64-float
//     slice in the v_mfma_f32_32x32x2_f32 B-operand layout (forward: W[o][k] slices of W1 and W2;
//     backward, one layer per kernel: the transposed slice) for the whole life of the workgroup,
//     so the only per-MFMA operand fetch is one LDS read of the activation.
//   * Activations move through two [64][132] LDS tiles (+4 float pad: ds_read_b128 of a column
//     block is bank-conflict free).  fp32 in, fp32 accumulate: the MFMA result is bitwise an fmaf
//     chain, which is what the 1e-4 parity budget needs.  Everything that is a small GEMM runs on
//     the matrix pipe too (6 -> 128 input layer, the 16-wide heads and the 6-wide input-layer
//     gradient on v_mfma_f32_16x16x4_f32): VALU loops over LDS next to MFMA phases are what a tile
//     used to wait for.
//   * The backward keeps one 128x128 dW in accumulator registers across all of the workgroup's tiles
//     and writes ONE partial per workgroup; partials are folded in index order by the Adam kernel --
//     no float atomics anywhere, results are bit-reproducible.
//   * The batched engine advances B independent pairs per launch, every pair at its own level and
//     iteration; the early-stop rule runs on the device in double, so the host never syncs per
//     iteration (the reference syncs three times: registration.py:226-232).  Slot refill, pair
//     preparation and the final all-point warp are batched single launches as well.
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fmaf only).
#include "ndp_device.h"

#include <cstdio>
#include <cstring>

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// -DNDP_PHASE_TIMING: experiment builds only (tools/phase_timing.py) -- thread 0 of every workgroup adds the shader
// cycles it spent in each phase of a tile to g_phase[]; compiled out of the product library.
#ifdef NDP_PHASE_TIMING
__device__ unsigned long long g_phase[96];
__shared__ unsigned long long pt_acc[12];                 // per-workgroup accumulators (LDS: no global traffic per stamp)
__shared__ unsigned long long pt_clk[2];                  // workgroup (0, 0): shader-cycle and 100 MHz real-time counters at its start
// g_phase[94] / [95]: shader cycles / 100 MHz ticks that workgroup (0, 0) of the instrumented kernels lived -- their ratio x 0.1 is the
// shader clock in GHz the launch actually ran at (tools/phase_timing.py prints it; under a power cap it is far from 2.4)
#define PT_INIT                                           \
    do {                                                  \
        if (threadIdx.x < 12) pt_acc[threadIdx.x] = 0;    \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { pt_clk[0] = __builtin_readcyclecounter(); pt_clk[1] = __builtin_amdgcn_s_memrealtime(); } \
        __syncthreads();                                  \
    } while (0)
#define PT_FLUSH(base)                                                                           \
    do {                                                                                         \
        __syncthreads();                                                                         \
        if (threadIdx.x < 12 && pt_acc[threadIdx.x]) atomicAdd(&g_phase[(base) + threadIdx.x], pt_acc[threadIdx.x]); \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {                            \
            atomicAdd(&g_phase[94], __builtin_readcyclecounter() - pt_clk[0]);                   \
            atomicAdd(&g_phase[95], __builtin_amdgcn_s_memrealtime() - pt_clk[1]);               \
        }                                                                                        \
    } while (0)
#ifndef PT_TID
#define PT_TID 0                                          /* the stamping thread (wave-specialised experiments look at other waves too) */
#endif
#define PT_DECL unsigned long long pt_last = __builtin_readcyclecounter()
/* the branch is wave-uniform and the counter scalar: pt_last lives in two SGPRs (a per-thread copy cost the register-capped kernels \
   two VGPRs over their whole body -- the fused backward's timing build spilled) */                                               \
#define PT(id)                                                                  \
    do {                                                                        \
        if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) == PT_TID) {       \
            const unsigned long long pt_now = __builtin_readcyclecounter();     \
            if (threadIdx.x == PT_TID) pt_acc[(id) % 12] += pt_now - pt_last;   \
            pt_last = pt_now;                                                   \
        }                                                                       \
    } while (0)
#else
#define PT_INIT
#define PT_FLUSH(base)
#define PT_DECL
#define PT(id)
#endif
#ifdef NDP_PHASE_TIMING_FINE                              /* stamps inside a barrier interval (they pin the schedule around them) */
#define PTF(id) PT(id)
#else
#define PTF(id)
#endif

// ------------------------------------------------------------------------------------------------
// LDS carve (floats).  All scratch lives in the dynamic region (16-byte aligned offsets).
// ------------------------------------------------------------------------------------------------
#define NDP_WHROWS 12                     /* head rows staged in LDS (at most 6 + 1 + 3 + 1 = 11 are used) */
enum : int {
    L_BUFA = 0,
    L_BUFB = L_BUFA + 64 * NDP_LD,
    L_HO = L_BUFB,                        // [64][16] head outputs: reuses bufB, which is dead after layer 2
    L_PE = L_BUFB + 64 * NDP_LD,          // 2 x [64][9] posenc, double-buffered across tiles (stride 9: conflict-free)
    L_XS = L_PE + 2 * 64 * 9,             // 2 x [64][4] level input x
    L_WH = L_XS + 2 * 64 * 4,             // [12][NDP_LD] head weights
    L_BH = L_WH + NDP_WHROWS * NDP_LD,    // [16]
    L_FWD_TOTAL = L_BH + NDP_NHMAX
};
static constexpr int kSmemFwdBytes = L_FWD_TOTAL * 4;     // 80 640 B: two workgroups per CU
static_assert(2 * kSmemFwdBytes <= 160 * 1024, "forward LDS carve must allow two workgroups per CU");

struct LevelJob {
    const float *params;
    float freq;
    const float *x_in;
    float *x_out;
    float *act;        // [3][plane][128] or nullptr
    float *heads;      // [plane][NDP_HROW] or nullptr: 16 scaled head outputs + 6 posenc values
    float *nonrig;     // [n] or nullptr: gate value per point (levels with the nonrigidity head)
    int n;             // live points
    int plane;         // rows per activation plane (capacity, multiple of 64)
    int n_tiles;       // live tiles = ceil(n / 64)
    int tile0, tile_step;
};

// lane-resident slice of a 128x128 matrix in the 32x32x2 B-operand layout
//   forward : w[ks] = W[32*wv + l31][64*h + ks]        (contraction index k = 64*h + ks)
//   backward: w[ks] = W[64*h + ks][32*wv + l31]        (contraction index o = 64*h + ks)
// Both go through LDS: the matrix is pulled from L2/HBM by LDS-DMA as 1 KiB blocks (two consecutive rows per instruction,
// perfectly coalesced) into the row-pair padded image the backward tiles use (float index of (r, c) = 260 (r >> 1) +
// 128 (r & 1) + c; 64 pairs = 66 560 B, the two tile buffers of either carve), and the lanes pick their slices out of LDS.
// (Straight from global, a lane's 64 floats are 16 float4 loads that touch 64 different cache lines per instruction --
//  eight times the line requests the data needs: 27-28k cycles of prologue per workgroup, and per LEVEL in the final warp.)
#define WIMG_PAIR 260
__device__ __forceinline__ int wimg_row(int r) { return WIMG_PAIR * (r >> 1) + NDP_W * (r & 1); }
// wave wv lays down rows 32wv .. 32wv+31 of W (16 row pairs); asynchronous, wait with s_waitcnt vmcnt(0)
__device__ __forceinline__ void wimg_load_rows(const float *W, float *img /*LDS*/, int wv, int lane) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = 16 * wv + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(W + 2 * NDP_W * q + 4 * lane),
                                         (__attribute__((address_space(3))) void *)(img + WIMG_PAIR * q), 16, 0, 0);
    }
}
// forward slice: the rows a wave reads are the rows it loaded itself, so no workgroup barrier is needed -- only its own
// DMA (vmcnt) before the reads, and its own reads (lgkmcnt) before the image is overwritten by the next matrix.
__device__ __forceinline__ void load_w_fwd(const float *W, float *img /*LDS*/, int wv, int l31, int h, float (&w)[64]) {
    wimg_load_rows(W, img, wv, threadIdx.x & 63);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float *src = img + wimg_row(32 * wv + l31) + 64 * h;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * i);
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// backward (transposed) slice: a lane's column crosses the rows of all four waves -> barrier on both sides
__device__ __forceinline__ void load_w_bwd(const float *W, float *img /*LDS*/, int wv, int l31, int h, float (&w)[64]) {
    wimg_load_rows(W, img, wv, threadIdx.x & 63);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float *src = img + 32 * wv + l31;
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = src[wimg_row(64 * h + i)];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// OUT^T[o][p] += sum_k W[o][k] * in[p][k] for the 64 points of a tile: the weight slice is the MFMA A operand
// (row m = this lane's output feature 32wv + l31), the activation row of point l31 (+32) the B operand.  In the
// resulting C layout a lane holds point p = l31 (acc0) / l31 + 32 (acc1) and, per register group g = r >> 2, the FOUR
// CONSECUTIVE output features 32wv + 8g + 4h + (r & 3): epilogues read/write row-major tiles with b128 LDS accesses.
__device__ __forceinline__ void tile_gemm_64x32(const float *in /*LDS [64][LD]*/, const float (&w)[64],
                                                int l31, int h, f32x16 &acc0, f32x16 &acc1) {
    const float *r0 = in + l31 * NDP_LD + 64 * h;
    const float *r1 = r0 + 32 * NDP_LD;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 a0 = *reinterpret_cast<const float4 *>(r0 + 4 * i);
        const float4 a1 = *reinterpret_cast<const float4 *>(r1 + 4 * i);
        acc0 = MFMA32(w[4 * i], a0.x, acc0);     acc1 = MFMA32(w[4 * i], a1.x, acc1);
        acc0 = MFMA32(w[4 * i + 1], a0.y, acc0); acc1 = MFMA32(w[4 * i + 1], a1.y, acc1);
        acc0 = MFMA32(w[4 * i + 2], a0.z, acc0); acc1 = MFMA32(w[4 * i + 2], a1.z, acc1);
        acc0 = MFMA32(w[4 * i + 3], a0.w, acc0); acc1 = MFMA32(w[4 * i + 3], a1.w, acc1);
        // keep the compiler from hoisting all 32 operand reads to the top (64 extra live VGPRs -> spills
        // at the 256-register budget of two workgroups per CU); 4 iterations in flight are plenty
        if ((i & 3) == 3) asm volatile("" ::: "memory");
    }
}
// accumulators that start at the layer's bias: one extra MFMA k-step with A = (bias, 0), B = (1, 0) puts bias[o] into
// every element exactly (0 + b * 1), so the chain stays "bias, then k = 0 .. K-1" without 16 bias registers per lane
__device__ __forceinline__ void acc_init_bias(float bias_lane, int h, f32x16 &acc0, f32x16 &acc1) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    const float a = h == 0 ? bias_lane : 0.f, b = h == 0 ? 1.0f : 0.f;
    acc0 = MFMA32(a, b, z);
    acc1 = MFMA32(a, b, z);
}
// ReLU epilogue of tile_gemm_64x32: [p][32wv + 8g + 4h .. +3] <- max(acc, 0), eight ds_write_b128 per lane
__device__ __forceinline__ void epilogue_relu(const f32x16 &acc0, const f32x16 &acc1, float *out /*LDS [64][LD]*/,
                                              int wv, int l31, int h) {
    float *o0 = out + l31 * NDP_LD + 32 * wv + 4 * h, *o1 = o0 + 32 * NDP_LD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4 *>(o0 + 8 * g) = make_float4(fmaxf(acc0[4 * g], 0.f), fmaxf(acc0[4 * g + 1], 0.f),
                                                                fmaxf(acc0[4 * g + 2], 0.f), fmaxf(acc0[4 * g + 3], 0.f));
        *reinterpret_cast<float4 *>(o1 + 8 * g) = make_float4(fmaxf(acc1[4 * g], 0.f), fmaxf(acc1[4 * g + 1], 0.f),
                                                                fmaxf(acc1[4 * g + 2], 0.f), fmaxf(acc1[4 * g + 3], 0.f));
    }
}
// backward epilogue of tile_gemm_64x32: z = d * [hmask > 0] -> zout (same tile coordinates), b128 reads and writes
__device__ __forceinline__ void epilogue_mask(const f32x16 &d0, const f32x16 &d1, const float *hmask /*LDS*/,
                                              float *zout /*LDS*/, int wv, int l31, int h) {
    const int off = l31 * NDP_LD + 32 * wv + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 m0 = *reinterpret_cast<const float4 *>(hmask + off + 8 * g);
        const float4 m1 = *reinterpret_cast<const float4 *>(hmask + off + 32 * NDP_LD + 8 * g);
        *reinterpret_cast<float4 *>(zout + off + 8 * g) =
            make_float4(m0.x > 0.f ? d0[4 * g] : 0.f, m0.y > 0.f ? d0[4 * g + 1] : 0.f,
                        m0.z > 0.f ? d0[4 * g + 2] : 0.f, m0.w > 0.f ? d0[4 * g + 3] : 0.f);
        *reinterpret_cast<float4 *>(zout + off + 32 * NDP_LD + 8 * g) =
            make_float4(m1.x > 0.f ? d1[4 * g] : 0.f, m1.y > 0.f ? d1[4 * g + 1] : 0.f,
                        m1.z > 0.f ? d1[4 * g + 2] : 0.f, m1.w > 0.f ? d1[4 * g + 3] : 0.f);
    }
}

// [64][128] tile: LDS (row stride NDP_LD) -> global, 8 float4 per thread, fully coalesced
__device__ __forceinline__ void store_tile_from_lds(const float *src /*LDS*/, float *dst /*global [64][128]*/) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = t + 256 * i;
        reinterpret_cast<float4 *>(dst)[idx] = *reinterpret_cast<const float4 *>(src + (idx >> 5) * NDP_LD + 4 * (idx & 31));
    }
}

// C/D layout of v_mfma_f32_32x32x2_f32: reg r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------
// Level forward  (nets.py:111-140)
// ------------------------------------------------------------------------------------------------
// lane-resident operands of one level (weight-stationary for as long as the level lasts)
struct FwdWeights {
    float w1[64], w2[64], w0b[3];
    float bias0, bias1, bias2;
};

// Loads a level's weights into registers and stages its head matrix in LDS (row stride NDP_LD, so that the
// 16x16x4 MFMA B-operand reads are conflict-free).  The caller must pass a barrier before the first head phase
// (the tile's own barriers do) and after the last one before reloading.
__device__ __forceinline__ void fwd_load_weights(const HeadCfg &hc, const float *P, float *sm, FwdWeights &fw) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l31 = lane & 31, h = lane >> 5;
    float *whs = sm + L_WH, *bhs = sm + L_BH;
    const ndp_layer_desc dd = {NDP_W, 2, hc.motion, hc.rotfmt, 0, hc.mlp_scale};
    const float *W0 = P + ndp_off_W0(&dd), *b0 = P + ndp_off_b0(&dd);
    const float *W1 = P + ndp_off_Wi(&dd, 1), *b1 = P + ndp_off_bi(&dd, 1);
    const float *W2 = P + ndp_off_Wi(&dd, 2), *b2 = P + ndp_off_bi(&dd, 2);
    const float *Wh = P + ndp_off_Wi(&dd, 3);      // == ndp_off_Wh for nonrigidity = 0
    const float *bh = Wh + hc.nh * NDP_W;
    load_w_fwd(W1, sm + L_BUFA, wv, l31, h, fw.w1);
    load_w_fwd(W2, sm + L_BUFA, wv, l31, h, fw.w2);
    fw.bias1 = b1[32 * wv + l31];
    fw.bias2 = b2[32 * wv + l31];
    // layer 0 (6 -> 128) also runs on the matrix pipe: K = 6 = 3 k-steps of the 32x32x2 MFMA
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) fw.w0b[ks] = W0[(32 * wv + l31) * 6 + 2 * ks + h];
    fw.bias0 = b0[32 * wv + l31];
    for (int i = t; i < NDP_WHROWS * NDP_W; i += 256)
        whs[(i >> 7) * NDP_LD + (i & 127)] = (i < hc.nh * NDP_W) ? Wh[i] : 0.f;
    if (t < NDP_NHMAX) bhs[t] = (t < hc.nh) ? bh[t] : 0.f;
}

// where one tile's input comes from and where its outputs go
struct TileIO {
    const float *x_in;      // global [n][3], or nullptr: the tile's input already sits in LDS (xs)
    const float *shift_in;  // [3] subtracted from x_in (or nullptr)
    float *x_out;           // global [n][3], or nullptr: the warped points replace xs (next level reads them)
    const float *shift_out; // [3] added to x_out (or nullptr)
    float *act;             // [3][plane][128] or nullptr
    float *heads;           // [plane][NDP_HROW] or nullptr
    float *nonrig;          // [n] or nullptr
    int n, plane;
};

// level input of point `base + lane`, coordinate `axis` (zero beyond n)
__device__ __forceinline__ float fwd_fetch_x(const TileIO &io, int base, int lane, int axis) {
    const int p = base + lane;
    if (p >= io.n) return 0.f;
    const float xa = io.x_in[3 * (size_t)p + axis];
    return io.shift_in ? xa - io.shift_in[axis] : xa;
}

// positional encoding (nets.py:164-177) of one coordinate of one point -> pe / xs of the given LDS set
__device__ __forceinline__ void fwd_posenc(float xa, float freq, int lane, int axis, float *pe, float *xs, bool write_x) {
    float sn, cs;
    sincosf(xa * freq, &sn, &cs);
    pe[lane * 9 + 2 * axis] = sn;
    pe[lane * 9 + 2 * axis + 1] = cs;
    if (write_x) xs[4 * lane + axis] = xa;
}

// One 64-point tile from its positional encoding (pe) to the scaled head outputs (ho, which reuses bufB):
// 3 layers on the 32x32x2 MFMA, heads on the 16x16x4 MFMA; activations -> HBM.  Ends with a barrier.
__device__ __forceinline__ void fwd_tile_core(const HeadCfg &hc, const FwdWeights &fw, const TileIO &io, int base,
                                              float *sm, const float *pe) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l31 = lane & 31, h = lane >> 5;
    float *bufA = sm + L_BUFA, *bufB = sm + L_BUFB;
    float *whs = sm + L_WH, *bhs = sm + L_BH, *ho = sm + L_HO;
    PT_DECL;
    // ---- layer 0 (MFMA, bitwise the k = 0..5 fmaf chain starting from the bias) -> bufA
    {
        f32x16 acc0, acc1;
        acc_init_bias(fw.bias0, h, acc0, acc1);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const float a0 = pe[l31 * 9 + 2 * ks + h], a1 = pe[(l31 + 32) * 9 + 2 * ks + h];
            acc0 = MFMA32(fw.w0b[ks], a0, acc0);
            acc1 = MFMA32(fw.w0b[ks], a1, acc1);
        }
        epilogue_relu(acc0, acc1, bufA, wv, l31, h);
    }
    PT(2);
    __syncthreads();
    PT(3);
    // ---- layer 1 (MFMA) bufA -> bufB ; h0 goes to HBM as float4 rows while the matrix pipe works
    {
        f32x16 acc0, acc1;
        acc_init_bias(fw.bias1, h, acc0, acc1);
        if (io.act) store_tile_from_lds(bufA, io.act + (size_t)base * NDP_W);
        tile_gemm_64x32(bufA, fw.w1, l31, h, acc0, acc1);
        epilogue_relu(acc0, acc1, bufB, wv, l31, h);
    }
    PT(4);
    __syncthreads();
    PT(5);
    // ---- layer 2 (MFMA) bufB -> bufA ; h1 -> HBM
    {
        f32x16 acc0, acc1;
        acc_init_bias(fw.bias2, h, acc0, acc1);
        if (io.act) store_tile_from_lds(bufB, io.act + ((size_t)io.plane + base) * NDP_W);
        tile_gemm_64x32(bufB, fw.w2, l31, h, acc0, acc1);
        epilogue_relu(acc0, acc1, bufA, wv, l31, h);
    }
    PT(6);
    __syncthreads();                                                                          // bufB (h1) is dead from here
    PT(7);
    if (io.act) store_tile_from_lds(bufA, io.act + (2 * (size_t)io.plane + base) * NDP_W);   // h2 -> HBM
    PT(11);
    // ---- heads (nets.py:117,125,146) on the 16x16x4 MFMA: wave w owns points 16w..16w+15, the 16 columns are
    //      the head rows (zero beyond nh).  A[p][k]: lane = p + 16*(k mod 4 group); D[p][j]: lane = j + 16*(p/4).
    {
        const int l15 = lane & 15, lk = lane >> 4;
        f32x4 acc;
        const float bj = bhs[l15];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = bj;
        const float *arow = bufA + (16 * wv + l15) * NDP_LD + 4 * lk;
        const float *brow = whs + (l15 < NDP_WHROWS ? l15 : 0) * NDP_LD + 4 * lk;
        const bool live = l15 < NDP_WHROWS;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 a = *reinterpret_cast<const float4 *>(arow + 16 * q);
            float4 b = *reinterpret_cast<const float4 *>(brow + 16 * q);
            if (!live) b = make_float4(0.f, 0.f, 0.f, 0.f);
            acc = MFMA16(a.x, b.x, acc);
            acc = MFMA16(a.y, b.y, acc);
            acc = MFMA16(a.z, b.z, acc);
            acc = MFMA16(a.w, b.w, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ho[(16 * wv + 4 * lk + r) * NDP_NHMAX + l15] = hc.mlp_scale * acc[r];
    }
    PT(8);
    __syncthreads();
    PT(9);
}

// warp (nets.py:119-129) of the tile's 64 points by ONE wave (lane = point): reads ho / xs / pe of the tile.
__device__ __forceinline__ void fwd_warp(const HeadCfg &hc, const TileIO &io, int base, int lane, const float *ho,
                                         const float *pe, float *xs) {
    const int p = base + lane;
    const float *o = ho + lane * NDP_NHMAX;
    if (io.heads) {
        float *hr = io.heads + (size_t)p * NDP_HROW;
#pragma unroll
        for (int j = 0; j < NDP_NHMAX; j += 4)
            *reinterpret_cast<float4 *>(hr + j) = *reinterpret_cast<const float4 *>(o + j);
        const float *pr = pe + lane * 9;
        *reinterpret_cast<float4 *>(hr + 16) = make_float4(pr[0], pr[1], pr[2], pr[3]);
        *reinterpret_cast<float4 *>(hr + 20) = make_float4(pr[4], pr[5], 0.f, 0.f);
    }
    if (p < io.n) {
        PointHead c;
        float out[3];
        head_warp_fwd(hc, o, xs + 4 * lane, c, out);
        if (io.x_out) {
            if (io.shift_out) { out[0] += io.shift_out[0]; out[1] += io.shift_out[1]; out[2] += io.shift_out[2]; }
            io.x_out[3 * (size_t)p] = out[0]; io.x_out[3 * (size_t)p + 1] = out[1]; io.x_out[3 * (size_t)p + 2] = out[2];
        } else {
            xs[4 * lane] = out[0]; xs[4 * lane + 1] = out[1]; xs[4 * lane + 2] = out[2];
        }
        if (io.nonrig) io.nonrig[p] = c.nr;
    }
}

// All tiles of a workgroup through one level.  Software pipeline across tiles: while wave 0 warps tile i, waves
// 1..3 (one per coordinate axis) encode tile i+1 into the other pe/xs set from an x value they fetched at the top
// of tile i, so neither the x load latency nor sincosf sits on the critical path.
__device__ __forceinline__ void level_fwd_body(const HeadCfg &hc, const LevelJob &job, float *sm) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    FwdWeights fw;
    fwd_load_weights(hc, job.params, sm, fw);
    TileIO io;
    io.x_in = job.x_in; io.shift_in = nullptr; io.x_out = job.x_out; io.shift_out = nullptr;
    io.act = job.act; io.heads = job.heads; io.nonrig = job.nonrig; io.n = job.n; io.plane = job.plane;
    float *pe = sm + L_PE, *xs = sm + L_XS, *ho = sm + L_HO;
    int tile = job.tile0, cur = 0;
    if (tile >= job.n_tiles) return;
    if (wv > 0) fwd_posenc(fwd_fetch_x(io, tile * NDP_TILE, lane, wv - 1), job.freq, lane, wv - 1, pe, xs, true);
    __syncthreads();
    for (; tile < job.n_tiles; tile += job.tile_step) {
        const int next = tile + job.tile_step;
        float xn = 0.f;
        if (wv > 0 && next < job.n_tiles) xn = fwd_fetch_x(io, next * NDP_TILE, lane, wv - 1);
        fwd_tile_core(hc, fw, io, tile * NDP_TILE, sm, pe + cur * (64 * 9));
        if (wv == 0) fwd_warp(hc, io, tile * NDP_TILE, lane, ho, pe + cur * (64 * 9), xs + cur * (64 * 4));
        else if (next < job.n_tiles)
            fwd_posenc(xn, job.freq, lane, wv - 1, pe + (cur ^ 1) * (64 * 9), xs + (cur ^ 1) * (64 * 4), true);
        __syncthreads();
        cur ^= 1;
    }
}

// Whole pyramid for one 64-point tile per workgroup: the points stay in LDS from level to level, the weights of
// each level are re-read from L2 (135 KB per level; the grid is sized so that several clouds fill the chip).
struct WarpJobs {
    ndp_warp_job j[NDP_MAX_WARP_JOBS];
};
// A workgroup carries TWO 64-point tiles through all m levels (the two posenc / point sets of the LDS carve): the 135 KB of
// a level's weights are pulled from L2 once per 128 points instead of once per 64 -- the weight prologue was half of the
// kernel -- and the posenc of the second tile overlaps the warp of the first exactly as in the level kernel.
#define NDP_PYR_TILES 2
extern "C" __global__ void __launch_bounds__(256, 2)
k_pyramid_fwd(ndp_layer_desc desc, int m, int k0, int p_stride, WarpJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const ndp_warp_job jb = jobs.j[blockIdx.y];
    const int base0 = blockIdx.x * NDP_TILE * NDP_PYR_TILES;
    if (base0 >= jb.n) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    float *pe = sm + L_PE, *xs = sm + L_XS, *ho = sm + L_HO;
    const bool two = base0 + NDP_TILE < jb.n;                      // the second tile holds points
    TileIO io;
    io.act = nullptr; io.heads = nullptr; io.nonrig = nullptr; io.n = jb.n; io.plane = 0;
    io.x_in = jb.x; io.shift_in = jb.shift_in;
    for (int l = 0; l < m; ++l) {
        const HeadCfg hc = make_head_cfg(desc_at_level(desc, l));
        FwdWeights fw;
        fwd_load_weights(hc, jb.params + (size_t)l * p_stride, sm, fw);
        io.x_out = l == m - 1 ? jb.x_out : nullptr;
        io.shift_out = l == m - 1 ? jb.shift_out : nullptr;
        const float freq = ldexpf(1.0f, l + 1 + k0);
        if (wv > 0) {
            const float xa = l == 0 ? fwd_fetch_x(io, base0, lane, wv - 1) : xs[4 * lane + wv - 1];
            fwd_posenc(xa, freq, lane, wv - 1, pe, xs, l == 0);
        }
        __syncthreads();
        fwd_tile_core(hc, fw, io, base0, sm, pe);
        if (wv == 0) fwd_warp(hc, io, base0, lane, ho, pe, xs);
        else if (two) {                                             // second tile's encoding while wave 0 warps the first
            const float xa = l == 0 ? fwd_fetch_x(io, base0 + NDP_TILE, lane, wv - 1) : xs[64 * 4 + 4 * lane + wv - 1];
            fwd_posenc(xa, freq, lane, wv - 1, pe + 64 * 9, xs + 64 * 4, l == 0);
        }
        __syncthreads();
        if (two) {
            fwd_tile_core(hc, fw, io, base0 + NDP_TILE, sm, pe + 64 * 9);
            if (wv == 0) fwd_warp(hc, io, base0 + NDP_TILE, lane, ho, pe + 64 * 9, xs + 64 * 4);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Level backward (autograd of nets.py:111-140 wrt the level's parameters), split by layer so that each
// kernel keeps only ONE 128x128 weight slice + ONE 128x128 gradient accumulator in registers
// (<= 256 VGPR+AGPR per lane => two workgroups per CU, whose load / VALU / MFMA phases overlap):
//   bwd2: dO -> dz2 = (dO Wh) * [h2>0] ; dWh += dO^T h2 ; dbh ; dW2 += dz2^T h1 ; db2 ; dh1 = dz2 W2 ;
//         dz1 = dh1 * [h1>0]  -> written over the (now dead) h2 plane of the activation store
//   bwd1: dW1 += dz1^T h0 ; db1 ; dh0 = dz1 W1 ; dz0 = dh0 * [h0>0] ; [dW0 | db0] += dz0^T [pe | 1]
// The per-point head backward (dO) is done before, one thread per point (k_head_bwd / k_eng_loss).
//
// LDS tiles of the backward are filled by LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base + 16 B x lane, so one
// instruction lays down 1 KiB = two consecutive rows of the [64][128] tile, contiguously).  The image is therefore padded
// per ROW PAIR, not per row:   float index of (row r, column c) = 260 (r >> 1) + 128 (r & 1) + c
// (16 B of pad after every 1 KiB block).  Everything stays base + immediate (an XOR swizzle of the 16-byte chunks needs a
// VGPR per address and spilled), the 16-lane groups of a ds_read_b128 over 16 rows hit 8 distinct 16-B slots (2-way, noise
// next to 64-cycle MFMAs), and the b32 operand reads of one row are conflict-free.  No staging registers (the register-
// staged loads had started to serialise -- one load in flight at a time -- once the head stage's accumulators moved into
// bwd2 at the 256-register cap), no ds_write pass, and the global side is perfectly linear: lane l of block q reads
// src + 1 KiB q + 16 B l.
// ------------------------------------------------------------------------------------------------
#define BP_PAIR 260                       /* floats per row pair: 2 x 128 + 4 pad */
#define BP_TILE (32 * BP_PAIR)            /* floats per [64][128] tile image */
#define NDP_PES 74                        /* posenc row stride in bwd1: banks 10 c + 4 lk never collide for c < 6, lk < 2 */
enum : int {
    LB_BUFA = 0,
    LB_BUFB = LB_BUFA + BP_TILE,
    LB_DO = LB_BUFB + BP_TILE,            // [64][17] (stride 17: conflict-free MFMA operand reads over the points)
    LB_PE = LB_DO + 64 * 17,              // [6][NDP_PES] posenc rows (bwd1)
    LB_WH = LB_DO + 64 * 17,              // [NDP_WHROWS][128] head matrix, rows >= nh zero (bwd2; shares the posenc slot of bwd1)
    LB_TOTAL = LB_WH + NDP_WHROWS * NDP_W
};
static_assert(6 * NDP_PES <= NDP_WHROWS * NDP_W, "posenc rows must fit the shared slot");
static constexpr int kSmemBwdBytes = LB_TOTAL * 4;       // 77.1 KB: two workgroups per CU
static_assert(2 * kSmemBwdBytes <= 160 * 1024, "backward LDS carve must allow two workgroups per CU");

struct BwdJob {
    const float *params;
    float *act;             // [3][plane][128]; plane 2 (h2) is overwritten with dz1 by bwd2
    const float *heads;     // [plane][NDP_HROW]
    const float *dO;        // [plane][16]
    float *gpart;           // this workgroup's partial [P]
    int n, plane, n_tiles, tile0, tile_step;
    // layer-generic view used by bwd2 (the NDP callers derive it from `act`; the NSFP chain walks its 8 planes):
    float *dz_plane;        // [plane][128] gradient wrt the layer's pre-activation, rewritten in place for the layer below
    const float *h_plane;   // [plane][128] the layer's input activation (post-ReLU)
    int w_off, b_off;       // offsets of the layer's weight / bias inside params and inside the partial
    int from_dO, wh_off, nh; // bwd2: recompute dz from dO through the nh head rows at params + wh_off (else: read dz_plane)
};

// NDP level: which slice of the flat parameter block the two generic backward stages work on
__host__ __device__ inline void bwd_job_ndp_layer2(BwdJob &job, int nh) {
    const ndp_layer_desc dd = {NDP_W, 2, 0, 0, 0, 0.f};
    job.w_off = ndp_off_Wi(&dd, 2); job.b_off = ndp_off_bi(&dd, 2);
    job.from_dO = 1; job.wh_off = ndp_off_Wi(&dd, 3); job.nh = nh;
}

// [64][128] tile: global -> padded LDS tile through registers (NSFP forward layers; the backward uses LDS-DMA)
__device__ __forceinline__ void load_tile_to_lds(const float *src /*[64][128] global*/, float *dst /*LDS [64][LD]*/) {
    const int t = threadIdx.x;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = reinterpret_cast<const float4 *>(src)[t + 256 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = t + 256 * i;           // float4 index 0..2047
        *reinterpret_cast<float4 *>(dst + (idx >> 5) * NDP_LD + 4 * (idx & 31)) = v[i];
    }
}

__device__ __forceinline__ int bp_row(int r) { return BP_PAIR * (r >> 1) + NDP_W * (r & 1); }

// [64][128] tile: global -> LDS image by LDS-DMA, 8 x 1 KiB per wave (one row pair per instruction), asynchronous:
// the caller waits with glds_wait() before the barrier that publishes the tile.
__device__ __forceinline__ void glds_tile(const float *src /*global [64][128]*/, float *dst /*LDS image*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = 8 * wv + i;                                  // row pair
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(src + 2 * NDP_W * q + 4 * lane),
            (__attribute__((address_space(3))) void *)(dst + BP_PAIR * q), 16, 0, 0);
    }
}
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// tile_gemm_64x32 over an LDS-DMA tile image
__device__ __forceinline__ void tile_gemm_64x32_sw(const float *in /*LDS image*/, const float (&w)[64],
                                                   int l31, int h, f32x16 &acc0, f32x16 &acc1) {
    const float *r0 = in + bp_row(l31) + 64 * h;
    const float *r1 = r0 + 16 * BP_PAIR;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float4 a0 = *reinterpret_cast<const float4 *>(r0 + 4 * i);
        const float4 a1 = *reinterpret_cast<const float4 *>(r1 + 4 * i);
        acc0 = MFMA32(w[4 * i], a0.x, acc0);     acc1 = MFMA32(w[4 * i], a1.x, acc1);
        acc0 = MFMA32(w[4 * i + 1], a0.y, acc0); acc1 = MFMA32(w[4 * i + 1], a1.y, acc1);
        acc0 = MFMA32(w[4 * i + 2], a0.z, acc0); acc1 = MFMA32(w[4 * i + 2], a1.z, acc1);
        acc0 = MFMA32(w[4 * i + 3], a0.w, acc0); acc1 = MFMA32(w[4 * i + 3], a1.w, acc1);
        if ((i & 3) == 3) asm volatile("" ::: "memory");
    }
}

// backward epilogue of tile_gemm_64x32_sw: z = d * [hmask > 0] -> zout (same tile coordinates), b128 reads and writes
__device__ __forceinline__ void epilogue_mask_sw(const f32x16 &d0, const f32x16 &d1, const float *hmask /*LDS*/,
                                                 float *zout /*LDS*/, int wv, int l31, int h) {
    const int off = bp_row(l31) + 32 * wv + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 m0 = *reinterpret_cast<const float4 *>(hmask + off + 8 * g);
        const float4 m1 = *reinterpret_cast<const float4 *>(hmask + off + 16 * BP_PAIR + 8 * g);
        *reinterpret_cast<float4 *>(zout + off + 8 * g) =
            make_float4(m0.x > 0.f ? d0[4 * g] : 0.f, m0.y > 0.f ? d0[4 * g + 1] : 0.f,
                        m0.z > 0.f ? d0[4 * g + 2] : 0.f, m0.w > 0.f ? d0[4 * g + 3] : 0.f);
        *reinterpret_cast<float4 *>(zout + off + 16 * BP_PAIR + 8 * g) =
            make_float4(m1.x > 0.f ? d1[4 * g] : 0.f, m1.y > 0.f ? d1[4 * g + 1] : 0.f,
                        m1.z > 0.f ? d1[4 * g + 2] : 0.f, m1.w > 0.f ? d1[4 * g + 3] : 0.f);
    }
}

// [64][128] tile: LDS image -> global, 8 float4 per thread, fully coalesced (thread t: rows (t >> 5) + 8i)
__device__ __forceinline__ void store_tile_from_lds_sw(const float *src /*LDS image*/, float *dst /*global [64][128]*/) {
    const int t = threadIdx.x;
    const float *s0 = src + bp_row(t >> 5) + 4 * (t & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        reinterpret_cast<float4 *>(dst)[t + 256 * i] = *reinterpret_cast<const float4 *>(s0 + 4 * BP_PAIR * i);
}

// dW[mt] += dz^T h   (rows o = 32*mt.., cols k = 32wv + l31), contraction over the tile's 64 points: A = dz[p][l31 + 32m],
// B = h[p][32wv + l31], both conflict-free b32 reads of one row.
// (A variant with the dW rows permuted so that one ds_read_b128 feeds all four A operands, software-pipelined by
//  hand, measured SLOWER: bwd1 0.214 ms against 0.180 ms -- the compiler's own schedule of the b32 reads wins.)
// COLSUM: the A operands are dz[p][l31 + 32m] for the 32 points of this lane's half -- adding them up as they pass gives
// the column sums of dz (the layer's bias gradient) on the idle VALU: cs[m] += dz[32h .. 32h+31][l31 + 32m].
template <bool COLSUM>
__device__ __forceinline__ void tile_outer_128x32_sw(const float *dz /*LDS image*/, const float *hin /*LDS image*/,
                                                     int wv, int l31, int h, f32x16 (&dW)[4], float (&cs)[4]) {
#pragma unroll 2
    for (int ks = 0; ks < 32; ++ks) {
        const int ro = 16 * BP_PAIR * h + bp_row(ks);                // row p = 32h + ks
        const float b = hin[ro + 32 * wv + l31];
        const float *dr = dz + ro + l31;
        const float a0 = dr[0], a1 = dr[32], a2 = dr[64], a3 = dr[96];
        dW[0] = MFMA32(a0, b, dW[0]);
        dW[1] = MFMA32(a1, b, dW[1]);
        dW[2] = MFMA32(a2, b, dW[2]);
        dW[3] = MFMA32(a3, b, dW[3]);
        if (COLSUM) { cs[0] += a0; cs[1] += a1; cs[2] += a2; cs[3] += a3; }
    }
}
// fold the two half-tile partials of tile_outer_128x32_sw<true> (taken from wave 0) -> out[128]   (sc: >= 256 floats of LDS)
__device__ __forceinline__ void outer_colsum_finish(const float (&cs)[4], float *sc, float *out) {
    const int t = threadIdx.x;
    __syncthreads();
    if (t < 64) {
#pragma unroll
        for (int m = 0; m < 4; ++m) sc[(t >> 5) * NDP_W + 32 * m + (t & 31)] = cs[m];
    }
    __syncthreads();
    if (t < NDP_W) out[t] = sc[t] + sc[NDP_W + t];
}

__device__ __forceinline__ void store_dW(float *g, const f32x16 (&dW)[4], int wv, int l31, int h) {
    const int col = 32 * wv + l31;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[(32 * m + mfma_row(r, h)) * NDP_W + col] = dW[m][r];
}

// point of (k-step ks, lane group lk) in the 16x16x4 MFMA stages that contract over a tile's 64 points: rows 4 apart
// are 8 banks apart in the tile image, so the b32 B reads of a 32-lane group collide 2-way at worst
__device__ __forceinline__ int mfma16_point(int ks, int lk) { return 16 * (ks >> 2) + 4 * lk + (ks & 3); }

// hidden layer l: dW_l += dz_l^T h_{l-1} ; db_l ; dh_{l-1} = dz_l W_l ; dz_{l-1} = dh_{l-1} * [h_{l-1} > 0] written over dz_l.
// job.from_dO: l is the layer right below the heads.  Its dz = (dO Wh) * [h > 0] is computed here from dO (K = 16: eight
// k-steps) over the activation tile, in place -- never stored to HBM -- and the head stage of the backward rides along:
// dWh += dO^T h on the 16x16x4 MFMA (before h is overwritten), dbh from the registers that carry dO.
// (Round 1 had a separate head kernel that read the whole h plane a second time.)
__device__ __forceinline__ void bwd2_body(const HeadCfg &hc, const BwdJob &job, float *sm) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int l15 = lane & 15, lk = lane >> 4;
    float *bufA = sm + LB_BUFA, *bufB = sm + LB_BUFB;
    const float *W2 = job.params + job.w_off;
    float w2t[64];
    load_w_bwd(W2, sm + LB_BUFA, wv, l31, h, w2t);
    f32x16 dW2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW2[m][r] = 0.f;
    float gb2[4] = {0.f, 0.f, 0.f, 0.f};            // db of this layer: column sums of dz, taken inside the dW outer product
    float *dOs = sm + LB_DO, *whs = sm + LB_WH;
    // head matrix (the MFMA A operand of dz = dO Wh) staged in LDS once: 8 registers through the GEMM phases were the
    // difference between spilling and not
    if (job.from_dO)
        for (int i = t; i < NDP_WHROWS * NDP_W; i += 256) whs[i] = i < job.nh * NDP_W ? job.params[job.wh_off + i] : 0.f;
    f32x4 gWha, gWhb;                               // dWh[j = 4lk + r][k = 32wv + l15 (a) / + 16 (b)]
#pragma unroll
    for (int r = 0; r < 4; ++r) { gWha[r] = 0.f; gWhb[r] = 0.f; }
    float4 gbh = make_float4(0.f, 0.f, 0.f, 0.f);   // partial sums of dO[.][4(t&3) ..]
    // the input-activation tile (bufA) of tile i+1 is requested as soon as tile i is done with it, under tile i's store
    if (job.tile0 < job.n_tiles) glds_tile(job.h_plane + (size_t)job.tile0 * NDP_TILE * NDP_W, bufA);
    for (int tile = job.tile0; tile < job.n_tiles; tile += job.tile_step) {
        const int base = tile * NDP_TILE;
        float *plane2 = job.dz_plane + (size_t)base * NDP_W;
        PT_DECL;
        glds_tile(plane2, bufB);                                                    // h (becomes dz below), or dz
        if (job.from_dO) {
            const float4 dv = reinterpret_cast<const float4 *>(job.dO + (size_t)base * NDP_NHMAX)[t];
            float *dr = dOs + (t >> 2) * 17 + 4 * (t & 3);
            dr[0] = dv.x; dr[1] = dv.y; dr[2] = dv.z; dr[3] = dv.w;
            gbh.x += dv.x; gbh.y += dv.y; gbh.z += dv.z; gbh.w += dv.w;
            glds_wait();
            __syncthreads();
            {   // dWh += dO^T h over the tile's 64 points (16 k-steps x two 16-column blocks of this wave's slab)
#pragma unroll 4
                for (int ks = 0; ks < 16; ++ks) {
                    const int p = mfma16_point(ks, lk);
                    const float a = dOs[p * 17 + l15];
                    const float *br = bufB + bp_row(p) + 32 * wv + l15;
                    const float b0 = br[0], b1 = br[16];
                    gWha = MFMA16(a, b0, gWha);
                    gWhb = MFMA16(a, b1, gWhb);
                }
            }
            f32x16 z0, z1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { z0[r] = 0.f; z1[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < NDP_WHROWS / 2; ++ks) {                         // head rows j = 2ks + h < 12 (at most 11 exist)
                const float a = whs[(2 * ks + h) * NDP_W + 32 * wv + l31];
                const float b0 = dOs[l31 * 17 + 2 * ks + h], b1 = dOs[(l31 + 32) * 17 + 2 * ks + h];
                z0 = MFMA32(a, b0, z0);
                z1 = MFMA32(a, b1, z1);
            }
            epilogue_mask_sw(z0, z1, bufB, bufB, wv, l31, h);                       // own 32-column slab, in place
        } else {
            glds_wait();
        }
        PT(0);
        __syncthreads();
        PT(1);
        {
            f32x16 d0, d1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
            tile_outer_128x32_sw<true>(bufB, bufA, wv, l31, h, dW2, gb2);
            tile_gemm_64x32_sw(bufB, w2t, l31, h, d0, d1);
            PT(2);
            __syncthreads();                       // every wave is done reading dz2
            PT(3);
            // dz1 goes through LDS so that HBM sees coalesced float4 rows (and the epilogue needs one base
            // address instead of 32 per-element addresses, which used to cost 58 spilled registers)
            epilogue_mask_sw(d0, d1, bufA, bufB, wv, l31, h);
        }
        PT(4);
        __syncthreads();
        PT(5);
        // bufA (the mask of the epilogue above) is dead from here: the next tile's copy starts now, under the store
        if (tile + job.tile_step < job.n_tiles)
            glds_tile(job.h_plane + (size_t)(tile + job.tile_step) * NDP_TILE * NDP_W, bufA);
        store_tile_from_lds_sw(bufB, plane2);
        PT(6);
        __syncthreads();
        PT(7);
    }
    float *G = job.gpart;
    store_dW(G + job.w_off, dW2, wv, l31, h);
    outer_colsum_finish(gb2, sm + LB_BUFA, G + job.b_off);
    if (!job.from_dO) return;
    // ---- head stage results: dWh rows j < nh, dbh
    float *gwh = G + job.wh_off;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = 4 * lk + r;
        if (j < job.nh) {
            gwh[j * NDP_W + 32 * wv + l15] = gWha[r];
            gwh[j * NDP_W + 32 * wv + 16 + l15] = gWhb[r];
        }
    }
    float *sh = sm + LB_BUFB;                       // [256] float4: the dO row partials (the tiles are dead)
    reinterpret_cast<float4 *>(sh)[t] = gbh;
    __syncthreads();
    if (t < job.nh) {
        float s = sh[t];                                                          // thread 4q + (j >> 2), component j & 3
#pragma unroll 8
        for (int q = 1; q < 64; ++q) s += sh[4 * (4 * q + (t >> 2)) + (t & 3)];
        gwh[job.nh * NDP_W + t] = s;
    }
}

// hidden layer 1 and the input layer: dW1 += dz1^T h0 ; db1 ; dh0 = dz1 W1 ; dz0 = dh0 * [h0 > 0] ;
// [dW0 | db0]^T += [pe | 1]^T dz0 (16x16x4 MFMA: rows = the 6 posenc channels and a row of ones, columns = this wave's 32 outputs)
__device__ __forceinline__ void bwd1_body(const HeadCfg &hc, const BwdJob &job, float *sm) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int l15 = lane & 15, lk = lane >> 4;
    float *bufA = sm + LB_BUFA, *bufB = sm + LB_BUFB, *pe = sm + LB_PE;
    const ndp_layer_desc dd = {NDP_W, 2, hc.motion, hc.rotfmt, 0, hc.mlp_scale};
    const float *W1 = job.params + ndp_off_Wi(&dd, 1);
    float w1t[64];
    load_w_bwd(W1, sm + LB_BUFA, wv, l31, h, w1t);
    f32x16 dW1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW1[m][r] = 0.f;
    f32x4 gW0a, gW0b;                              // [dW0 | db0]^T[c][o]: c = 4*lk + r (c = 6: db0), o = 32wv + l15 (a) / + 16 (b)
#pragma unroll
    for (int r = 0; r < 4; ++r) { gW0a[r] = 0.f; gW0b[r] = 0.f; }
    float gb1[4] = {0.f, 0.f, 0.f, 0.f};

    // the h0 tile (bufA) of tile i+1 is requested as soon as tile i is done with it, under tile i's dW0 stage
    if (job.tile0 < job.n_tiles) glds_tile(job.act + (size_t)job.tile0 * NDP_TILE * NDP_W, bufA);
    for (int tile = job.tile0; tile < job.n_tiles; tile += job.tile_step) {
        const int base = tile * NDP_TILE;
        PT_DECL;
        // ---- dz1 tile -> bufB (LDS-DMA), posenc -> pe
        glds_tile(job.act + (2 * (size_t)job.plane + base) * NDP_W, bufB);
        if (t < 64) {
            const float *hr = job.heads + (size_t)(base + t) * NDP_HROW;
            const float4 pa = *reinterpret_cast<const float4 *>(hr + 16);
            const float4 pb = *reinterpret_cast<const float4 *>(hr + 20);
            pe[t] = pa.x; pe[NDP_PES + t] = pa.y; pe[2 * NDP_PES + t] = pa.z; pe[3 * NDP_PES + t] = pa.w;
            pe[4 * NDP_PES + t] = pb.x; pe[5 * NDP_PES + t] = pb.y;
        }
        glds_wait();
        PT(0);
        __syncthreads();
        PT(1);
        // ---- dW1 += dz1^T h0 (+ db1) ; dh0 = dz1 W1
        f32x16 d0, d1;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
            tile_outer_128x32_sw<true>(bufB, bufA, wv, l31, h, dW1, gb1);
            tile_gemm_64x32_sw(bufB, w1t, l31, h, d0, d1);
        }
        PT(2);
        __syncthreads();
        PT(3);
        // ---- dz0 = dh0 * [h0 > 0] -> bufB
        epilogue_mask_sw(d0, d1, bufA, bufB, wv, l31, h);
        PT(4);
        __syncthreads();
        PT(5);
        // bufA (h0: the mask of the epilogue above) is dead from here: the next tile's h0 arrives under the dW0 stage
        if (tile + job.tile_step < job.n_tiles)
            glds_tile(job.act + (size_t)(tile + job.tile_step) * NDP_TILE * NDP_W, bufA);
        // ---- [dW0 | db0]^T += [pe | 1]^T dz0 on the 16x16x4 MFMA: A[c][p] = pe[c][p] (c < 6), 1 (c = 6), B[p][o] = dz0[p][o]
        {
            const float *ap = pe + (l15 < 6 ? l15 : 0) * NDP_PES;
#pragma unroll 4
            for (int ks = 0; ks < 16; ++ks) {
                const int p = mfma16_point(ks, lk);
                float a = ap[p];
                if (l15 >= 6) a = l15 == 6 ? 1.0f : 0.f;
                const float *br = bufB + bp_row(p) + 32 * wv + l15;
                const float b0 = br[0], b1 = br[16];
                gW0a = MFMA16(a, b0, gW0a);
                gW0b = MFMA16(a, b1, gW0b);
            }
        }
        PT(6);
        __syncthreads();
        PT(7);
    }
    float *G = job.gpart;
    store_dW(G + ndp_off_Wi(&dd, 1), dW1, wv, l31, h);
    {   // dW0[o][c] (c < 6) and db0[o] (c = 6): lane holds c = 4*lk + r for its two columns
        float *gw0 = G + ndp_off_W0(&dd), *gb0 = G + ndp_off_b0(&dd);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * lk + r;
            if (c < 6) {
                gw0[(32 * wv + l15) * 6 + c] = gW0a[r];
                gw0[(32 * wv + 16 + l15) * 6 + c] = gW0b[r];
            } else if (c == 6) {
                gb0[32 * wv + l15] = gW0a[r];
                gb0[32 * wv + 16 + l15] = gW0b[r];
            }
        }
    }
    outer_colsum_finish(gb1, sm + LB_BUFA, G + ndp_off_bi(&dd, 1));
}

// ------------------------------------------------------------------------------------------------
// standalone kernels
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256, 2)
k_level_fwd(HeadCfg hc, LevelJob job) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    job.tile0 = blockIdx.x;
    job.tile_step = gridDim.x;
    level_fwd_body(hc, job, sm);
}

extern "C" __global__ void __launch_bounds__(256, 2)
k_level_bwd2(HeadCfg hc, BwdJob job, int p_stride) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    job.tile0 = blockIdx.x;
    job.tile_step = gridDim.x;
    job.gpart += (size_t)blockIdx.x * p_stride;
    bwd2_body(hc, job, sm);
}

extern "C" __global__ void __launch_bounds__(256, 2)
k_level_bwd1(HeadCfg hc, BwdJob job, int p_stride) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    job.tile0 = blockIdx.x;
    job.tile_step = gridDim.x;
    job.gpart += (size_t)blockIdx.x * p_stride;
    bwd1_body(hc, job, sm);
}

// dO[p][16] = mlp_scale * dL/d(scaled head outputs) for p < n, zero rows up to `plane`
extern "C" __global__ void __launch_bounds__(256)
k_head_bwd(HeadCfg hc, const float *x, const float *heads, const float *g, const float *g_nr, int n, int plane, float *dO) {
    __shared__ __attribute__((aligned(16))) float rows[256 * NDP_NHMAX];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= plane) return;
    float *out = dO + (size_t)p * NDP_NHMAX;
    if (p < n) {
        const float xv[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
        const float gv[3] = {g[3 * p], g[3 * p + 1], g[3 * p + 2]};
        point_head_bwd(hc, heads + (size_t)p * NDP_HROW, xv, gv, g_nr ? g_nr[p] : 0.f, rows + threadIdx.x * NDP_NHMAX, out);
    } else {
#pragma unroll
        for (int j = 0; j < NDP_NHMAX; j += 4) *reinterpret_cast<float4 *>(out + j) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

extern "C" __global__ void k_grad_reduce(const float *gpart, int n_part, int p_stride, int P, float *grads) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float s = gpart[i];
    for (int g = 1; g < n_part; ++g) s += gpart[(size_t)g * p_stride + i];
    grads[i] = s;
}

// ---- brute-force 1-NN.  Two queries per thread; references staged in LDS as SoA (x[], y[], z[]) so that
// one ds_read_b128 feeds four references; distances in packed fp32 (v_pk_add/mul/fma: two references per
// instruction, same fma chain and therefore the same bits as the scalar form); the running minimum is
// tracked per 16-reference sub-chunk with v_min3 and the exact (lowest) index is recovered by re-scanning
// the winning sub-chunk.  ~3.7 VALU instructions per distance instead of ~10.
#define NN_STAGE 2048
#ifndef NN_SUB
#define NN_SUB 16
#endif
#define NN_QPB 512                    /* queries per workgroup (standalone operator: two per thread) */
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_dist2(f32x2 X, f32x2 Y, f32x2 Z, f32x2 qx, f32x2 qy, f32x2 qz) {
    const f32x2 dx = qx - X, dy = qy - Y, dz = qz - Z;
    f32x2 dd = dx * dx;
    dd = __builtin_elementwise_fma(dy, dy, dd);
    dd = __builtin_elementwise_fma(dz, dz, dd);
    return dd;
}

// NQ queries per thread (queries qbase + t + 256*w): the reference tile read from LDS is shared by NQ queries
template <int NQ>
__device__ __forceinline__ void nn_body(const float *q, int nq, const float *r, int nr, float *d2, int *idx,
                                        int qbase, float *sm /*[3][NN_STAGE]*/) {
    const int t = threadIdx.x;
    float *xs = sm, *ys = sm + NN_STAGE, *zs = sm + 2 * NN_STAGE;
    float qc[NQ][3];
    f32x2 qx[NQ], qy[NQ], qz[NQ];
    float best[NQ];
    int sc_best[NQ];                                       // winning sub-chunk (global index)
#pragma unroll
    for (int w = 0; w < NQ; ++w) {
        const int i = qbase + 256 * w + t;
        qc[w][0] = qc[w][1] = qc[w][2] = 0.f;
        if (i < nq) { qc[w][0] = q[3 * (size_t)i]; qc[w][1] = q[3 * (size_t)i + 1]; qc[w][2] = q[3 * (size_t)i + 2]; }
        qx[w] = f32x2{qc[w][0], qc[w][0]}; qy[w] = f32x2{qc[w][1], qc[w][1]}; qz[w] = f32x2{qc[w][2], qc[w][2]};
        best[w] = INFINITY;
        sc_best[w] = -1;
    }
    for (int c0 = 0; c0 < nr; c0 += NN_STAGE) {
        const int cn = min(NN_STAGE, nr - c0);
        const int cpad = (cn + NN_SUB - 1) / NN_SUB * NN_SUB;
        __syncthreads();
        {   // stage: all loads first, then the LDS stores (one exposed latency, not eight)
            float v[NN_STAGE / 256][3];
#pragma unroll
            for (int k = 0; k < NN_STAGE / 256; ++k) {
                const int j = t + 256 * k;
                const float nanv = __builtin_nanf("");
                v[k][0] = v[k][1] = v[k][2] = nanv;       // padding never wins a minimum nor an equality
                if (j < cn) {
                    const float *rp = r + 3 * (size_t)(c0 + j);
                    v[k][0] = rp[0]; v[k][1] = rp[1]; v[k][2] = rp[2];
                }
            }
#pragma unroll
            for (int k = 0; k < NN_STAGE / 256; ++k) {
                const int j = t + 256 * k;
                if (j < cpad) { xs[j] = v[k][0]; ys[j] = v[k][1]; zs[j] = v[k][2]; }
            }
        }
        __syncthreads();
        const int nsub = cpad / NN_SUB;
        for (int sc = 0; sc < nsub; ++sc) {
            float m[NQ];
#pragma unroll
            for (int w = 0; w < NQ; ++w) m[w] = INFINITY;
#pragma unroll
            for (int u = 0; u < NN_SUB / 4; ++u) {
                const int o = sc * NN_SUB + 4 * u;
                const float4 X = *reinterpret_cast<const float4 *>(xs + o);
                const float4 Y = *reinterpret_cast<const float4 *>(ys + o);
                const float4 Z = *reinterpret_cast<const float4 *>(zs + o);
                const f32x2 X0 = {X.x, X.y}, X1 = {X.z, X.w}, Y0 = {Y.x, Y.y}, Y1 = {Y.z, Y.w}, Z0 = {Z.x, Z.y}, Z1 = {Z.z, Z.w};
#pragma unroll
                for (int w = 0; w < NQ; ++w) {
                    const f32x2 a0 = pk_dist2(X0, Y0, Z0, qx[w], qy[w], qz[w]), a1 = pk_dist2(X1, Y1, Z1, qx[w], qy[w], qz[w]);
                    m[w] = fminf(fminf(m[w], a0.x), a0.y);
                    m[w] = fminf(fminf(m[w], a1.x), a1.y);
                }
            }
            const int gsc = (c0 / NN_SUB) + sc;
#pragma unroll
            for (int w = 0; w < NQ; ++w)
                if (m[w] < best[w]) { best[w] = m[w]; sc_best[w] = gsc; }
        }
    }
    // exact lowest index inside the winning sub-chunk (same arithmetic -> bitwise equality is safe)
#pragma unroll
    for (int w = 0; w < NQ; ++w) {
        const int i = qbase + 256 * w + t;
        if (i >= nq) continue;
        int bi = -1;
        if (sc_best[w] >= 0) {
            const int j0 = sc_best[w] * NN_SUB, j1 = min(j0 + NN_SUB, nr);
            for (int j = j1 - 1; j >= j0; --j) {
                const float dx = qc[w][0] - r[3 * (size_t)j], dy = qc[w][1] - r[3 * (size_t)j + 1], dz = qc[w][2] - r[3 * (size_t)j + 2];
                const float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (dd == best[w]) bi = j;                // descending j: the last hit is the lowest index
            }
        }
        d2[i] = best[w];
        idx[i] = bi;
    }
}

extern "C" __global__ void __launch_bounds__(256)
k_nn(const float *x, int S, const float *y, int T, float *d2x, int *idx_x, float *d2y, int *idx_y) {
    __shared__ __attribute__((aligned(16))) float sm[3 * NN_STAGE];
    const int bx = (S + NN_QPB - 1) / NN_QPB;
    if ((int)blockIdx.x < bx) nn_body<2>(x, S, y, T, d2x, idx_x, blockIdx.x * NN_QPB, sm);
    else nn_body<2>(y, T, x, S, d2y, idx_y, (blockIdx.x - bx) * NN_QPB, sm);
}

// Latency shape of the exact 1-NN (few pairs resident: one pair must spread over the chip).  A workgroup owns 64
// queries, one per lane; its four waves each scan a quarter of every 2048-reference stage (LDS, broadcast reads, the
// same packed arithmetic and sub-chunk bookkeeping as nn_body), then the four candidates of a query are folded in
// reference order (strict <: the earliest quarter keeps ties).  S/64 + T/64 workgroups per pair instead of S/512 + T/512.
template <int NW = 4>
__device__ __forceinline__ void nn_lat_body(const float *q, int nq, const float *r, int nr, float *d2, int *idx,
                                            int qbase, float *sm /*[3][NN_STAGE] + [NW][64] + [NW][64]*/) {
    constexpr int NT = 64 * NW;                                     // threads of the workgroup: NW waves share the 64 queries
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    float *xs = sm, *ys = sm + NN_STAGE, *zs = sm + 2 * NN_STAGE;
    float *cb = sm + 3 * NN_STAGE;
    int *ci = reinterpret_cast<int *>(cb + NT);
    const int i = qbase + lane;
    // (the query is requested unconditionally at a clamped index, next to the stage's references: behind `if (i < nq)` it was a global
    //  round trip of its own in front of them; a lane beyond nq scans with the last query and writes nothing)
    float qc[3];
    {
        const float *qp = q + 3 * (size_t)min(i, nq - 1);
        qc[0] = qp[0]; qc[1] = qp[1]; qc[2] = qp[2];
    }
    const f32x2 qx = {qc[0], qc[0]}, qy = {qc[1], qc[1]}, qz = {qc[2], qc[2]};
    float best = INFINITY;
    int sc_best = -1;
    for (int c0 = 0; c0 < nr; c0 += NN_STAGE) {
        const int cn = min(NN_STAGE, nr - c0);
        const int cpad = (cn + NN_SUB - 1) / NN_SUB * NN_SUB;
        __syncthreads();
        {
            float v[NN_STAGE / NT][3];
#pragma unroll
            for (int k = 0; k < NN_STAGE / NT; ++k) {
                const int j = t + NT * k;
                const float nanv = __builtin_nanf("");
                v[k][0] = v[k][1] = v[k][2] = nanv;
                if (j < cn) { const float *rp = r + 3 * (size_t)(c0 + j); v[k][0] = rp[0]; v[k][1] = rp[1]; v[k][2] = rp[2]; }
            }
#pragma unroll
            for (int k = 0; k < NN_STAGE / NT; ++k) {
                const int j = t + NT * k;
                if (j < cpad) { xs[j] = v[k][0]; ys[j] = v[k][1]; zs[j] = v[k][2]; }
            }
        }
        __syncthreads();
        const int nsub = cpad / NN_SUB, per = (nsub + NW - 1) / NW;           // sub-chunks of this stage, per wave
        for (int sc = wv * per; sc < min(nsub, (wv + 1) * per); ++sc) {
            float m = INFINITY;
#pragma unroll
            for (int u = 0; u < NN_SUB / 4; ++u) {
                const int o = sc * NN_SUB + 4 * u;
                const float4 X = *reinterpret_cast<const float4 *>(xs + o);
                const float4 Y = *reinterpret_cast<const float4 *>(ys + o);
                const float4 Z = *reinterpret_cast<const float4 *>(zs + o);
                const f32x2 X0 = {X.x, X.y}, X1 = {X.z, X.w}, Y0 = {Y.x, Y.y}, Y1 = {Y.z, Y.w}, Z0 = {Z.x, Z.y}, Z1 = {Z.z, Z.w};
                const f32x2 a0 = pk_dist2(X0, Y0, Z0, qx, qy, qz), a1 = pk_dist2(X1, Y1, Z1, qx, qy, qz);
                m = fminf(fminf(m, a0.x), a0.y);
                m = fminf(fminf(m, a1.x), a1.y);
            }
            if (m < best) { best = m; sc_best = c0 / NN_SUB + sc; }
        }
    }
    // Which reference of the winning sub-chunk: its NN_SUB candidates are requested TOGETHER -- from the LDS stage when there was only
    // one (the references are still there: the same values), from global memory at a clamped index otherwise -- and compared from the
    // last to the first (the lowest index of the minimum stays).  Until round 6 a loop of one global round trip per candidate: sixteen
    // dependent round trips, a third of the batch-1 stage.
    int bi = -1;
    if (sc_best >= 0 && i < nq) {
        const int j0 = sc_best * NN_SUB;
        float rx[NN_SUB], ry[NN_SUB], rz[NN_SUB];
        if (nr <= NN_STAGE) {
#pragma unroll
            for (int u = 0; u < NN_SUB; ++u) { rx[u] = xs[j0 + u]; ry[u] = ys[j0 + u]; rz[u] = zs[j0 + u]; }   // (NaN beyond nr: never equal)
        } else {
#pragma unroll
            for (int u = 0; u < NN_SUB; ++u) {
                const float *rp = r + 3 * (size_t)min(j0 + u, nr - 1);
                rx[u] = rp[0]; ry[u] = rp[1]; rz[u] = rp[2];
            }
        }
#pragma unroll
        for (int u = NN_SUB - 1; u >= 0; --u) {
            const float dx = qc[0] - rx[u], dy = qc[1] - ry[u], dz = qc[2] - rz[u];
            const float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (j0 + u < nr && dd == best) bi = j0 + u;
        }
    }
    // The quarters of one stage are in reference order, but a later stage's quarter w precedes nothing of an earlier
    // stage: a wave's running best is over ITS quarters of all stages, so ties across waves must be broken by index.
    cb[64 * wv + lane] = best;
    ci[64 * wv + lane] = bi;
    __syncthreads();
    if (wv == 0 && i < nq) {
        float b = cb[lane];
        int k = ci[lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float v = cb[64 * w + lane];
            const int kv = ci[64 * w + lane];
            if (v < b || (v == b && kv >= 0 && (k < 0 || kv < k))) { b = v; k = kv; }
        }
        d2[i] = b;
        idx[i] = k;
    }
}
static constexpr int kSmemNnLatBytes = (3 * NN_STAGE + 512) * 4;

// sum_i sqrt(d2_i) [d2_i < trunc], deterministic block reduction (all 256 threads get the value)
__device__ __forceinline__ float l1_sum(const float *d2, int n, float trunc, float *scratch) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = d2[i];
        s += (v >= trunc) ? 0.f : sqrtf(v);
    }
    return block_sum_256(s, scratch);
}
__device__ __forceinline__ float sq_sum(const float *x, const float *tt, int K, float *scratch) {
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float e0 = x[3 * k] - tt[3 * k], e1 = x[3 * k + 1] - tt[3 * k + 1], e2 = x[3 * k + 2] - tt[3 * k + 2];
        s += fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
    }
    return block_sum_256(s, scratch);
}

extern "C" __global__ void __launch_bounds__(256)
k_chamfer_bwd(const float *x, int S, const float *y, int T, float trunc, const float *d2x, const int *idx_x,
              const float *d2y, const int *idx_y, float *loss, float *gx, int point_sum) {
    __shared__ float scratch[256];
    // point_reduction (loss.py:233-235): "mean" divides each direction's sum by its point count, "sum" does not
    const float Sdiv = point_sum ? 1.0f : (float)S, Tdiv = point_sum ? 1.0f : (float)T;
    if (blockIdx.x == 0) {
        const float sx = l1_sum(d2x, S, trunc, scratch);
        const float sy = l1_sum(d2y, T, trunc, scratch);
        if (threadIdx.x == 0) loss[0] = sx / Sdiv + sy / Tdiv;
    }
    if (!gx) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const float xi[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    float g[3] = {0.f, 0.f, 0.f};
    if (!(d2x[i] >= trunc)) {
        const float *yy = y + 3 * idx_x[i];
        const float inv = 1.0f / (Sdiv * sqrtf(d2x[i]));
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] = (xi[a] - yy[a]) * inv;
    }
    for (int j = 0; j < T; ++j) {           // ascending j: same order as the oracle / the CPU reference
        if (idx_y[j] == i && !(d2y[j] >= trunc)) {
            const float inv = 1.0f / (Tdiv * sqrtf(d2y[j]));
#pragma unroll
            for (int a = 0; a < 3; ++a) g[a] = fmaf(xi[a] - y[3 * j + a], inv, g[a]);
        }
    }
    gx[3 * i] = g[0]; gx[3 * i + 1] = g[1]; gx[3 * i + 2] = g[2];
}

extern "C" __global__ void __launch_bounds__(256)
k_landmark(const float *x, const float *tt, int K, float *loss, float *gx) {
    __shared__ float scratch[256];
    const float invK = 1.0f / (float)K;
    if (blockIdx.x == 0) {
        const float s = sq_sum(x, tt, K, scratch);
        if (threadIdx.x == 0) loss[0] = s * invK;
    }
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < K && gx) {
#pragma unroll
        for (int a = 0; a < 3; ++a) gx[3 * k + a] = 2.0f * (x[3 * k + a] - tt[3 * k + a]) * invK;
    }
}

// torch.optim.Adam single-tensor update, op for op (see oracle ndp_o_adam)
__device__ __forceinline__ void adam_update(float &p, float g, float &m, float &v, float w1, float b2, float w2,
                                            float neg_step, float bc2s, float eps) {
    const float mi = m + w1 * (g - m);
    float vi = v * b2;
    vi = vi + (w2 * g) * g;
    const float denom = sqrtf(vi) / bc2s + eps;
    p = p + (neg_step * mi) / denom;
    m = mi;
    v = vi;
}

extern "C" __global__ void k_adam(float *p, const float *g, float *m, float *v, int P, float w1, float b2, float w2,
                                  float neg_step, float bc2s, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float pi = p[i], mi = m[i], vi = v[i];
    adam_update(pi, g[i], mi, vi, w1, b2, w2, neg_step, bc2s, eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
}

// ------------------------------------------------------------------------------------------------
// batched engine kernels: blockIdx.y = pair
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float level_freq(int level, int k0) { return ldexpf(1.0f, level + 1 + k0); }

extern "C" __global__ void __launch_bounds__(256, 2)
k_eng_fwd(ndp_engine e, int parity) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y;
    const ndp_pair_state st = e.state[parity * e.B + b];
    if (st.level >= e.m) return;
    if (e.gmax && blockIdx.x == 0 && threadIdx.x == 0) e.gmax[b] = 0;   // this tick's max |dO| starts from zero (k_eng_loss raises it)
    const ndp_pair_geom gm = e.geom[b];
    LevelJob job;
    job.n = gm.K + gm.S;
    job.n_tiles = (job.n + NDP_TILE - 1) / NDP_TILE;
    if ((int)blockIdx.x >= job.n_tiles) return;
    const HeadCfg hc = make_head_cfg(desc_at_level(e.desc, st.level));
    job.params = e.params + ((size_t)b * e.m + st.level) * e.p_stride;
    job.freq = level_freq(st.level, e.k0);
    job.nonrig = nullptr;
    float *pts = e.pts + (size_t)b * 2 * e.n_cap * 3;
    job.x_in = pts + (size_t)st.cur * e.n_cap * 3;
    job.x_out = pts + (size_t)(st.cur ^ 1) * e.n_cap * 3;
    job.act = e.act + (size_t)b * 3 * e.n_cap * NDP_W;
    job.heads = e.heads + (size_t)b * e.n_cap * NDP_HROW;
    job.plane = e.n_cap;
    job.tile0 = blockIdx.x;
    job.tile_step = gridDim.x;
    PT_INIT;
    level_fwd_body(hc, job, sm);
    PT_FLUSH(12);
}

// XCD-aware placement of a (nvb, B) grid whose nvb workgroups per pair share that pair's data: hardware block L = y nvb + x runs on
// XCD L % 8 (observed dispatch order, MI355X_MICROARCH.md: used for speed only -- any placement computes the same thing), so the
// workgroups of one pair are taken from blocks that are congruent mod 8: they then share ONE XCD's L2 instead of pulling the pair's
// targets, indices and partials into eight of them.  (Pairs beyond the last full group of eight keep the plain order.)
__device__ __forceinline__ void xcd_pair_block(int nvb, int B, int &b, int &vb) {
    const int L = blockIdx.y * nvb + blockIdx.x, nfull = B & ~7;
    if (L < nfull * nvb) {
        const int slot = L >> 3;
        b = (slot / nvb) * 8 + (L & 7);
        vb = slot % nvb;
    } else {
        const int r = L - nfull * nvb;
        b = nfull + r / nvb;
        vb = r % nvb;
    }
}
// Per-thread head rows in LDS (run-time row offsets live there) are NDP_LROW = 20 floats apart, not 16: the 16-byte accesses of an
// eight-lane group then hit eight different bank quads and the scalar ones 4-way instead of 16-way ((16 t) mod 32 has two values,
// (20 t) mod 32 eight) -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of k_eng_loss 0.68 before.
#define NDP_LROW 20
#include "ndp_fwd_split.inc"

// ------------------------------------------------------------------------------------------------
// One-pass exact 1-NN for the engine: every squared distance d2(x_i, y_j) is evaluated ONCE and serves both
// directions (loss.py:177-178 calls knn_points twice; SURVEY 8(d) counts 8 S T FLOP for one pass).
//   workgroup = ALL sources x NN1_YCH consecutive targets (SoA in LDS, broadcast ds_read_b128, packed fp32).  The
//   sources are walked in rounds of 512: a thread keeps two of them in registers, wave w of round r owns the 128-source
//   block 4r + w.
//   ROW minimum (nearest target of a source): thread-private, tracked per 16-target sub-chunk, the winning sub-chunk
//   re-scanned exactly at the end of the round; one partial {d2, idx} per (source, target chunk) goes to HBM and is
//   folded by whoever reads it (first chunk wins ties = lowest index).
//   COLUMN minimum (nearest source of a target): over the 128 sources of a wave it is a cross-lane reduction -- a
//   transposed butterfly over the 16 column registers of a sub-chunk (v_permlane32_swap, v_permlane16_swap, DPP
//   row_mirror / row_half_mirror / quad_perm: 35 instructions per 16 targets x 128 sources) -- into an LDS table
//   [block][target]; when all rounds are done the workgroup folds the blocks in order (first block wins ties) and
//   re-scans the winning block for the exact lowest index with the same arithmetic, from a copy of the sources in LDS.
//   d2 and idx are bit-identical to the two-pass brute force (k_nn); nothing but the row partials needs a second look.
// ------------------------------------------------------------------------------------------------
#define NN1_XW 128                    /* sources per wave and round: granularity of the column table */
#define NN1_XB (4 * NN1_XW)           /* sources per round */
#define NN1_YCH 256                   /* targets per workgroup (512: 0.138 ms, 128: 0.137 + a slower row fold, 1024: 0.197) */
#define NN1_XLD (NN1_XW + 1)          /* LDS stride of a 128-source block (the re-scan reads different blocks per lane) */

// v_min_f32 / v_min3_f32 without the canonicalising v_max the compiler puts in front of every fminf operand it cannot
// prove quiet (30 of them per 16-target sub-chunk): the instruction itself returns the other operand for a quiet NaN,
// which is all the NaN padding needs
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float dpp_row_mirror(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_half_mirror(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}

// c[16]: per-lane values of 16 columns -> minimum over the 64 lanes of every column; lane L returns column L >> 2
__device__ __forceinline__ float wave_colmin16(const float (&c)[16], int lane) {
    float d[8], e[4], f[2];
#pragma unroll
    for (int r = 0; r < 8; ++r) {                  // lanes L and L ^ 32: lower half keeps column r, upper half column r + 8
        const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[r]), __float_as_uint(c[r + 8]), false, false);
        d[r] = vmin(__uint_as_float(s[0]), __uint_as_float(s[1]));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                  // rows of 16 lanes: even rows keep column r, odd rows column r + 4
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(d[r]), __float_as_uint(d[r + 4]), false, false);
        e[r] = vmin(__uint_as_float(s[0]), __uint_as_float(s[1]));
    }
    const bool b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int r = 0; r < 2; ++r) {                  // lane l and 15 - l of a row: bit 3 clear keeps r, set keeps r + 2
        const float keep = b3 ? e[r + 2] : e[r], give = b3 ? e[r] : e[r + 2];
        f[r] = vmin(keep, dpp_row_mirror(give));
    }
    const float keep = b2 ? f[1] : f[0], give = b2 ? f[0] : f[1];
    float g = vmin(keep, dpp_row_half_mirror(give));           // lane l and 7 - l of a half row
    g = vmin(g, dpp_quad<0xB1>(g));                            // the four lanes of a quad hold the same column:
    g = vmin(g, dpp_quad<0x4E>(g));                            // fold them (quad_perm [1,0,3,2] then [2,3,0,1])
    return g;
}

struct NnPart { float d2; int idx; };

__host__ __device__ inline int nn1_row_chunks(int t_cap) { return (t_cap + NN1_YCH - 1) / NN1_YCH; }
__host__ __device__ inline int nn1_col_blocks(int n_cap) { return (n_cap + NN1_XW - 1) / NN1_XW; }
// dynamic LDS of the one-pass kernel (floats): target chunk, column table, re-scan results, (optionally) the sources
__host__ __device__ inline int nn1_lds_floats(int n_cap, bool stage_x) {
    const int nb = nn1_col_blocks(n_cap);
    return 3 * NN1_YCH + nb * NN1_YCH + (stage_x ? 3 * nb * NN1_XLD : 0);
}
__host__ __device__ inline bool nn1_stage_x(int n_cap) { return nn1_lds_floats(n_cap, true) * 4 <= 80 * 1024; }

// nearest target of NS sources (i[0..NS-1]; i < 0: skipped) from the row partials of the live target chunks (strict <: the
// first chunk keeps ties).  The partials of up to 8 chunks x NS sources are requested together: with one load in flight per
// thread the fold of S = 8192 x 24 chunks by the loss workgroup was a 0.2 ms latency chain.  cstep: the partials are indexed by
// 256-target chunk; a producer whose workgroups cover 512 targets (k_eng_nn_mx8) writes every SECOND slot only -- cstep = 2.
template <int NS>
__device__ __forceinline__ void nn_row_fold_n(const NnPart *rowpart /*[chunks][n_cap]*/, int n_cap, int T, const int (&i)[NS],
                                              NnPart (&r)[NS], int cstep = 1) {
#pragma unroll
    for (int s = 0; s < NS; ++s) { r[s].d2 = INFINITY; r[s].idx = -1; }
    if (!rowpart) return;
    const int live = ((T + NN1_YCH - 1) / NN1_YCH + cstep - 1) / cstep;
    n_cap *= cstep;                                                  // (slot c of the producer is chunk c * cstep)
    for (int c0 = 0; c0 < live; c0 += 8) {
        NnPart q[NS][8];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                q[s][k].d2 = INFINITY; q[s][k].idx = -1;
                if (c0 + k < live && i[s] >= 0) q[s][k] = rowpart[(size_t)(c0 + k) * n_cap + i[s]];
            }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (q[s][k].d2 < r[s].d2) r[s] = q[s][k];
    }
}
__device__ __forceinline__ NnPart nn_row_fold(const NnPart *rowpart, int n_cap, int T, int i, int cstep = 1) {
    const int ii[1] = {i};
    NnPart r[1];
    nn_row_fold_n<1>(rowpart, n_cap, T, ii, r, cstep);
    return r[0];
}

// sources [S][3] at xs, targets [T][3] at ys; this workgroup: all sources x targets y0 .. y0 + NN1_YCH - 1.
// rowpart: [n_cap] partials of THIS target chunk; d2y / idx_y: final results for the chunk's targets (idx_y = -1 for
// y0 + j in [T, t_out)).
template <bool STAGE_X>
__device__ __forceinline__ void nn1_body(const float *xs, int S, const float *ys, int T, int y0, int t_out,
                                         NnPart *rowpart, float *d2y, int *idx_y, float *sm) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nb = (S + NN1_XW - 1) / NN1_XW;      // live source blocks
    float *lx = sm, *ly = sm + NN1_YCH, *lz = sm + 2 * NN1_YCH;
    float *colp = sm + 3 * NN1_YCH;                // [nb][NN1_YCH]
    float *sx = colp + nb * NN1_YCH;               // STAGE_X: [3][nb * NN1_XLD] sources, SoA per block
    const float nanv = __builtin_nanf("");
    const int cn = min(NN1_YCH, T - y0);           // >= 1
    const int cpad = (cn + 15) & ~15;
    for (int j = t; j < NN1_YCH; j += 256) {       // stage the target chunk (NaN padding: never wins a minimum nor an equality)
        float v0 = nanv, v1 = nanv, v2 = nanv;
        if (j < cn) { const float *rp = ys + 3 * (size_t)(y0 + j); v0 = rp[0]; v1 = rp[1]; v2 = rp[2]; }
        lx[j] = v0; ly[j] = v1; lz[j] = v2;
    }
    if (STAGE_X) {
        const int sn = nb * NN1_XLD;
        for (int i = t; i < nb * NN1_XW; i += 256) {
            float v0 = nanv, v1 = nanv, v2 = nanv;
            if (i < S) { v0 = xs[3 * (size_t)i]; v1 = xs[3 * (size_t)i + 1]; v2 = xs[3 * (size_t)i + 2]; }
            const int o = (i >> 7) * NN1_XLD + (i & 127);
            sx[o] = v0; sx[sn + o] = v1; sx[2 * sn + o] = v2;
        }
    }
    __syncthreads();
    for (int xw = wv; xw < nb; xw += 4) {          // this wave's source blocks; no barrier inside
        float qc[2][3];
        f32x2 qx[2], qy[2], qz[2];
        float best[2];
        int sc_best[2];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int i = xw * NN1_XW + 64 * w + lane;
            qc[w][0] = qc[w][1] = qc[w][2] = nanv;                 // a missing source never wins a minimum
            if (i < S) { qc[w][0] = xs[3 * (size_t)i]; qc[w][1] = xs[3 * (size_t)i + 1]; qc[w][2] = xs[3 * (size_t)i + 2]; }
            qx[w] = f32x2{qc[w][0], qc[w][0]}; qy[w] = f32x2{qc[w][1], qc[w][1]}; qz[w] = f32x2{qc[w][2], qc[w][2]};
            best[w] = INFINITY;
            sc_best[w] = -1;
        }
        float *cp = colp + xw * NN1_YCH + (lane >> 2);
        // (reading the NEXT sub-chunk's 12 broadcast ds_read_b128 ahead of the arithmetic measured slower: 0.141 vs 0.133 ms,
        //  132 registers instead of 70)
        for (int sc = 0; sc < cpad / 16; ++sc) {
            float c[16], m[2] = {INFINITY, INFINITY};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int o = 16 * sc + 4 * u;
                const float4 X = *reinterpret_cast<const float4 *>(lx + o);
                const float4 Y = *reinterpret_cast<const float4 *>(ly + o);
                const float4 Z = *reinterpret_cast<const float4 *>(lz + o);
                const f32x2 X0 = {X.x, X.y}, X1 = {X.z, X.w}, Y0 = {Y.x, Y.y}, Y1 = {Y.z, Y.w}, Z0 = {Z.x, Z.y}, Z1 = {Z.z, Z.w};
                const f32x2 a0 = pk_dist2(X0, Y0, Z0, qx[0], qy[0], qz[0]), a1 = pk_dist2(X1, Y1, Z1, qx[0], qy[0], qz[0]);
                const f32x2 b0 = pk_dist2(X0, Y0, Z0, qx[1], qy[1], qz[1]), b1 = pk_dist2(X1, Y1, Z1, qx[1], qy[1], qz[1]);
                m[0] = vmin3(m[0], a0.x, a0.y); m[0] = vmin3(m[0], a1.x, a1.y);
                m[1] = vmin3(m[1], b0.x, b0.y); m[1] = vmin3(m[1], b1.x, b1.y);
                c[4 * u] = vmin(a0.x, b0.x); c[4 * u + 1] = vmin(a0.y, b0.y);
                c[4 * u + 2] = vmin(a1.x, b1.x); c[4 * u + 3] = vmin(a1.y, b1.y);
            }
#pragma unroll
            for (int w = 0; w < 2; ++w)
                if (m[w] < best[w]) { best[w] = m[w]; sc_best[w] = sc; }
            const float g = wave_colmin16(c, lane);
            if ((lane & 3) == 0) cp[16 * sc] = g;
        }
        // exact lowest index inside the winning sub-chunk (same arithmetic -> bitwise equality is safe)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int i = xw * NN1_XW + 64 * w + lane;
            if (i >= S) continue;
            int bi = -1;
            if (sc_best[w] >= 0) {
                const int j0 = 16 * sc_best[w];
                for (int j = j0 + 15; j >= j0; --j) {
                    const float dx = qc[w][0] - lx[j], dy = qc[w][1] - ly[j], dz = qc[w][2] - lz[j];
                    const float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if (dd == best[w]) bi = y0 + j;               // descending j: the last hit is the lowest index
                }
            }
            NnPart r; r.d2 = best[w]; r.idx = bi;
            rowpart[i] = r;
        }
    }
    __syncthreads();
    // ---- columns: fold the blocks in order (the first block keeps ties), then the exact lowest index inside the winning
    //      block, one thread per target, candidates from the LDS copy of the sources
    for (int jj = t; jj < NN1_YCH; jj += 256) {
        if (y0 + jj >= t_out) break;
        if (jj >= cn) { idx_y[y0 + jj] = -1; continue; }
        float cbest = INFINITY;
        int blk = -1;
        for (int k = 0; k < nb; ++k) {
            const float v = colp[k * NN1_YCH + jj];
            if (v < cbest) { cbest = v; blk = k; }
        }
        int r = -1;
        if (blk >= 0) {
            const float q0 = lx[jj], q1 = ly[jj], q2 = lz[jj];
            const int k0 = blk * NN1_XW;
            if (STAGE_X) {
                const int sn = nb * NN1_XLD;
                const float *bx = sx + blk * NN1_XLD;
                for (int k = NN1_XW - 1; k >= 0; --k) {               // padding beyond S is NaN: never equal
                    const float dx = bx[k] - q0, dy = bx[sn + k] - q1, dz = bx[2 * sn + k] - q2;
                    const float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if (dd == cbest) r = k0 + k;                      // descending k: the last hit is the lowest index
                }
            } else {
                for (int k = min(k0 + NN1_XW, S) - 1; k >= k0; --k) {
                    const float dx = xs[3 * (size_t)k] - q0, dy = xs[3 * (size_t)k + 1] - q1, dz = xs[3 * (size_t)k + 2] - q2;
                    const float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if (dd == cbest) r = k;
                }
            }
        }
        d2y[y0 + jj] = cbest;
        idx_y[y0 + jj] = r;
    }
}

// few pairs resident: blockIdx.x < ceil(n_cap/64): 64 source samples -> targets;  else 64 targets -> source samples.
// Writes the final d2x / idx_x / d2y / idx_y (no partials): e.nn_mode = 1 tells the loss kernel to read them.
template <int NW>
__device__ __forceinline__ void eng_nn_lat_stage(const ndp_engine &e, int parity, float *sm) {
    const int b = blockIdx.y;
    // (level, buffer parity and geometry requested side by side, ONE test: see eng_nn_mx_body)
    const ndp_pair_state *stp = e.state + (size_t)parity * e.B + b;
    struct { int level, cur; } st;
    st.level = stp->level; st.cur = stp->cur;
    const ndp_pair_geom gm = e.geom[b];
    if ((st.level >= e.m) | (gm.S == 0) | (st.cur < 0) | (e.w_cd == 0.f)) return;
    const float *xw = e.pts + ((size_t)b * 2 + (st.cur ^ 1)) * e.n_cap * 3 + 3 * gm.K;
    const float *y = e.tgt + (size_t)b * e.t_cap * 3;
    const int bx = e.n_cap / 64;
    if ((int)blockIdx.x < bx) {
        const int qb = blockIdx.x * 64;
        if (qb >= gm.S) return;
        nn_lat_body<NW>(xw, gm.S, y, gm.T, e.d2x + (size_t)b * e.n_cap, e.idx_x + (size_t)b * e.n_cap, qb, sm);
    } else {
        const int qb = (blockIdx.x - bx) * 64;
        int *iy = e.idx_y + (size_t)b * e.t_cap;
        if (qb < gm.T) nn_lat_body<NW>(y, gm.T, xw, gm.S, e.d2y + (size_t)b * e.t_cap, iy, qb, sm);
        if (threadIdx.x < 64 && qb + (int)threadIdx.x >= gm.T && qb + (int)threadIdx.x < e.t_cap) iy[qb + threadIdx.x] = -1;
    }
}
extern "C" __global__ void __launch_bounds__(256)
k_eng_nn_lat(ndp_engine e, int parity) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    eng_nn_lat_stage<4>(e, parity, sm);
}
// the same with eight waves per 64 queries (each scans an eighth of every stage): what the engine launches (round 4) -- at batch 1
// the stage is one workgroup's latency, 10.7 -> us (the fold over the waves keeps the lowest index: same results)
extern "C" __global__ void __launch_bounds__(512)
k_eng_nn_lat8(ndp_engine e, int parity) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    eng_nn_lat_stage<8>(e, parity, sm);
}
// sixteen waves per 64 queries (each scans a sixteenth of every stage): the engine's launch since round 6 when the pair count is small
// enough for the stage to be ONE workgroup's latency (B <= 2: the scan of a stage is half as long; same fold, same results)
extern "C" __global__ void __launch_bounds__(1024)
k_eng_nn_lat16(ndp_engine e, int parity) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    eng_nn_lat_stage<16>(e, parity, sm);
}

extern "C" __global__ void __launch_bounds__(256)
k_eng_nn(ndp_engine e, int parity, int stage_x) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y;
    const ndp_pair_state st = e.state[parity * e.B + b];
    if (st.level >= e.m) return;
    const ndp_pair_geom gm = e.geom[b];
    if (gm.S == 0 || e.w_cd == 0.f) return;
    const int y0 = blockIdx.x * NN1_YCH;
    int *iy = e.idx_y + (size_t)b * e.t_cap;
    if (y0 >= gm.T) {                               // keep the -1 padding beyond T
        for (int j = y0 + threadIdx.x; j < min(y0 + NN1_YCH, e.t_cap); j += 256) iy[j] = -1;
        return;
    }
    const float *xw = e.pts + ((size_t)b * 2 + (st.cur ^ 1)) * e.n_cap * 3 + 3 * gm.K;
    const float *y = e.tgt + (size_t)b * e.t_cap * 3;
    NnPart *rowpart = reinterpret_cast<NnPart *>(e.nn_row) + ((size_t)b * nn1_row_chunks(e.t_cap) + blockIdx.x) * e.n_cap;
    if (stage_x) nn1_body<true>(xw, gm.S, y, gm.T, y0, e.t_cap, rowpart, e.d2y + (size_t)b * e.t_cap, iy, sm);
    else nn1_body<false>(xw, gm.S, y, gm.T, y0, e.t_cap, rowpart, e.d2y + (size_t)b * e.t_cap, iy, sm);
}

// the same kernel as a standalone operator on one pair, plus the fold of its row partials
extern "C" __global__ void __launch_bounds__(256)
k_nn1(const float *x, int S, const float *y, int T, int n_cap, float *ws_row, float *d2y, int *idx_y, int stage_x) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int y0 = blockIdx.x * NN1_YCH;
    if (y0 >= T) return;
    NnPart *rowpart = reinterpret_cast<NnPart *>(ws_row) + (size_t)blockIdx.x * n_cap;
    if (stage_x) nn1_body<true>(x, S, y, T, y0, T, rowpart, d2y, idx_y, sm);
    else nn1_body<false>(x, S, y, T, y0, T, rowpart, d2y, idx_y, sm);
}
extern "C" __global__ void __launch_bounds__(256)
k_nn1_rows(int S, int T, int n_cap, const float *ws_row, float *d2x, int *idx_x) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const NnPart r = nn_row_fold(reinterpret_cast<const NnPart *>(ws_row), n_cap, T, i);
    d2x[i] = r.d2;
    idx_x[i] = r.idx;
}

#include "ndp_nn_matrix.inc"
// Does the engine's nearest-neighbour stage run as k_eng_nn_mx8 (one workgroup and ONE row partial per 512 targets)?  The launcher and
// the loss stage's fold of the row partials ask the same question.
__host__ __device__ inline bool eng_nn_mx8(const ndp_engine &e) {
    return e.w_cd != 0.f && e.t_cap > 0 && e.nn_mode == 2 && nn2_lds_floats(e.n_cap, 8) * 4 <= 160 * 1024 && !(e.gemm_mode & 128);
}

// Loss, early-stop decision and dL/dx' for every pair (one launch per tick).
//   last workgroup of a pair: loss (registration.py:193-212; loss.py:185-258), the stop rule in double
//                     (registration.py:226-232) and the pair's next state;
//   the others      : the gradient of the loss wrt their 256 warped points -- own nearest-neighbour
//                     term, then the targets whose nearest source point it is, in ascending target
//                     index (the order a sequential CPU scatter-add produces), no atomics.
#define LG_CHUNK 2048
struct LossSmem {
    float red[256];
    int cnt[256], start[256];                                             // per-point bucket sizes / offsets
    int order[LG_CHUNK];                                                  // targets grouped by their nearest source point
    __attribute__((aligned(16))) float rows[256 * NDP_LROW];              // per-thread head rows
};
// block reductions over the 256 ACTIVE threads of a workgroup (t: their index; the others only keep the barriers company -- the
// persistent small-batch tick runs this stage on the lower half of its 512-thread workgroups)
__device__ __forceinline__ float block_sum_256_t(float v, float *scratch, int t, bool act) {
    if (act) scratch[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s >= 64; s >>= 1) {
        if (act && t < s) scratch[t] = scratch[t] + scratch[t + s];
        __syncthreads();
    }
    // the tree's last six levels live in one wave: lane shuffles instead of LDS + a barrier per level (lane t < s adds the value of lane
    // t + s exactly as scratch[t] + scratch[t + s] did: the same association, the same bits)
    if (act && t < 64) {
        float x = scratch[t];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) x = x + __shfl_down(x, s);
        if (t == 0) scratch[0] = x;
    }
    __syncthreads();
    const float r = scratch[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float l1_sum_t(const float *d2, int n, float trunc, float *scratch, int t, bool act) {
    float s = 0.f;
    for (int i0 = act ? t : n; i0 < n; i0 += 8 * 256) {             // eight values requested together, added in index order (one global round
        float v[8];                                                 // trip per 2048 entries instead of eight: round 6)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + 256 * u < n ? d2[i0 + 256 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 256 * u < n) s += (v[u] >= trunc) ? 0.f : sqrtf(v[u]);
    }
    return block_sum_256_t(s, scratch, t, act);
}
__device__ __forceinline__ float sq_sum_t(const float *x, const float *tt, int K, float *scratch, int t, bool act) {
    float s = 0.f;
    for (int k = act ? t : K; k < K; k += 256) {
        const float e0 = x[3 * k] - tt[3 * k], e1 = x[3 * k + 1] - tt[3 * k + 1], e2 = x[3 * k + 2] - tt[3 * k + 2];
        s += fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
    }
    return block_sum_256_t(s, scratch, t, act);
}
// One virtual 256-thread block of the loss stage: vb < nvb - 1 the gradient of 256 warped points, vb == nvb - 1 loss + decision.
// t: index among the block's 256 active threads; act = false: a thread that only takes part in the barriers.
__device__ __forceinline__ void eng_loss_body(const ndp_engine &e, int parity, int b, int vb, int nvb, int t, bool act, LossSmem &sm_) {
    float *red = sm_.red, *rows = sm_.rows;
    int *cnt = sm_.cnt, *start = sm_.start, *order = sm_.order;
    PT_INIT;
    PT_DECL;
    // Only the scalar fields are read here; the per-level array travels memory to memory in the one thread that writes the next state
    // (a by-value copy of the struct parked 80 bytes in scratch in EVERY thread of the launch, behind a wait for its loads: round 6).
    const ndp_pair_state *stp = e.state + (size_t)parity * e.B + b;
    struct { int level, iter, break_counter, adam_t, cur, total_steps, total_evals; double loss_prev; } st;
    st.level = stp->level; st.iter = stp->iter; st.break_counter = stp->break_counter; st.adam_t = stp->adam_t; st.cur = stp->cur;
    st.total_steps = stp->total_steps; st.total_evals = stp->total_evals; st.loss_prev = stp->loss_prev;
    const ndp_pair_geom gm = e.geom[b];                                  // (requested next to the state, not behind the test on it)
    ndp_pair_state *nst = e.state + (size_t)(parity ^ 1) * e.B + b;
    if (st.level >= e.m) {
        if (vb == nvb - 1 && t == 0 && act) { *nst = *stp; nst->decision = NDP_DEC_IDLE; }
        return;
    }
    const int n = gm.K + gm.S;
    const float *x_out = e.pts + ((size_t)b * 2 + (st.cur ^ 1)) * e.n_cap * 3;
    const float *ldmk_t = e.ldmk_t + (size_t)b * e.n_cap * 3;
    const float *tgt = e.tgt + (size_t)b * e.t_cap * 3;
    const float *d2y = e.d2y + (size_t)b * e.t_cap;
    const int *idx_y = e.idx_y + (size_t)b * e.t_cap;
    // nearest target of a source: folded here from the one-pass kernel's per-chunk partials (nn_row_fold)
    const NnPart *rowpart = reinterpret_cast<const NnPart *>(e.nn_row) + (size_t)b * nn1_row_chunks(e.t_cap) * e.n_cap;
    const bool rows_final = e.nn_mode == 1;          // latency shape: d2x / idx_x already hold the answer
    const int rows_cstep = eng_nn_mx8(e) ? 2 : 1;    // the 8-wave matrix-pipe kernel leaves one partial per 512 targets
    const bool use_cd = gm.S > 0 && e.w_cd != 0.f;
    const HeadCfg hcl = make_head_cfg(desc_at_level(e.desc, st.level));
    const bool use_reg = e.w_reg > 0.f && hcl.nonrig;
    const float *hrec = e.heads + (size_t)b * e.n_cap * NDP_HROW;

    if (vb == nvb - 1) {                 // the extra workgroup of the pair: loss + decision, concurrently with the gradient workgroups
        float loss = 0.f;
        PT(0);
        if (gm.K > 0) loss = sq_sum_t(x_out, ldmk_t, gm.K, red, t, act) * (1.0f / (float)gm.K);
        if (use_cd) {
            float sx = 0.f;
            if (rows_final) {
                for (int i = act ? t : gm.S; i < gm.S; i += 256) {
                    const float v = e.d2x[(size_t)b * e.n_cap + i];
                    sx += (v >= e.trunc) ? 0.f : sqrtf(v);
                }
            } else {
                for (int i0 = act ? t : gm.S; i0 < gm.S; i0 += 2 * 256) {                 // same per-thread order as one source at a time
                    int ii[2];
                    NnPart r[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) ii[s] = i0 + 256 * s < gm.S ? i0 + 256 * s : -1;
                    nn_row_fold_n<2>(rowpart, e.n_cap, gm.T, ii, r, rows_cstep);
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (ii[s] >= 0) sx += (r[s].d2 >= e.trunc) ? 0.f : sqrtf(r[s].d2);
                }
            }
            PT(1);
            sx = block_sum_256_t(sx, red, t, act);
            PT(2);
            const float sy = l1_sum_t(d2y, gm.T, e.trunc, red, t, act);
            PT(3);
            const float lcd = sx / (float)gm.S + sy / (float)gm.T;
            loss = gm.K > 0 ? loss + e.w_cd * lcd : lcd;
        }
        if (use_reg) {                                   // registration.py:216-220: + w_reg * BCELoss(nonrigidity, 0)
            float acc = 0.f;
            for (int i = act ? t : n; i < n; i += 256) {
                const float nr = 1.0f / (1.0f + expf(-hrec[(size_t)i * NDP_HROW + hcl.row_nr]));
                float l1 = logf(1.0f - nr);
                if (l1 < -100.0f) l1 = -100.0f;
                acc += -l1;
            }
            acc = block_sum_256_t(acc, red, t, act);
            loss = loss + e.w_reg * (acc * (1.0f / (float)n));
        }
        if (t == 0 && act) {
            int bc = st.break_counter;
            double lp = st.loss_prev;
            bool stop = false;
            if (e.early_stop) {
                const double L = (double)loss;
                if (L < 1e-4) stop = true;
                else {
                    if (fabs(lp - L) < lp * e.break_threshold_ratio) bc += 1;
                    if (bc >= e.max_break_count) stop = true;
                    else lp = L;
                }
            }
            const int decision = stop ? NDP_DEC_ADVANCE : (st.iter + 1 >= e.iters ? NDP_DEC_STEP_ADVANCE : NDP_DEC_STEP);
            // the next state = this one with the fields below replaced, written field by field (a private copy of the struct with its
            // per-level array lived in scratch, and every thread of every workgroup paid the 64-byte store that initialised it)
            *nst = *stp;
            nst->loss = loss;
            nst->decision = decision;
            nst->total_evals = st.total_evals + 1;
            nst->step_level = st.level;
            nst->step_t = st.adam_t + 1;
            if (decision != NDP_DEC_ADVANCE) nst->total_steps = st.total_steps + 1;
            if (decision == NDP_DEC_STEP) {
                nst->iter = st.iter + 1;
                nst->adam_t = st.adam_t + 1;
                nst->break_counter = bc;
                nst->loss_prev = lp;
            } else {                                       // registration.py:242-249 + :179-180
                nst->level = st.level + 1;
                nst->iter = 0;
                nst->adam_t = 0;
                nst->break_counter = 0;
                nst->loss_prev = 1e6;
                nst->cur = st.cur ^ 1;
            }
            if (decision != NDP_DEC_STEP) nst->evals_per_level[st.level] = st.iter + 1;
        }
        PT(4);
        PT_FLUSH(48);
        return;
    }
    // ---- gradient of the loss wrt the warped points of this workgroup
    const int p = act ? vb * 256 + t : e.n_cap;                          // (an inactive thread owns no point)
    if (vb * 256 >= n) return;
    float *dO_row = e.dO + ((size_t)b * e.n_cap + p) * NDP_NHMAX;
    float w[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};
    // Requested up front, next to the warped point: the point's level input, which only the head backward at the end needs -- behind
    // the scatter it and the head record (below) were one more global round trip in the open (round 6).
    float xv[3] = {0.f, 0.f, 0.f};
    if (p < n) {
        const float *xin = e.pts + ((size_t)b * 2 + st.cur) * e.n_cap * 3 + 3 * p;
        w[0] = x_out[3 * p]; w[1] = x_out[3 * p + 1]; w[2] = x_out[3 * p + 2];
        xv[0] = xin[0]; xv[1] = xin[1]; xv[2] = xin[2];
    }
    const int i_self = p - gm.K;                     // sample index (negative for landmarks)
    // Every phase below is a chain of 1-2 us global round trips (tools/phase_timing.py), so what can be requested now is: the
    // chunk's nearest-source indices (local point of target c0 + t + 256 k, -1: not ours) travel with the row partials.
    const bool scatter = use_cd && vb * 256 + 255 >= gm.K;      // workgroup holds at least one sample
    const int i_lo = vb * 256 - gm.K;                      // sample index of thread 0
    // (UNCONDITIONAL loads at a clamped index -- idx_y is padded with -1 up to t_cap: as `cond ? idx_y[j] : -1` every one of the eight
    //  became a branch around a load with its own wait, eight dependent global round trips at the top of every gradient workgroup,
    //  a third of its time: round 6)
    int li[LG_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < LG_CHUNK / 256; ++k) li[k] = -1;
    if (scatter) {
        const int jcap = e.t_cap - 1;
        int raw[LG_CHUNK / 256];
#pragma unroll
        for (int k = 0; k < LG_CHUNK / 256; ++k) raw[k] = idx_y[min(t + 256 * k, jcap)];
#pragma unroll
        for (int k = 0; k < LG_CHUNK / 256; ++k) li[k] = act && t + 256 * k < gm.T ? raw[k] - i_lo : -1;
    }
    if (p < gm.K) {
        const float invK = 1.0f / (float)gm.K;
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] = 2.0f * (w[a] - ldmk_t[3 * p + a]) * invK;
    } else if (p < n && use_cd) {
        NnPart nx;
        if (rows_final) { nx.d2 = e.d2x[(size_t)b * e.n_cap + i_self]; nx.idx = e.idx_x[(size_t)b * e.n_cap + i_self]; }
        else {
            nx = nn_row_fold(rowpart, e.n_cap, gm.T, i_self, rows_cstep);
            e.d2x[(size_t)b * e.n_cap + i_self] = nx.d2;             // kept for inspection; nothing on the path reads them
            e.idx_x[(size_t)b * e.n_cap + i_self] = nx.idx;
        }
        const float d2 = nx.d2;
        if (!(d2 >= e.trunc)) {
            const float *yy = tgt + 3 * nx.idx;
            const float inv = 1.0f / ((float)gm.S * sqrtf(d2));
#pragma unroll
            for (int a = 0; a < 3; ++a) g[a] = (w[a] - yy[a]) * inv;
        }
    }
    // the head record travels under the counting sort's first passes (16 registers that the row fold above had no room for)
    static_assert(NDP_NHMAX == 16, "four float4 per head row");
    float4 hr0 = make_float4(0.f, 0.f, 0.f, 0.f), hr1 = hr0, hr2 = hr0, hr3 = hr0;
    if (p < n) {
        const float4 *hsrc = reinterpret_cast<const float4 *>(hrec + (size_t)p * NDP_HROW);
        hr0 = hsrc[0]; hr1 = hsrc[1]; hr2 = hsrc[2]; hr3 = hsrc[3];
    }
#define LOSS_HR_STORE() do { if (p < n) { float4 *hd_ = reinterpret_cast<float4 *>(rows + t * NDP_LROW); hd_[0] = hr0; hd_[1] = hr1; hd_[2] = hr2; hd_[3] = hr3; } } while (0)
    if (!scatter) LOSS_HR_STORE();
    PT(0);
    if (scatter) {
        // Targets whose nearest source point belongs to this workgroup, grouped per point by a counting
        // sort in LDS (O(T) per workgroup instead of a T-long scan per point), each group then sorted so
        // that the contributions are added in ascending target index -- the order of a sequential CPU
        // scatter-add, hence bit-identical to the oracle -- without any float atomics.
        const bool live = p >= gm.K && p < n;
        for (int c0 = 0; c0 < gm.T; c0 += LG_CHUNK) {
            const int cn = min(LG_CHUNK, gm.T - c0);
            if (c0 > 0) {
#pragma unroll
                for (int k = 0; k < LG_CHUNK / 256; ++k) {
                    const int j = t + 256 * k;
                    const int raw = idx_y[min(c0 + j, e.t_cap - 1)];
                    li[k] = act && j < cn ? raw - i_lo : -1;
                }
            }
            __syncthreads();
            if (act) cnt[t] = 0;
            __syncthreads();
            // pass 1: count
#pragma unroll
            for (int k = 0; k < LG_CHUNK / 256; ++k)
                if (li[k] >= 0 && li[k] < 256) atomicAdd(&cnt[li[k]], 1);
            __syncthreads();
            PT(1);
            // exclusive scan of cnt -> start: inclusive scan inside each wave by lane shuffles, the four wave totals through LDS (two
            // barriers; until round 6 a Hillis-Steele scan through LDS with sixteen of them -- integers: the same offsets)
            const int mine = cnt[t];
            int inc = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(inc, d);
                if ((t & 63) >= d) inc += v;
            }
            int *wsum = reinterpret_cast<int *>(red);                     // (the gradient workgroups have no other use for `red`)
            if (act && (t & 63) == 63) wsum[t >> 6] = inc;
            __syncthreads();
            int my_start = inc - mine;
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) my_start += w2 < (t >> 6) ? wsum[w2] : 0;
            if (act) start[t] = my_start;                                 // becomes the fill cursor
            if (c0 == 0) LOSS_HR_STORE();
            __syncthreads();
            PT(2);
            // pass 2: fill
#pragma unroll
            for (int k = 0; k < LG_CHUNK / 256; ++k)
                if (li[k] >= 0 && li[k] < 256) order[atomicAdd(&start[li[k]], 1)] = c0 + t + 256 * k;
            __syncthreads();
            PT(3);
            if (live && mine > 0) {
                int *bk = order + my_start;                               // this thread's private range
                for (int a = 1; a < mine; ++a) {                          // insertion sort, ascending target index
                    const int v = bk[a];
                    int q = a - 1;
                    while (q >= 0 && bk[q] > v) { bk[q + 1] = bk[q]; --q; }
                    bk[q + 1] = v;
                }
                for (int a0 = 0; a0 < mine; a0 += 4) {                    // four entries requested together, added in order
                    float d2q[4], yq[4][3];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = bk[min(a0 + u, mine - 1)];
                        d2q[u] = d2y[j];
                        yq[u][0] = tgt[3 * j]; yq[u][1] = tgt[3 * j + 1]; yq[u][2] = tgt[3 * j + 2];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (a0 + u < mine && !(d2q[u] >= e.trunc)) {
                            const float inv = 1.0f / ((float)gm.T * sqrtf(d2q[u]));
                            g[0] = fmaf(w[0] - yq[u][0], inv, g[0]);
                            g[1] = fmaf(w[1] - yq[u][1], inv, g[1]);
                            g[2] = fmaf(w[2] - yq[u][2], inv, g[2]);
                        }
                    }
                }
            }
        }
        if (gm.K > 0 && live) { g[0] = e.w_cd * g[0]; g[1] = e.w_cd * g[1]; g[2] = e.w_cd * g[2]; }   // registration.py:197
    }
    PT(4);
    // ---- per-point head backward: dO = mlp_scale * dL/d(scaled head outputs); zero rows pad the last tile
    float amax = 0.f;
    if (p < n) {
        float g_nr = 0.f;
        if (use_reg) {                                   // d/dnr of w_reg * mean(-log(1 - nr)), torch's BCE backward clamp
            const float nr = 1.0f / (1.0f + expf(-rows[t * NDP_LROW + hcl.row_nr]));
            const float den = (1.0f - nr) * nr;
            g_nr = e.w_reg * ((1.0f / (float)n) * (nr / (den > 1e-12f ? den : 1e-12f)));
        }
        point_head_bwd(hcl, nullptr, xv, g, g_nr, rows + t * NDP_LROW, dO_row, &amax);
    } else if (p < e.n_cap) {
#pragma unroll
        for (int j = 0; j < NDP_NHMAX; j += 4) *reinterpret_cast<float4 *>(dO_row + j) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (e.gmax) {                                        // the pair's max |dO|: the split backward scales its gradient operands by it
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        float *wm = rows;                                // (every thread is done with its row; one atomic per workgroup, not per wave)
        __syncthreads();
        if (act && (t & 63) == 0) wm[t >> 6] = amax;
        __syncthreads();
        if (t == 0 && act) {
            const float m4 = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
            if (m4 > 0.f) atomicMax(e.gmax + b, __float_as_uint(m4));      // non-negative floats order like their bit patterns
        }
    }
    PT(5);
    PT_FLUSH(36);
}

extern "C" __global__ void __launch_bounds__(256, 5)                     // five waves per SIMD: what the 31 KB of LDS allow (<= 96 registers)
k_eng_loss(ndp_engine e, int parity) {
    __shared__ LossSmem sm_;
    int b, vb;
    xcd_pair_block(gridDim.x, gridDim.y, b, vb);
    eng_loss_body(e, parity, b, vb, gridDim.x, threadIdx.x, true, sm_);
}

// backward of the live tiles of every pair that takes an Adam step this tick (two launches, see bwd2/bwd1)
__device__ __forceinline__ bool eng_bwd_job(const ndp_engine &e, int parity, BwdJob &job, bool zero_idle_partial) {
    const int b = blockIdx.y;
    const ndp_pair_state ns = e.state[(size_t)(parity ^ 1) * e.B + b];     // written by k_eng_loss this tick
    if (ns.decision == NDP_DEC_IDLE || ns.decision == NDP_DEC_ADVANCE) return false;
    const ndp_pair_geom gm = e.geom[b];
    const int n = gm.K + gm.S;
    const int n_tiles = (n + NDP_TILE - 1) / NDP_TILE;
    float *gpart = e.gpart + ((size_t)b * e.G + blockIdx.x) * e.p_stride;
    if ((int)blockIdx.x >= n_tiles) {                  // no tile for this workgroup: its partial is zero
        if (zero_idle_partial) for (int i = threadIdx.x; i < e.P; i += 256) gpart[i] = 0.f;
        return false;
    }
    job.params = e.params + ((size_t)b * e.m + ns.step_level) * e.p_stride;
    job.act = e.act + (size_t)b * 3 * e.n_cap * NDP_W;
    job.heads = e.heads + (size_t)b * e.n_cap * NDP_HROW;
    job.dO = e.dO + (size_t)b * e.n_cap * NDP_NHMAX;
    job.gpart = gpart;
    job.n = n; job.plane = e.n_cap; job.n_tiles = n_tiles;
    job.tile0 = blockIdx.x; job.tile_step = gridDim.x;
    job.dz_plane = job.act + 2 * (size_t)e.n_cap * NDP_W;
    job.h_plane = job.act + (size_t)e.n_cap * NDP_W;
    job.from_dO = 0; job.wh_off = 0; job.nh = 0;
    return true;
}

extern "C" __global__ void __launch_bounds__(256, 2)
k_eng_bwd2(ndp_engine e, int parity) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    BwdJob job;
    if (!eng_bwd_job(e, parity, job, true)) return;                 // first backward kernel of the tick: idle partials read as zero
    bwd_job_ndp_layer2(job, make_head_cfg(desc_at_level(e.desc, e.state[(size_t)(parity ^ 1) * e.B + blockIdx.y].step_level)).nh);
    PT_INIT;
    bwd2_body(make_head_cfg(desc_at_level(e.desc, 0)), job, sm);
    PT_FLUSH(0);
}

extern "C" __global__ void __launch_bounds__(256, 2)
k_eng_bwd1(ndp_engine e, int parity) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    BwdJob job;
    if (!eng_bwd_job(e, parity, job, false)) return;
    PT_INIT;
    bwd1_body(make_head_cfg(desc_at_level(e.desc, 0)), job, sm);
    PT_FLUSH(24);
}

#include "ndp_bwd_split.inc"
#include "ndp_bwd_fused.inc"

// fold the G partial gradients in index order, Adam step, level hand-over (fresh Adam state).
// (Four parameters per thread on 16-byte accesses: no faster at 128 pairs -- 0.0305 against 0.0315 ms -- and TWICE as slow at batch 1,
//  where the G = 32 partials are folded by a quarter of the threads: 0.023 against 0.012 ms.  One parameter per thread it stays.)
// parameter i of pair b: fold, Adam, hand-over (the whole update stage is this, for every i < P)
__device__ __forceinline__ void eng_update_param(const ndp_engine &e, int b, const ndp_pair_state &ns, const ndp_layer_desc &dl, int i) {
    float *m = e.adam_m + (size_t)b * e.p_stride, *v = e.adam_v + (size_t)b * e.p_stride;
    if (i >= ndp_param_count(&dl)) {                     // level 0 has no gate row: nothing to step, keep moments clean
        if (ns.decision != NDP_DEC_STEP) { m[i] = 0.f; v[i] = 0.f; }
        return;
    }
    if (ns.decision != NDP_DEC_ADVANCE) {
        const float *gp = e.gpart + (size_t)b * e.G * e.p_stride;
        float *p = e.params + ((size_t)b * e.m + ns.step_level) * e.p_stride;
        float pi = p[i], mi = m[i], vi = v[i];                           // (requested with the partials, not behind their fold)
        float g = __builtin_nontemporal_load(gp + i);
        int k = 1;
        // The other partials are requested TOGETHER (clamped index, added in index order while k < G): batch 1 folds G = 32 of them, and as
        // three batches of eight plus a tail of seven single loads that was eleven dependent global round trips (round 6).
        if (e.G > 9) {
            for (; k < e.G; k += 32) {
                float q[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) q[u] = gp[(size_t)min(k + u, e.G - 1) * e.p_stride + i];
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (k + u < e.G) g += q[u];
            }
        } else if (e.G > 3) {
            for (; k < e.G; k += 8) {
                float q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = gp[(size_t)min(k + u, e.G - 1) * e.p_stride + i];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k + u < e.G) g += q[u];
            }
        }
        for (; k < e.G; ++k) g += gp[(size_t)k * e.p_stride + i];
        adam_update(pi, g, mi, vi, e.adam_w1, e.adam_b2, e.adam_w2, e.adam_tab[2 * ns.step_t],
                    e.adam_tab[2 * ns.step_t + 1], e.adam_eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
    if (ns.decision != NDP_DEC_STEP) { m[i] = 0.f; v[i] = 0.f; }             // registration.py:176
}
extern "C" __global__ void __launch_bounds__(256)
k_eng_update(ndp_engine e, int parity) {
    const int b = blockIdx.y;
    // (the three fields the step needs, requested side by side and tested once: behind the test on the decision the level and the step
    //  number were a second dependent scalar round trip)
    const ndp_pair_state *nsp = e.state + (size_t)(parity ^ 1) * e.B + b;   // written by k_eng_loss this tick
    ndp_pair_state ns;
    ns.decision = nsp->decision; ns.step_level = nsp->step_level; ns.step_t = nsp->step_t;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if ((ns.decision == NDP_DEC_IDLE) | (i >= e.P) | (ns.step_level < 0) | (ns.step_t < 0)) return;
    eng_update_param(e, b, ns, desc_at_level(e.desc, ns.step_level), i);
}
// The update stage when the fused backward has stepped the two 128 x 128 matrices behind its tile loop (bf_adam_in_tail: G == 1):
// what is left -- [W0 | b0], b1, [b2 | Wh | bh] -- on a COMPACT grid (8 workgroups per pair at six heads instead of 136: the empty
// ones of the full grid cost more than the step itself, 17 us at 256 pairs).  Same eng_update_param, every decision handled there.
__host__ __device__ inline int upd_rest_count(int P) { const ndp_layer_desc dd = {NDP_W, 2, 0, 0, 0, 0.f}; return ndp_off_Wi(&dd, 1) + NDP_W + (P - ndp_off_bi(&dd, 2)); }
extern "C" __global__ void __launch_bounds__(256)
k_eng_update_rest(ndp_engine e, int parity) {
    const int b = blockIdx.y;
    const ndp_pair_state *nsp = e.state + (size_t)(parity ^ 1) * e.B + b;   // written by k_eng_loss this tick
    ndp_pair_state ns;
    ns.decision = nsp->decision; ns.step_level = nsp->step_level; ns.step_t = nsp->step_t;
    const ndp_layer_desc dd = {NDP_W, 2, 0, 0, 0, 0.f};
    const int n0 = ndp_off_Wi(&dd, 1);                              // [0, n0): W0 | b0
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int i = c < n0 ? c : (c < n0 + NDP_W ? ndp_off_bi(&dd, 1) + (c - n0) : ndp_off_bi(&dd, 2) + (c - n0 - NDP_W));
    if ((ns.decision == NDP_DEC_IDLE) | (i >= e.P) | (ns.step_level < 0) | (ns.step_t < 0)) return;
    eng_update_param(e, b, ns, desc_at_level(e.desc, ns.step_level), i);
}

#include "ndp_tick_small.inc"
#include "ndp_generic.inc"

// ------------------------------------------------------------------------------------------------
// Neural scene-flow prior baseline (nets.py:256-292): one launch per layer
// ------------------------------------------------------------------------------------------------
// h1 = relu(W1 x + b1): thread -> (row, 4 consecutive outputs), coalesced float4 rows; zero rows beyond n
extern "C" __global__ void __launch_bounds__(256)
k_nsfp_in(const float *params, const float *x, int n, float *h1 /*[plane][128]*/, int plane) {
    const int idx = blockIdx.x * 256 + threadIdx.x;           // float4 index
    const int p = idx >> 5, o = 4 * (idx & 31);
    if (p >= plane) return;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < n) {
        const float *W = params + ndp_nsfp_off_W(1), *b = params + ndp_nsfp_off_b(1);
        const float x0 = x[3 * (size_t)p], x1 = x[3 * (size_t)p + 1], x2 = x[3 * (size_t)p + 2];
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *w = W + 3 * (o + c);
            const float z = fmaf(w[2], x2, fmaf(w[1], x1, fmaf(w[0], x0, b[o + c])));
            v[c] = z > 0.f ? z : 0.f;
        }
        r = make_float4(v[0], v[1], v[2], v[3]);
    }
    reinterpret_cast<float4 *>(h1)[idx] = r;
}

// y = relu(W h + b), 128 -> 128, tiles of 64 points; wave w owns output columns [32w, 32w+32), weight slice stationary
extern "C" __global__ void __launch_bounds__(256, 2)
k_nsfp_dense(const float *W, const float *b, const float *hin, float *hout, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l31 = lane & 31, h = lane >> 5;
    float *bufA = sm, *bufB = sm + 64 * NDP_LD;
    float w[64];
    load_w_fwd(W, sm, wv, l31, h, w);
    __syncthreads();                               // the weight image shares the tile buffers
    const float bias = b[32 * wv + l31];
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        load_tile_to_lds(hin + (size_t)tile * NDP_TILE * NDP_W, bufA);
        __syncthreads();
        f32x16 acc0, acc1;
        acc_init_bias(bias, h, acc0, acc1);
        tile_gemm_64x32(bufA, w, l31, h, acc0, acc1);
        epilogue_relu(acc0, acc1, bufB, wv, l31, h);
        __syncthreads();
        store_tile_from_lds(bufB, hout + (size_t)tile * NDP_TILE * NDP_W);
        __syncthreads();
    }
}

// x_out = x + W9 h8 + b9: thread (point = t & 63, coordinate = t >> 6 < 3), four independent fmaf chains
extern "C" __global__ void __launch_bounds__(256)
k_nsfp_out(const float *params, const float *h8, const float *x, int n, float *x_out) {
    __shared__ __attribute__((aligned(16))) float tile[64 * NDP_LD];
    __shared__ __attribute__((aligned(16))) float w9[3 * NDP_W];
    const int t = threadIdx.x, base = blockIdx.x * NDP_TILE;
    load_tile_to_lds(h8 + (size_t)base * NDP_W, tile);
    for (int i = t; i < 3 * NDP_W; i += 256) w9[i] = params[ndp_nsfp_off_W(NDP_NSFP_LAYERS) + i];
    __syncthreads();
    const int p = base + (t & 63), j = t >> 6;
    if (j < 3 && p < n) {
        const float *hr = tile + (t & 63) * NDP_LD, *wr = w9 + j * NDP_W;
        float a0 = params[ndp_nsfp_off_b(NDP_NSFP_LAYERS) + j], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
        for (int k4 = 0; k4 < 32; ++k4) {
            const float4 hv = *reinterpret_cast<const float4 *>(hr + 4 * k4);
            const float4 wv4 = *reinterpret_cast<const float4 *>(wr + 4 * k4);
            a0 = fmaf(wv4.x, hv.x, a0); a1 = fmaf(wv4.y, hv.y, a1);
            a2 = fmaf(wv4.z, hv.z, a2); a3 = fmaf(wv4.w, hv.w, a3);
        }
        x_out[3 * (size_t)p + j] = x[3 * (size_t)p + j] + ((a0 + a1) + (a2 + a3));
    }
}

// dO[p][0..2] = g[p], zero elsewhere (rows up to plane): the output layer then runs through the head-stage kernel
extern "C" __global__ void __launch_bounds__(256)
k_nsfp_pack_g(const float *g, int n, int plane, float *dO) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= plane) return;
    float4 *o = reinterpret_cast<float4 *>(dO + (size_t)p * NDP_NHMAX);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    o[0] = p < n ? make_float4(g[3 * (size_t)p], g[3 * (size_t)p + 1], g[3 * (size_t)p + 2], 0.f) : z;
    o[1] = z; o[2] = z; o[3] = z;
}

// dW1[o][c] += sum_p dz1[p][o] x[p][c] ; db1[o] += sum_p dz1[p][o]: thread holds 8 rows x 4 columns of every tile
extern "C" __global__ void __launch_bounds__(256)
k_nsfp_in_bwd(const float *dz1 /*[plane][128]*/, const float *x, int n, int n_tiles, float *gpart, int p_stride) {
    __shared__ __attribute__((aligned(16))) float sc[8][NDP_W * 4];
    const int t = threadIdx.x, rg = t >> 5, o = 4 * (t & 31);
    float aw[4][3], ab[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { ab[c] = 0.f; aw[c][0] = aw[c][1] = aw[c][2] = 0.f; }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int base = tile * NDP_TILE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = base + rg + 8 * i;
            const float4 z = *reinterpret_cast<const float4 *>(dz1 + (size_t)p * NDP_W + o);
            float xv[3] = {0.f, 0.f, 0.f};
            if (p < n) { xv[0] = x[3 * (size_t)p]; xv[1] = x[3 * (size_t)p + 1]; xv[2] = x[3 * (size_t)p + 2]; }
            const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ab[c] += zz[c];
#pragma unroll
                for (int a = 0; a < 3; ++a) aw[c][a] = fmaf(zz[c], xv[a], aw[c][a]);
            }
        }
    }
    // fold the 8 row groups in group order
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float *s = &sc[rg][4 * (o + c)];
        s[0] = aw[c][0]; s[1] = aw[c][1]; s[2] = aw[c][2]; s[3] = ab[c];
    }
    __syncthreads();
    if (t < NDP_W) {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8)
#pragma unroll
            for (int a = 0; a < 4; ++a) r[a] += sc[g8][4 * t + a];
        float *G = gpart + (size_t)blockIdx.x * p_stride;
        G[ndp_nsfp_off_W(1) + 3 * t] = r[0]; G[ndp_nsfp_off_W(1) + 3 * t + 1] = r[1]; G[ndp_nsfp_off_W(1) + 3 * t + 2] = r[2];
        G[ndp_nsfp_off_b(1) + t] = r[3];
    }
}

// ---- pair preparation (registration.py:150-164) and slot (re)fill, batched over pairs ----------------------
// means of two clouds: blockIdx.x = 0 source, 1 target.  Double accumulation in a fixed order, one rounding.
extern "C" __global__ void __launch_bounds__(1024)
k_pair_means(const float *src, int n_src, const float *tgt, int n_tgt, float *means) {
    __shared__ double red[3][1024];
    const float *x = blockIdx.x ? tgt : src;
    const int n = blockIdx.x ? n_tgt : n_src, t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int p0 = t; p0 < n; p0 += 4 * 1024) {               // four independent loads in flight per thread
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + 1024 * u;
            v[u][0] = v[u][1] = v[u][2] = 0.f;
            if (p < n) { v[u][0] = x[3 * (size_t)p]; v[u][1] = x[3 * (size_t)p + 1]; v[u][2] = x[3 * (size_t)p + 2]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 += (double)v[u][0]; s1 += (double)v[u][1]; s2 += (double)v[u][2]; }
    }
    red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if (t < d) { red[0][t] += red[0][t + d]; red[1][t] += red[1][t + d]; red[2][t] += red[2][t + d]; }
        __syncthreads();
    }
    if (t < 4) means[4 * blockIdx.x + t] = t < 3 ? (float)(red[t][0] / (double)n) : 0.f;
}

struct LoadJobs {
    ndp_load_job j[NDP_MAX_LOAD_JOBS];
};
// the means of the raw clouds of the jobs that ask for them (n_src > 0), ONE launch per load call instead of one k_pair_means per pair
// (24 576 launches per bench run, 4.6 % of the kernel time under two engines' contention: profiles/r04_bench_kernel_stats.csv):
// blockIdx.y = job, blockIdx.x = 0 source / 1 target; per block the code of k_pair_means -- same order, same bits
extern "C" __global__ void __launch_bounds__(1024)
k_pair_means_jobs(LoadJobs jobs) {
    __shared__ double red[3][1024];
    const ndp_load_job jb = jobs.j[blockIdx.y];
    if (!jb.params || !jb.means || jb.n_src <= 0) return;
    const float *x = blockIdx.x ? jb.tgt : jb.src;
    const int n = blockIdx.x ? jb.n_tgt : jb.n_src, t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int p0 = t; p0 < n; p0 += 4 * 1024) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + 1024 * u;
            v[u][0] = v[u][1] = v[u][2] = 0.f;
            if (p < n) { v[u][0] = x[3 * (size_t)p]; v[u][1] = x[3 * (size_t)p + 1]; v[u][2] = x[3 * (size_t)p + 2]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 += (double)v[u][0]; s1 += (double)v[u][1]; s2 += (double)v[u][2]; }
    }
    red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if (t < d) { red[0][t] += red[0][t + d]; red[1][t] += red[1][t + d]; red[2][t] += red[2][t + d]; }
        __syncthreads();
    }
    if (t < 4) jb.means[4 * blockIdx.x + t] = t < 3 ? (float)(red[t][0] / (double)n) : 0.f;
}
extern "C" __global__ void __launch_bounds__(256)
k_eng_load(ndp_engine e, int parity, LoadJobs jobs) {
    const ndp_load_job jb = jobs.j[blockIdx.y];
    const int b = jb.slot, t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    ndp_pair_state *st = e.state + (size_t)parity * e.B + b;
    if (!jb.params) {                                        // park: the slot reads as finished
        if (t == 0) {
            ndp_pair_state c;
            memset(&c, 0, sizeof c);
            c.level = e.m;
            c.decision = NDP_DEC_IDLE;
            *st = c;
        }
        return;
    }
    float ms[3] = {0.f, 0.f, 0.f}, mt[3] = {0.f, 0.f, 0.f};
    if (jb.means) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { ms[a] = jb.means[a]; mt[a] = jb.means[4 + a]; }
    }
    const int n = jb.K + jb.S;
    // centred landmarks + centred source samples -> point buffer 0 (rest of the plane zero)
    float *pts = e.pts + (size_t)b * 2 * e.n_cap * 3;
    for (int i = t; i < e.n_cap; i += stride) {
        float v[3] = {0.f, 0.f, 0.f};
        if (i < n) {
            const float *q = i < jb.K ? jb.ldmk_s + 3 * (size_t)i
                                      : jb.src + 3 * (size_t)(jb.perm_s ? jb.perm_s[i - jb.K] : i - jb.K);
#pragma unroll
            for (int a = 0; a < 3; ++a) v[a] = q[a] - ms[a];
        }
        pts[3 * i] = v[0]; pts[3 * i + 1] = v[1]; pts[3 * i + 2] = v[2];
    }
    float *lt = e.ldmk_t + (size_t)b * e.n_cap * 3;
    for (int i = t; i < jb.K; i += stride) {
#pragma unroll
        for (int a = 0; a < 3; ++a) lt[3 * i + a] = jb.ldmk_t[3 * (size_t)i + a] - mt[a];
    }
    float *tg = e.tgt + (size_t)b * e.t_cap * 3;
    for (int i = t; i < jb.T; i += stride) {
        const float *q = jb.tgt + 3 * (size_t)(jb.perm_t ? jb.perm_t[i] : i);
#pragma unroll
        for (int a = 0; a < 3; ++a) tg[3 * i + a] = q[a] - mt[a];
    }
    // parameters of every level, fresh Adam moments
    {
        const float4 *src = reinterpret_cast<const float4 *>(jb.params);
        float4 *dst = reinterpret_cast<float4 *>(e.params + (size_t)b * e.m * e.p_stride);
        const int n4 = e.m * e.p_stride / 4;
        for (int i = t; i < n4; i += stride) dst[i] = src[i];
        float4 *am = reinterpret_cast<float4 *>(e.adam_m + (size_t)b * e.p_stride);
        float4 *av = reinterpret_cast<float4 *>(e.adam_v + (size_t)b * e.p_stride);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = t; i < e.p_stride / 4; i += stride) { am[i] = z; av[i] = z; }
    }
    if (t == 0) {
        ndp_pair_geom g;
        g.K = jb.K; g.S = jb.S; g.T = jb.T; g.pad = 0;
        e.geom[b] = g;
        ndp_pair_state c;
        memset(&c, 0, sizeof c);
        c.loss_prev = 1e6;                                   // registration.py:179
        *st = c;
    }
}

// ------------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
static int fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
static int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}
#define HIP_TRY(expr, what)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return hip_fail(_e, what);      \
    } while (0)

static int check_desc(const ndp_layer_desc *d) {
    if (!d) return fail(NDP_E_INVALID, "null layer descriptor");
    if (gen_is_generic(*d) && !gen_supported(*d))                       // 128 / 3: the MFMA kernels; anything else: csrc/ndp_generic.inc
        return fail(NDP_E_UNSUPPORTED, "width must be 1..256 and depth 1..4 (width=128, depth=3 run on the MFMA kernels, the rest on the generic fp32 kernels)");
    if (d->motion < 0 || d->motion > 2) return fail(NDP_E_INVALID, "bad motion type");
    if (d->motion != NDP_MOTION_SFLOW && (d->rotfmt < NDP_ROT_AXIS_ANGLE || d->rotfmt > NDP_ROT_6D))
        return fail(NDP_E_INVALID, "bad rotation_format");
    return 0;
}
static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

static int set_smem(const void *fn, int bytes) {
    static thread_local const void *done[32];
    for (auto d : done) if (d == fn) return 0;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "hipFuncSetAttribute");
    for (auto &d : done) if (!d) { d = fn; break; }
    return 0;
}

#ifdef NDP_PHASE_TIMING
extern "C" int ndp_debug_phase_read(unsigned long long *out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 96) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[96] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifndef NDP_BUILD_ID
#define NDP_BUILD_ID "unversioned"
#endif
extern "C" int ndp_version(void) { return 203; }           // 201: ndp_load_job gained n_src / n_tgt (88 bytes), `means` in/out; 202: h2 as a plane image under gemm_mode 7; 203: gemm_mode bits 512 / 1024, at G == 1 the matrix blocks of gpart are not written
extern "C" const char *ndp_last_error(void) { return g_err; }
static const char k_build_tag[] = "NDP_BUILD_ID=" NDP_BUILD_ID;        // the loader finds this tag in the file without loading it
extern "C" const char *ndp_build_id(void) { return k_build_tag + 13; }
extern "C" int ndp_abi_sizes(int *out) {
    if (!out) return fail(NDP_E_INVALID, "ndp_abi_sizes: null pointer");
    out[0] = (int)sizeof(ndp_layer_desc); out[1] = (int)sizeof(ndp_pair_geom); out[2] = (int)sizeof(ndp_pair_state);
    out[3] = (int)sizeof(ndp_engine);     out[4] = (int)sizeof(ndp_warp_job);  out[5] = (int)sizeof(ndp_load_job);
    return 0;
}

extern "C" int ndp_level_fwd(const ndp_layer_desc *desc, const float *params, int level, int k0,
                             const float *x, int n, float *x_out, float *act, float *heads, float *nonrig_out,
                             void *stream) {
    if (int rc = check_desc(desc)) return rc;
    if (n < 0 || !params || (n > 0 && (!x || !x_out))) return fail(NDP_E_INVALID, "ndp_level_fwd: null pointer / negative n");
    if (!aligned16(params) || (act && !aligned16(act)) || (heads && !aligned16(heads)))
        return fail(NDP_E_INVALID, "ndp_level_fwd: params/act/heads must be 16-byte aligned");
    if (n == 0) return 0;
    LevelJob job;
    job.params = params; job.freq = ldexpf(1.0f, level + 1 + k0);
    job.x_in = x; job.x_out = x_out; job.act = act; job.heads = heads;
    job.nonrig = desc->nonrigidity ? nonrig_out : nullptr;
    job.n = n; job.n_tiles = (n + NDP_TILE - 1) / NDP_TILE; job.plane = job.n_tiles * NDP_TILE;
    job.tile0 = 0; job.tile_step = 0;
    if (gen_is_generic(*desc)) {                                         // act: [n_hidden + 1][plane][width]
        if (int rc = set_smem((const void *)k_gen_level_fwd, kSmemGenFwdMax)) return rc;
        hipLaunchKernelGGL(k_gen_level_fwd, dim3(job.n_tiles < 1024 ? job.n_tiles : 1024), dim3(256), gen_fwd_floats(desc->width) * 4, (hipStream_t)stream,
                           make_head_cfg(*desc), *desc, job);
        HIP_TRY(hipGetLastError(), "k_gen_level_fwd launch");
        return 0;
    }
    if (int rc = set_smem((const void *)k_level_fwd, kSmemFwdBytes)) return rc;
    // one tile per workgroup: measured best for the final all-point warp (more tiles per workgroup save weight
    // loads but lengthen the warp, and throughput dropped 478 -> 438 pairs/s at 4 tiles per workgroup)
    const int grid = job.n_tiles < 1024 ? job.n_tiles : 1024;
    hipLaunchKernelGGL(k_level_fwd, dim3(grid), dim3(256), kSmemFwdBytes, (hipStream_t)stream, make_head_cfg(*desc), job);
    HIP_TRY(hipGetLastError(), "k_level_fwd launch");
    return 0;
}

extern "C" int ndp_level_bwd(const ndp_layer_desc *desc, const float *params, int level, int k0,
                             const float *x, int n, float *act, const float *heads, const float *g, const float *g_nr,
                             float *dO_work, float *grads_part, int n_part, int p_stride, void *stream) {
    (void)level; (void)k0;
    if (int rc = check_desc(desc)) return rc;
    if (n <= 0 || !params || !x || !act || !heads || !g || !dO_work || !grads_part || n_part < 1)
        return fail(NDP_E_INVALID, "ndp_level_bwd: null pointer / bad sizes");
    if (p_stride < ndp_param_count(desc)) return fail(NDP_E_INVALID, "ndp_level_bwd: p_stride < P");
    if (!aligned16(params) || !aligned16(act) || !aligned16(heads) || !aligned16(dO_work))
        return fail(NDP_E_INVALID, "ndp_level_bwd: params/act/heads/dO_work must be 16-byte aligned");
    BwdJob job;
    memset(&job, 0, sizeof job);
    job.params = params; job.act = act; job.heads = heads; job.dO = dO_work; job.gpart = grads_part;
    job.n = n; job.n_tiles = (n + NDP_TILE - 1) / NDP_TILE; job.plane = job.n_tiles * NDP_TILE;
    hipStream_t s = (hipStream_t)stream;
    if (n_part > job.n_tiles) {
        // partials with no tile must read as zero
        HIP_TRY(hipMemsetAsync(grads_part + (size_t)job.n_tiles * p_stride, 0,
                               sizeof(float) * (size_t)(n_part - job.n_tiles) * p_stride, s), "memset");
        n_part = job.n_tiles;
    }
    const HeadCfg hc = make_head_cfg(*desc);
    hipLaunchKernelGGL(k_head_bwd, dim3((job.plane + 255) / 256), dim3(256), 0, s, hc, x, heads, g,
                       desc->nonrigidity ? g_nr : nullptr, n, job.plane, dO_work);
    if (gen_is_generic(*desc)) {
        if (int rc = set_smem((const void *)k_gen_level_bwd, kSmemGenBwdMax)) return rc;
        hipLaunchKernelGGL(k_gen_level_bwd, dim3(n_part), dim3(256), gen_bwd_floats(desc->width) * 4, s, hc, *desc, job, p_stride);
        HIP_TRY(hipGetLastError(), "generic level backward launch");
        return 0;
    }
    if (int rc = set_smem((const void *)k_level_bwd2, kSmemBwdBytes)) return rc;
    if (int rc = set_smem((const void *)k_level_bwd1, kSmemBwdBytes)) return rc;
    job.dz_plane = act + 2 * (size_t)job.plane * NDP_W;
    job.h_plane = act + (size_t)job.plane * NDP_W;
    bwd_job_ndp_layer2(job, hc.nh);
    hipLaunchKernelGGL(k_level_bwd2, dim3(n_part), dim3(256), kSmemBwdBytes, s, hc, job, p_stride);
    hipLaunchKernelGGL(k_level_bwd1, dim3(n_part), dim3(256), kSmemBwdBytes, s, hc, job, p_stride);
    HIP_TRY(hipGetLastError(), "level backward launch");
    return 0;
}

extern "C" int ndp_grad_reduce(const float *grads_part, int n_part, int p_stride, int P, float *grads, void *stream) {
    if (!grads_part || !grads || n_part < 1 || P < 1) return fail(NDP_E_INVALID, "ndp_grad_reduce: bad arguments");
    hipLaunchKernelGGL(k_grad_reduce, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, grads_part, n_part, p_stride, P, grads);
    HIP_TRY(hipGetLastError(), "k_grad_reduce launch");
    return 0;
}

extern "C" int ndp_pyramid_fwd_batch(const ndp_layer_desc *desc, int m, int k0, int p_stride,
                                     const ndp_warp_job *jobs, int n_jobs, void *stream);

extern "C" int ndp_pyramid_fwd(const ndp_layer_desc *desc, int m, int k0, const float *params_all, int p_stride,
                               const float *x, int n, float *x_out, void *stream) {
    if (int rc = check_desc(desc)) return rc;
    if (m < 0 || m > NDP_MAX_LEVELS || n < 0 || !x_out || (n > 0 && !x)) return fail(NDP_E_INVALID, "ndp_pyramid_fwd: bad arguments");
    if (n == 0) return 0;
    if (m == 0) {
        HIP_TRY(hipMemcpyAsync(x_out, x, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream), "memcpy");
        return 0;
    }
    ndp_warp_job job;
    memset(&job, 0, sizeof job);
    job.params = params_all; job.x = x; job.x_out = x_out; job.n = n;
    return ndp_pyramid_fwd_batch(desc, m, k0, p_stride, &job, 1, stream);
}

static int pyramid_fwd_batch_impl(const ndp_layer_desc *desc, int m, int k0, int p_stride, const ndp_warp_job *jobs, int n_jobs,
                                  void *stream, bool split, int tiles = P8_TILES) {
    if (int rc = check_desc(desc)) return rc;
    if (m < 1 || m > NDP_MAX_LEVELS || n_jobs < 0 || (n_jobs > 0 && !jobs) || p_stride < ndp_param_count(desc) || (p_stride & 3))
        return fail(NDP_E_INVALID, "ndp_pyramid_fwd_batch: bad arguments");
    const bool generic = gen_is_generic(*desc);                        // one arithmetic there: `split` and `tiles` select nothing
    if (generic) { if (int rc = set_smem((const void *)k_gen_pyramid_fwd, kSmemGenFwdMax)) return rc; }
    else if (split) { if (int rc = set_smem((const void *)k_pyramid_fwd8, kSmemPyr8Bytes)) return rc; }
    else if (int rc = set_smem((const void *)k_pyramid_fwd, kSmemFwdBytes)) return rc;
    if (split && (tiles < 1 || tiles > P8_TILES_MAX)) return fail(NDP_E_INVALID, "ndp_pyramid_fwd_batch_split_tiles: tiles per workgroup must be 1..8");
    const int per_wg = generic ? NDP_TILE : NDP_TILE * (split ? tiles : NDP_PYR_TILES);       // points per workgroup
    for (int j0 = 0; j0 < n_jobs; j0 += NDP_MAX_WARP_JOBS) {
        WarpJobs wj;
        memset(&wj, 0, sizeof wj);
        int cnt = 0, max_wgs = 0;
        for (int j = j0; j < n_jobs && cnt < NDP_MAX_WARP_JOBS; ++j) {
            const ndp_warp_job &q = jobs[j];
            if (q.n < 0 || (q.n > 0 && (!q.params || !q.x || !q.x_out))) return fail(NDP_E_INVALID, "ndp_pyramid_fwd_batch: null pointer / negative n");
            if (!aligned16(q.params)) return fail(NDP_E_INVALID, "ndp_pyramid_fwd_batch: params must be 16-byte aligned");
            if (q.n == 0) continue;
            wj.j[cnt++] = q;
            const int wgs = (q.n + per_wg - 1) / per_wg;                                          // workgroups of this cloud
            if (wgs > max_wgs) max_wgs = wgs;
        }
        if (!cnt) continue;
        if (generic) hipLaunchKernelGGL(k_gen_pyramid_fwd, dim3(max_wgs, cnt), dim3(256), gen_fwd_floats(desc->width) * 4, (hipStream_t)stream, *desc, m, k0, p_stride, wj);
        else if (split) hipLaunchKernelGGL(k_pyramid_fwd8, dim3(max_wgs, cnt), dim3(512), kSmemPyr8Bytes, (hipStream_t)stream, *desc, m, k0, p_stride, wj, tiles);
        else hipLaunchKernelGGL(k_pyramid_fwd, dim3(max_wgs, cnt), dim3(256), kSmemFwdBytes, (hipStream_t)stream, *desc, m, k0, p_stride, wj);
        HIP_TRY(hipGetLastError(), "k_pyramid_fwd launch");
    }
    return 0;
}

extern "C" int ndp_pyramid_fwd_batch(const ndp_layer_desc *desc, int m, int k0, int p_stride,
                                     const ndp_warp_job *jobs, int n_jobs, void *stream) {
    return pyramid_fwd_batch_impl(desc, m, k0, p_stride, jobs, n_jobs, stream, false);
}

// The same warp with the engine's split arithmetic (gemm_mode & 1): the 128-wide contractions as three-way bf16 splits on the
// bf16 MFMA (k_pyramid_fwd8) -- fp32-level accuracy (1e-5 of the fp32-MFMA kernel on warped coordinates), not bitwise the chain.
extern "C" int ndp_pyramid_fwd_batch_split(const ndp_layer_desc *desc, int m, int k0, int p_stride,
                                           const ndp_warp_job *jobs, int n_jobs, void *stream) {
    return pyramid_fwd_batch_impl(desc, m, k0, p_stride, jobs, n_jobs, stream, true);
}
// ... with `tiles` 64-point tiles per workgroup (1..8; the entry above: 4).  More tiles per workgroup = fewer weight prologues per cloud
// (less CU-time per cloud, the batched engine's choice) at a longer latency of the launch (fewer, longer workgroups).  Same bits.
extern "C" int ndp_pyramid_fwd_batch_split_tiles(const ndp_layer_desc *desc, int m, int k0, int p_stride,
                                                 const ndp_warp_job *jobs, int n_jobs, int tiles, void *stream) {
    return pyramid_fwd_batch_impl(desc, m, k0, p_stride, jobs, n_jobs, stream, true, tiles);
}

extern "C" int ndp_pair_means(const float *src, int n_src, const float *tgt, int n_tgt, float *means, void *stream) {
    if (!src || !tgt || !means || n_src < 1 || n_tgt < 1) return fail(NDP_E_INVALID, "ndp_pair_means: bad arguments");
    hipLaunchKernelGGL(k_pair_means, dim3(2), dim3(1024), 0, (hipStream_t)stream, src, n_src, tgt, n_tgt, means);
    HIP_TRY(hipGetLastError(), "k_pair_means launch");
    return 0;
}

static int check_engine(const ndp_engine *e, const char *who) {
    if (!e) return fail(NDP_E_INVALID, "null engine");
    if (int rc = check_desc(&e->desc)) return rc;
    if (e->B < 1 || e->G < 1 || e->m < 1 || e->m > NDP_MAX_LEVELS || e->n_cap % NDP_TILE || e->t_cap % NDP_TILE ||
        e->P != ndp_param_count(&e->desc) || e->p_stride < e->P || (e->p_stride & 3)) {
        snprintf(g_err, sizeof g_err, "%s: inconsistent engine descriptor", who);
        return NDP_E_INVALID;
    }
    if (!e->geom || !e->state || !e->pts || !e->params || !e->gpart || !e->adam_m || !e->adam_v || !e->act ||
        !e->heads || !e->adam_tab || !e->dO) {
        snprintf(g_err, sizeof g_err, "%s: null buffer", who);
        return NDP_E_INVALID;
    }
    if ((e->gemm_mode & 6) && !e->gmax) {
        snprintf(g_err, sizeof g_err, "%s: the split backward (gemm_mode & 6) needs the gmax buffer", who);
        return NDP_E_INVALID;
    }
    return 0;
}

extern "C" int ndp_engine_load(const ndp_engine *e, int tick, const ndp_load_job *jobs, int n_jobs, void *stream) {
    if (int rc = check_engine(e, "ndp_engine_load")) return rc;
    if (n_jobs < 0 || n_jobs > NDP_MAX_LOAD_JOBS || (n_jobs > 0 && !jobs)) return fail(NDP_E_INVALID, "ndp_engine_load: bad job count");
    if (n_jobs == 0) return 0;
    LoadJobs lj;
    memset(&lj, 0, sizeof lj);
    bool any_means = false;
    for (int j = 0; j < n_jobs; ++j) {
        const ndp_load_job &q = jobs[j];
        if (q.slot < 0 || q.slot >= e->B) return fail(NDP_E_INVALID, "ndp_engine_load: slot out of range");
        if (q.params) {
            if (q.K < 0 || q.S < 0 || q.T < 0 || q.K + q.S < 1 || q.K + q.S > e->n_cap || q.T > e->t_cap)
                return fail(NDP_E_INVALID, "ndp_engine_load: pair does not fit the engine capacities");
            if ((q.K > 0 && (!q.ldmk_s || !q.ldmk_t)) || (q.S > 0 && !q.src) || (q.T > 0 && (!q.tgt || !e->tgt)) ||
                (q.K > 0 && !e->ldmk_t))
                return fail(NDP_E_INVALID, "ndp_engine_load: null cloud pointer");
            if (!aligned16(q.params)) return fail(NDP_E_INVALID, "ndp_engine_load: params must be 16-byte aligned");
        }
        if (q.params && q.n_src > 0) {
            if (!q.means || !q.src || !q.tgt || q.n_tgt < 1) return fail(NDP_E_INVALID, "ndp_engine_load: means to compute need src, tgt, n_tgt and the means buffer");
            any_means = true;
        }
        lj.j[j] = q;
    }
    if (any_means) {
        hipLaunchKernelGGL(k_pair_means_jobs, dim3(2, n_jobs), dim3(1024), 0, (hipStream_t)stream, lj);
        HIP_TRY(hipGetLastError(), "k_pair_means_jobs launch");
    }
    hipLaunchKernelGGL(k_eng_load, dim3(32, n_jobs), dim3(256), 0, (hipStream_t)stream, *e, tick & 1, lj);
    HIP_TRY(hipGetLastError(), "k_eng_load launch");
    return 0;
}

static constexpr int kSmemDenseBytes = 2 * 64 * NDP_LD * 4;

extern "C" int ndp_nsfp_fwd(const float *params, const float *x, int n, float *x_out, float *act, float *tmp, void *stream) {
    if (n < 0 || !params || (n > 0 && (!x || !x_out)) || (n > 0 && !act && !tmp))
        return fail(NDP_E_INVALID, "ndp_nsfp_fwd: null pointer / negative n");
    if (!aligned16(params) || (act && !aligned16(act)) || (tmp && !aligned16(tmp)))
        return fail(NDP_E_INVALID, "ndp_nsfp_fwd: params/act/tmp must be 16-byte aligned");
    if (n == 0) return 0;
    if (int rc = set_smem((const void *)k_nsfp_dense, kSmemDenseBytes)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = (n + NDP_TILE - 1) / NDP_TILE, plane = n_tiles * NDP_TILE;
    const size_t psz = (size_t)plane * NDP_W;
    float *cur = act ? act : tmp;
    hipLaunchKernelGGL(k_nsfp_in, dim3((plane * 32 + 255) / 256), dim3(256), 0, s, params, x, n, cur, plane);
    const int grid = n_tiles < 512 ? n_tiles : 512;
    for (int l = 2; l <= NDP_NSFP_LAYERS - 1; ++l) {
        float *nxt = act ? act + (size_t)(l - 1) * psz : (cur == tmp ? tmp + psz : tmp);
        hipLaunchKernelGGL(k_nsfp_dense, dim3(grid), dim3(256), kSmemDenseBytes, s, params + ndp_nsfp_off_W(l),
                           params + ndp_nsfp_off_b(l), cur, nxt, n_tiles);
        cur = nxt;
    }
    hipLaunchKernelGGL(k_nsfp_out, dim3(n_tiles), dim3(256), 0, s, params, cur, x, n, x_out);
    HIP_TRY(hipGetLastError(), "nsfp forward launch");
    return 0;
}

extern "C" int ndp_nsfp_bwd(const float *params, const float *x, int n, float *act, const float *g,
                            float *dO_work, float *grads_part, int n_part, int p_stride, void *stream) {
    if (n <= 0 || !params || !x || !act || !g || !dO_work || !grads_part || n_part < 1)
        return fail(NDP_E_INVALID, "ndp_nsfp_bwd: null pointer / bad sizes");
    if (p_stride < ndp_nsfp_param_count()) return fail(NDP_E_INVALID, "ndp_nsfp_bwd: p_stride < P");
    if (!aligned16(params) || !aligned16(act) || !aligned16(dO_work))
        return fail(NDP_E_INVALID, "ndp_nsfp_bwd: params/act/dO_work must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    BwdJob job;
    memset(&job, 0, sizeof job);
    job.params = params; job.dO = dO_work; job.gpart = grads_part;
    job.n = n; job.n_tiles = (n + NDP_TILE - 1) / NDP_TILE; job.plane = job.n_tiles * NDP_TILE;
    const size_t psz = (size_t)job.plane * NDP_W;
    if (n_part > job.n_tiles) {                                  // partials with no tile must read as zero
        HIP_TRY(hipMemsetAsync(grads_part + (size_t)job.n_tiles * p_stride, 0,
                               sizeof(float) * (size_t)(n_part - job.n_tiles) * p_stride, s), "memset");
        n_part = job.n_tiles;
    }
    if (int rc = set_smem((const void *)k_level_bwd2, kSmemBwdBytes)) return rc;
    // output layer = a 3-row head stage folded into the layer-8 launch: dz8 = (g W9) * [h8 > 0] over plane 7 ; dW9 += g^T h8 ; db9
    ndp_layer_desc d3 = {NDP_W, 2, NDP_MOTION_SFLOW, NDP_ROT_AXIS_ANGLE, 0, 1.0f};
    const HeadCfg hc = make_head_cfg(d3);                        // nh = 3
    hipLaunchKernelGGL(k_nsfp_pack_g, dim3((job.plane + 255) / 256), dim3(256), 0, s, g, n, job.plane, dO_work);
    float *dz = act + 7 * psz;
    job.dz_plane = dz;
    // hidden layers 8..2: dW_l += dz_l^T h_{l-1} ; db_l ; dz_{l-1} = (dz_l W_l) * [h_{l-1} > 0], in place in `dz`
    // (layer 8 recomputes dz8 from dO through W9, the layers below read the dz the layer above left in the plane)
    for (int l = NDP_NSFP_LAYERS - 1; l >= 2; --l) {
        job.h_plane = act + (size_t)(l - 2) * psz;
        job.w_off = ndp_nsfp_off_W(l); job.b_off = ndp_nsfp_off_b(l);
        job.from_dO = l == NDP_NSFP_LAYERS - 1; job.wh_off = ndp_nsfp_off_W(NDP_NSFP_LAYERS); job.nh = 3;
        hipLaunchKernelGGL(k_level_bwd2, dim3(n_part), dim3(256), kSmemBwdBytes, s, hc, job, p_stride);
    }
    hipLaunchKernelGGL(k_nsfp_in_bwd, dim3(n_part), dim3(256), 0, s, dz, x, n, job.n_tiles, grads_part, p_stride);
    HIP_TRY(hipGetLastError(), "nsfp backward launch");
    return 0;
}

extern "C" int ndp_chamfer_nn_fwd(const float *x, int S, const float *y, int T,
                                  float *d2x, int *idx_x, float *d2y, int *idx_y, void *stream) {
    if (S <= 0 || T <= 0 || !x || !y || !d2x || !idx_x || !d2y || !idx_y) return fail(NDP_E_INVALID, "ndp_chamfer_nn_fwd: bad arguments");
    const int grid = (S + NN_QPB - 1) / NN_QPB + (T + NN_QPB - 1) / NN_QPB;
    hipLaunchKernelGGL(k_nn, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, S, y, T, d2x, idx_x, d2y, idx_y);
    HIP_TRY(hipGetLastError(), "k_nn launch");
    return 0;
}

extern "C" int ndp_chamfer_l1_bwd(const float *x, int S, const float *y, int T, float trunc,
                                  const float *d2x, const int *idx_x, const float *d2y, const int *idx_y,
                                  float *loss, float *gx, int point_sum, void *stream) {
    if (S <= 0 || T <= 0 || !x || !y || !d2x || !idx_x || !d2y || !idx_y || !loss) return fail(NDP_E_INVALID, "ndp_chamfer_l1_bwd: bad arguments");
    hipLaunchKernelGGL(k_chamfer_bwd, dim3((S + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, S, y, T, trunc, d2x, idx_x, d2y, idx_y, loss, gx, point_sum ? 1 : 0);
    HIP_TRY(hipGetLastError(), "k_chamfer_bwd launch");
    return 0;
}

extern "C" int ndp_landmark_mse_fwd_bwd(const float *x, const float *t, int K, float *loss, float *gx, void *stream) {
    if (K <= 0 || !x || !t || !loss) return fail(NDP_E_INVALID, "ndp_landmark_mse_fwd_bwd: bad arguments");
    hipLaunchKernelGGL(k_landmark, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, t, K, loss, gx);
    HIP_TRY(hipGetLastError(), "k_landmark launch");
    return 0;
}

extern "C" int ndp_adam_step(float *params, const float *grads, float *m, float *v, int P,
                             float w1, float b2, float w2, float neg_step, float bc2_sqrt, float eps, void *stream) {
    if (P <= 0 || !params || !grads || !m || !v) return fail(NDP_E_INVALID, "ndp_adam_step: bad arguments");
    hipLaunchKernelGGL(k_adam, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, P, w1, b2, w2, neg_step, bc2_sqrt, eps);
    HIP_TRY(hipGetLastError(), "k_adam launch");
    return 0;
}

// workgroups per pair of the split level kernels
static int engine_g8(const ndp_engine *e) { return (e->gemm_mode & 7) == 7 ? e->G : (e->G > 1 ? e->G / 2 : 1); }

// one tick = NDP_TICK_KERNELS launches; ev (optional): NDP_TICK_KERNELS + 1 events per tick recorded around them
// stages [stage_lo, stage_hi] of every tick: 0 forward, 1 nearest neighbours, 2 loss / decision / dL/dx', 3 bwd2, 4 bwd1, 5 update
static int engine_launch_ticks(const ndp_engine *e, int tick0, int n_ticks, hipStream_t s, hipEvent_t *ev, int stage_lo = 0, int stage_hi = NDP_TICK_KERNELS - 1) {
    if (int rc = check_engine(e, "ndp_engine_run")) return rc;
    const bool nn = e->w_cd != 0.f && e->t_cap > 0;
    if (nn && (!e->nn_row || !e->d2x || !e->d2y || !e->idx_x || !e->idx_y || !e->tgt))
        return fail(NDP_E_INVALID, "ndp_engine_run: Chamfer term without nearest-neighbour buffers");
    if (nn && (e->nn_mode < 0 || e->nn_mode > 2))
        return fail(NDP_E_INVALID, "ndp_engine_run: nn_mode must be 0 (one pass, vector pipe), 1 (latency shape) or 2 (one pass, matrix pipe)");
    // the column table of the one-pass vector kernel lives in LDS: only that shape has a size limit (the latency shape and
    // landmark-only engines never launch k_eng_nn)
    const int stage_x = nn1_stage_x(e->n_cap) ? 1 : 0;
    const int nn_lds = nn1_lds_floats(e->n_cap, stage_x) * 4;
    if (nn && e->nn_mode == 0) {
        if (nn_lds > 160 * 1024)
            return fail(NDP_E_UNSUPPORTED, "ndp_engine_run: n_cap too large for the one-pass nearest-neighbour kernel (nn_mode 0); use nn_mode 1");
        if (int rc = set_smem((const void *)k_eng_nn, nn_lds)) return rc;
    }
    if (int rc = set_smem((const void *)k_eng_fwd, kSmemFwdBytes)) return rc;
    if (int rc = set_smem((const void *)k_eng_bwd2, kSmemBwdBytes)) return rc;
    if (int rc = set_smem((const void *)k_eng_bwd1, kSmemBwdBytes)) return rc;
    if (nn && e->nn_mode == 2) {
        if (!nn2_fits(e->n_cap)) return fail(NDP_E_UNSUPPORTED, "ndp_engine_run: nn_mode 2 does not fit this n_cap (ndp_engine_nn_matrix_fits; the kernel walks the sources in passes of 2048, so this is not expected)");
        if (int rc = set_smem((const void *)k_eng_nn_mx, nn2_lds_floats(e->n_cap) * 4)) return rc;
        if (nn2_lds_floats(e->n_cap, 8) * 4 <= 160 * 1024) if (int rc = set_smem((const void *)k_eng_nn_mx8, nn2_lds_floats(e->n_cap, 8) * 4)) return rc;
    }
    // the matrix-pipe kernel in its 8-wave shape (512 targets per workgroup) unless gemm_mode bit 128 asks for the 4-wave one (A/B)
    const bool nn_mx8 = eng_nn_mx8(*e);
    const dim3 blk(256);
    const dim3 g_lvl(e->G, e->B);
    // bf16 kernels: one 8-wave workgroup per CU.  With all three of them on (mask 7) the engine is sized for that (G workgroups and G
    // partials per pair); in a mixed configuration they take half the fp32 grid and zero the partials they do not write.
    const dim3 g_fwd8(engine_g8(e), e->B);
    if (e->gemm_mode < 0 || e->gemm_mode > 2047) return fail(NDP_E_INVALID, "ndp_engine_run: gemm_mode is a mask of 1 (forward), 2 (bwd1), 4 (bwd2) on fp16 splits, 8 (the split forward keeps h0), 16 (bwd2 and bwd1 as two launches), 32 (the fused backward also writes dz1), 64 (the Adam step inside the fused backward), 128 / 256 / 512 / 1024 (see ndp_hip.h)");
    // both backward layers on the splits: ONE launch (k_eng_bwd_f, stage 3; stage 4 launches nothing) unless bit 16 asks for the two round-3 kernels
    // width / depth other than 128 / 3: the generic fp32 level kernels (csrc/ndp_generic.inc); gemm_mode selects nothing there
    const bool generic = gen_is_generic(e->desc);
    if (generic) {
        if (int rc = set_smem((const void *)k_eng_fwd_gen, kSmemGenFwdMax)) return rc;
        if (int rc = set_smem((const void *)k_eng_bwd_gen, kSmemGenBwdMax)) return rc;
    }
    const bool bwd_fused = !generic && (e->gemm_mode & 7) == 7 && !(e->gemm_mode & 16);   // (it reads h1 as the SPLIT forward's plane image: without bit 1 the two launches run)
    if (bwd_fused) if (int rc = set_smem((const void *)k_eng_bwd_f, kSmemBwdFBytes)) return rc;
    if (e->gemm_mode & 1) if (int rc = set_smem((const void *)k_eng_fwd8, kSmemFwd8Bytes)) return rc;
    if (e->gemm_mode & 2) if (int rc = set_smem((const void *)k_eng_bwd1_8, kSmemBwd18Bytes)) return rc;
    if (e->gemm_mode & 4) if (int rc = set_smem((const void *)k_eng_bwd2_8, kSmemBwd8Bytes)) return rc;
    const dim3 g_nn(nn1_row_chunks(e->t_cap), e->B);
    const dim3 g_nn_lat(e->n_cap / 64 + e->t_cap / 64, e->B);
    const dim3 g_upd((e->P + 255) / 256, e->B);
    const dim3 g_loss((e->n_cap + 255) / 256 + 1, e->B);   // + 1: the loss / decision workgroup
    // MEASURED VARIANT (gemm_mode bit 256, round 4): a handful of resident pairs run the whole chunk of ticks as ONE persistent launch whose
    // stages are separated by pair barriers (csrc/ndp_tick_small.inc) -- bitwise the launches, but 115 us per tick against 74 at batch 1:
    // five agent-scope release / acquire pairs per tick (the per-XCD L2s are not coherent: every release writes an L2 back) cost more
    // than six kernel boundaries.  Needs: fused split backward, latency-shape NN, one tile per workgroup, every workgroup resident.
    // one tile per level-kernel workgroup and few of them: the per-point warp rides in the forward launch (ndp_fwd_split.inc)
    const bool warp_in_fwd = (e->gemm_mode & 1) && (int)g_fwd8.x == e->n_cap / NDP_TILE && e->B * (int)g_fwd8.x <= 256;
    // everything else on the split forward: the workgroup warps its tiles' points behind its tile loop (eng_warp_tail); bit 512 of
    // gemm_mode keeps the separate k_eng_warp launch (A/B, tests)
#ifndef NDP_WARP_TAIL
#define NDP_WARP_TAIL 1
#endif
    const bool warp_tail = NDP_WARP_TAIL && (e->gemm_mode & 1) && !warp_in_fwd && !(e->gemm_mode & 512);
    const bool persistent = !ev && stage_lo == 0 && stage_hi == NDP_TICK_KERNELS - 1 && bwd_fused && (e->gemm_mode & 1) &&
                            (e->gemm_mode & 256) && !(e->gemm_mode & (32 | 64)) && (!nn || e->nn_mode == 1) && e->G == e->n_cap / NDP_TILE &&
                            e->B * e->G <= 256 && n_ticks > 0;
    // its pair barriers spin: every one of the B x G workgroups has to be RESIDENT (512 threads and ~147 KB of LDS each: one per CU).  The
    // occupancy query x the device's CU count has to cover the grid, else the per-stage launches below run instead; a second stream
    // that holds CUs (bench.py's two engines) can still delay residency -- the variant is for an engine that has the device to itself.
    bool persistent_fits = false;
    if (persistent) {
        if (int rc = set_smem((const void *)k_eng_tick_small, kSmemTickSmallBytes)) return rc;
        // (resident workgroups the device holds: queried once per device -- this is the batch-1 latency path, three HIP calls per
        //  ndp_engine_run were host time on it)
        static long long s_resident[64];                             // 0: not queried yet; -1: the query failed
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (s_resident[dev] == 0) {
                int per_cu = 0, cus = 0;
                s_resident[dev] = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_eng_tick_small, 512, kSmemTickSmallBytes) == hipSuccess &&
                                   hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0)
                                      ? (long long)per_cu * cus : -1;
            }
            persistent_fits = s_resident[dev] >= (long long)e->B * e->G;
        }
    }
    if (persistent && persistent_fits) {
        HIP_TRY(hipMemsetAsync(e->gmax + e->B, 0, sizeof(unsigned) * e->B, s), "pair barrier counters");
        hipLaunchKernelGGL(k_eng_tick_small, dim3(e->G, e->B), dim3(512), kSmemTickSmallBytes, s, *e, tick0, n_ticks);
        HIP_TRY(hipGetLastError(), "persistent tick launch");
        return 0;
    }
    for (int k = 0; k < n_ticks; ++k) {
        const int parity = (tick0 + k) & 1;
        hipEvent_t *q = ev ? ev + (size_t)k * (NDP_TICK_KERNELS + 1) : nullptr;
        int j = 0;
#define NDP_EV() do { if (q) (void)hipEventRecord(q[j++], s); } while (0)
#define NDP_ST(i) ((i) >= stage_lo && (i) <= stage_hi)
        NDP_EV();
        if (!NDP_ST(0)) {}
        else if (generic) hipLaunchKernelGGL(k_eng_fwd_gen, g_lvl, blk, gen_fwd_floats(e->desc.width) * 4, s, *e, parity);
        else if (e->gemm_mode & 1) {
            hipLaunchKernelGGL(k_eng_fwd8, g_fwd8, dim3(512), kSmemFwd8Bytes, s, *e, parity, warp_in_fwd ? 1 : (warp_tail ? 2 : 0));
            if (!warp_in_fwd && !warp_tail) hipLaunchKernelGGL(k_eng_warp, dim3((e->n_cap + 255) / 256, e->B), blk, 0, s, *e, parity);
        }
        else hipLaunchKernelGGL(k_eng_fwd, g_lvl, blk, kSmemFwdBytes, s, *e, parity);
        NDP_EV();
        if (!NDP_ST(1)) {}
        else if (nn && e->nn_mode == 1 && !(e->gemm_mode & 128) && e->B <= 2) hipLaunchKernelGGL(k_eng_nn_lat16, g_nn_lat, dim3(1024), (3 * NN_STAGE + 2 * 1024) * 4, s, *e, parity);
        else if (nn && e->nn_mode == 1 && !(e->gemm_mode & 128)) hipLaunchKernelGGL(k_eng_nn_lat8, g_nn_lat, dim3(512), (3 * NN_STAGE + 2 * 512) * 4, s, *e, parity);
        else if (nn && e->nn_mode == 1) hipLaunchKernelGGL(k_eng_nn_lat, g_nn_lat, blk, kSmemNnLatBytes, s, *e, parity);
        else if (nn_mx8) hipLaunchKernelGGL(k_eng_nn_mx8, dim3((e->t_cap + 511) / 512, e->B), dim3(512), nn2_lds_floats(e->n_cap, 8) * 4, s, *e, parity);
        else if (nn && e->nn_mode == 2) hipLaunchKernelGGL(k_eng_nn_mx, g_nn, blk, nn2_lds_floats(e->n_cap) * 4, s, *e, parity);
        else if (nn) hipLaunchKernelGGL(k_eng_nn, g_nn, blk, nn_lds, s, *e, parity, stage_x);
        NDP_EV();
        if (NDP_ST(2)) hipLaunchKernelGGL(k_eng_loss, g_loss, blk, 0, s, *e, parity);
        NDP_EV();
        if (!NDP_ST(3)) {}
        else if (generic) hipLaunchKernelGGL(k_eng_bwd_gen, g_lvl, blk, gen_bwd_floats(e->desc.width) * 4, s, *e, parity);
        else if (bwd_fused) hipLaunchKernelGGL(k_eng_bwd_f, g_fwd8, dim3(512), kSmemBwdFBytes, s, *e, parity);
        else if (e->gemm_mode & 4) hipLaunchKernelGGL(k_eng_bwd2_8, g_fwd8, dim3(512), kSmemBwd8Bytes, s, *e, parity);
        else hipLaunchKernelGGL(k_eng_bwd2, g_lvl, blk, kSmemBwdBytes, s, *e, parity);
        NDP_EV();
        if (!NDP_ST(4) || bwd_fused || generic) {}
        else if (e->gemm_mode & 2) hipLaunchKernelGGL(k_eng_bwd1_8, g_fwd8, dim3(512), kSmemBwd18Bytes, s, *e, parity);
        else hipLaunchKernelGGL(k_eng_bwd1, g_lvl, blk, kSmemBwdBytes, s, *e, parity);
        NDP_EV();
        if (!NDP_ST(5) || (bwd_fused && (e->gemm_mode & 64))) {}
        else if (bwd_fused && bf_adam_in_tail(*e)) hipLaunchKernelGGL(k_eng_update_rest, dim3((upd_rest_count(e->P) + 255) / 256, e->B), blk, 0, s, *e, parity);
        else hipLaunchKernelGGL(k_eng_update, g_upd, blk, 0, s, *e, parity);
        NDP_EV();
#undef NDP_ST
#undef NDP_EV
    }
    HIP_TRY(hipGetLastError(), "engine launch");
    return 0;
}

extern "C" int ndp_engine_nn_workspace(int n_cap, int t_cap, long long *row_floats) {
    if (n_cap < 0 || t_cap < 0 || n_cap % NDP_TILE || t_cap % NDP_TILE || !row_floats)
        return fail(NDP_E_INVALID, "ndp_engine_nn_workspace: capacities must be multiples of 64");
    *row_floats = 2LL * nn1_row_chunks(t_cap) * n_cap;          // NnPart = {float, int} per (target chunk, source)
    return 0;
}

extern "C" int ndp_chamfer_nn_onepass(const float *x, int S, const float *y, int T, float *d2x, int *idx_x, float *d2y,
                                      int *idx_y, float *ws_row, void *stream) {
    if (S <= 0 || T <= 0 || !x || !y || !d2x || !idx_x || !d2y || !idx_y || !ws_row)
        return fail(NDP_E_INVALID, "ndp_chamfer_nn_onepass: bad arguments");
    const int n_cap = (S + NDP_TILE - 1) / NDP_TILE * NDP_TILE;
    const int stage_x = nn1_stage_x(n_cap) ? 1 : 0;
    const int lds = nn1_lds_floats(n_cap, stage_x) * 4;
    if (lds > 160 * 1024) return fail(NDP_E_UNSUPPORTED, "ndp_chamfer_nn_onepass: S too large for the column table in LDS");
    if (int rc = set_smem((const void *)k_nn1, lds)) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_nn1, dim3((T + NN1_YCH - 1) / NN1_YCH), dim3(256), lds, s, x, S, y, T, n_cap, ws_row, d2y, idx_y, stage_x);
    hipLaunchKernelGGL(k_nn1_rows, dim3((S + 255) / 256), dim3(256), 0, s, S, T, n_cap, ws_row, d2x, idx_x);
    HIP_TRY(hipGetLastError(), "k_nn1 launch");
    return 0;
}

extern "C" int ndp_chamfer_nn_matrix(const float *x, int S, const float *y, int T, float *d2x, int *idx_x, float *d2y,
                                     int *idx_y, float *ws_row, void *stream) {
    if (S <= 0 || T <= 0 || !x || !y || !d2x || !idx_x || !d2y || !idx_y || !ws_row)
        return fail(NDP_E_INVALID, "ndp_chamfer_nn_matrix: bad arguments");
    const int n_cap = (S + NDP_TILE - 1) / NDP_TILE * NDP_TILE;
    if (!nn2_fits(n_cap)) return fail(NDP_E_UNSUPPORTED, "ndp_chamfer_nn_matrix: does not fit this S (ndp_engine_nn_matrix_fits; the kernel walks the sources in passes of 2048, so this is not expected)");
    const int lds = nn2_lds_floats(n_cap) * 4;
    if (int rc = set_smem((const void *)k_nn2, lds)) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_nn2, dim3((T + NN1_YCH - 1) / NN1_YCH), dim3(256), lds, s, x, S, y, T, n_cap, ws_row, d2y, idx_y);
    hipLaunchKernelGGL(k_nn1_rows, dim3((S + 255) / 256), dim3(256), 0, s, S, T, n_cap, ws_row, d2x, idx_x);
    HIP_TRY(hipGetLastError(), "k_nn2 launch");
    return 0;
}
extern "C" int ndp_engine_nn_matrix_fits(int n_cap) { return nn2_fits(n_cap) ? 1 : 0; }
extern "C" int ndp_engine_nn_onepass_fits(int n_cap) { return nn1_lds_floats(n_cap, nn1_stage_x(n_cap)) * 4 <= 160 * 1024 ? 1 : 0; }

extern "C" int ndp_engine_run(const ndp_engine *e, int tick0, int n_ticks, void *stream) {
    return engine_launch_ticks(e, tick0, n_ticks, (hipStream_t)stream, nullptr);
}

// ONE tick, only the launches of stages [stage_lo, stage_hi] (0 forward, 1 nearest neighbours, 2 loss / decision / dL/dx', 3 bwd2,
// 4 bwd1, 5 update): a test and measurement aid -- the buffers each kernel leaves behind (activations, dO, dz1, gradient partials)
// can be inspected between the stages.  Running the stages 0..5 of a tick in order, in any grouping, is ndp_engine_run(e, tick, 1).
extern "C" int ndp_engine_run_stages(const ndp_engine *e, int tick, int stage_lo, int stage_hi, void *stream) {
    if (stage_lo < 0 || stage_hi >= NDP_TICK_KERNELS || stage_lo > stage_hi) return fail(NDP_E_INVALID, "ndp_engine_run_stages: stages are 0..5, lo <= hi");
    return engine_launch_ticks(e, tick, 1, (hipStream_t)stream, nullptr, stage_lo, stage_hi);
}

// Profiling variant of ndp_engine_run: HIP events around every kernel of every tick, recorded on the launch stream;
// ms_out[NDP_TICK_KERNELS] receives the SUMMED duration of k_eng_fwd, k_eng_nn, k_eng_loss, k_eng_bwd2, k_eng_bwd1,
// k_eng_update.  Synchronises the stream before returning.  Used by bench.py for the roofline figures only.
extern "C" int ndp_engine_run_timed(const ndp_engine *e, int tick0, int n_ticks, void *stream, float *ms_out) {
    if (!e || !ms_out || n_ticks < 1 || n_ticks > 4096) return fail(NDP_E_INVALID, "ndp_engine_run_timed: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int per = NDP_TICK_KERNELS + 1;
    hipEvent_t *ev = new hipEvent_t[(size_t)n_ticks * per];
    for (int i = 0; i < n_ticks * per; ++i) (void)hipEventCreate(&ev[i]);
    int rc = engine_launch_ticks(e, tick0, n_ticks, s, ev);
    hipError_t err = hipStreamSynchronize(s);
    for (int j = 0; j < NDP_TICK_KERNELS; ++j) ms_out[j] = 0.f;
    if (rc == 0 && err == hipSuccess) {
        for (int k = 0; k < n_ticks; ++k)
            for (int j = 0; j < NDP_TICK_KERNELS; ++j) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, ev[(size_t)k * per + j], ev[(size_t)k * per + j + 1]);
                ms_out[j] += ms;
            }
    }
    for (int i = 0; i < n_ticks * per; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    if (rc) return rc;
    HIP_TRY(err, "ndp_engine_run_timed sync");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Scene-flow metrics on the device (loss.py:382-403, 431-471): per subset {all, overlap, ~overlap} the sum of the end-point
// errors and the counts behind AccS / AccR / Outlier.  One workgroup, fixed order, double accumulation.
// out[3][5] doubles: {sum err, #(err < .025 | rel < .025), #(err < .05 | rel < .05), #(rel > .3), #points}.
// ------------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(1024)
k_flow_metrics(const float *flow, const float *gt, const unsigned char *overlap, int n, double *out) {
    __shared__ double red[1024];
    double acc[3][5];
    for (int s = 0; s < 3; ++s)
        for (int k = 0; k < 5; ++k) acc[s][k] = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float d0 = flow[3 * (size_t)i] - gt[3 * (size_t)i], d1 = flow[3 * (size_t)i + 1] - gt[3 * (size_t)i + 1],
                    d2 = flow[3 * (size_t)i + 2] - gt[3 * (size_t)i + 2];
        const float g0 = gt[3 * (size_t)i], g1 = gt[3 * (size_t)i + 1], g2 = gt[3 * (size_t)i + 2];
        const float err = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
        const float rel = err / (sqrtf((g0 * g0 + g1 * g1) + g2 * g2) + 1e-20f);
        const double v[5] = {(double)err, (err < 0.025f || rel < 0.025f) ? 1.0 : 0.0, (err < 0.05f || rel < 0.05f) ? 1.0 : 0.0,
                             rel > 0.3f ? 1.0 : 0.0, 1.0};
        const int sub = overlap ? (overlap[i] ? 1 : 2) : 0;
        for (int k = 0; k < 5; ++k) {
            acc[0][k] += v[k];
            if (sub == 1) acc[1][k] += v[k];
            if (sub == 2) acc[2][k] += v[k];
        }
    }
    for (int s = 0; s < 3; ++s)
        for (int k = 0; k < 5; ++k) {
            red[threadIdx.x] = acc[s][k];
            __syncthreads();
            for (int d = 512; d > 0; d >>= 1) {
                if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
                __syncthreads();
            }
            if (threadIdx.x == 0) out[5 * s + k] = red[0];
            __syncthreads();
        }
}

extern "C" int ndp_flow_metrics(const float *flow, const float *flow_gt, const unsigned char *overlap, int n, double *out15, void *stream) {
    if (n < 0 || !out15 || (n > 0 && (!flow || !flow_gt))) return fail(NDP_E_INVALID, "ndp_flow_metrics: bad arguments");
    hipLaunchKernelGGL(k_flow_metrics, dim3(1), dim3(1024), 0, (hipStream_t)stream, flow, flow_gt, overlap, n, out15);
    HIP_TRY(hipGetLastError(), "k_flow_metrics launch");
    return 0;
}

#include "ndp_nerfies.inc"
#include "ndp_ed.inc"
