// ndp_device.h -- device-side helpers shared by the NDP kernels (gfx950 only).
//
// Per-point head math: rotation parameterisations and the SE3 / Sim3 / sflow warp with their
// backward, written op-for-op like the CPU oracle (oracle/ndp_oracle.c) so that the two agree
// to the last bits the transcendental functions allow.  The file is compiled with
// -ffp-contract=off: every fused multiply-add is an explicit fmaf.
//
// Reference: model/nets.py:111-161 (NDPLayer.forward, get_Rotation), model/rigid_body.py:19-56
// (euler_to_SO3), :89-119 (skew, exp_so3).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ndp_hip.h"

#define NDP_W      128          // hidden width the kernels are specialised for
#define NDP_LD     132          // LDS row stride (floats) of a [64][128] tile: +4 pad => b128 reads conflict-free
#define NDP_NHMAX  16           // head-output slots per point (rot.., scale, trn, nr; <= 11 used)
#define NDP_HROW   24           // row stride of the saved per-point record: 16 head outputs + 6 posenc values + 2 pad

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct HeadCfg {
    int motion, rotfmt, n_rot, row_scale, row_trn, nh;
    float mlp_scale;
};

__host__ __device__ inline HeadCfg make_head_cfg(const ndp_layer_desc &d) {
    HeadCfg h;
    h.motion = d.motion;
    h.rotfmt = d.rotfmt;
    h.n_rot = ndp_n_rot(&d);
    h.row_scale = ndp_head_row_scale(&d);
    h.row_trn = ndp_head_row_trn(&d);
    h.nh = ndp_n_heads(&d);
    h.mlp_scale = d.mlp_scale;
    return h;
}

__device__ __forceinline__ void mat3_mul(const float *A, const float *B, float *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = A[i * 3] * B[j];
            s = fmaf(A[i * 3 + 1], B[3 + j], s);
            s = fmaf(A[i * 3 + 2], B[6 + j], s);
            C[i * 3 + j] = s;
        }
}
__device__ __forceinline__ void mat3_mul_nt(const float *A, const float *B, float *C) {   // A * B^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = A[i * 3] * B[j * 3];
            s = fmaf(A[i * 3 + 1], B[j * 3 + 1], s);
            s = fmaf(A[i * 3 + 2], B[j * 3 + 2], s);
            C[i * 3 + j] = s;
        }
}
__device__ __forceinline__ void mat3_mul_tn(const float *A, const float *B, float *C) {   // A^T * B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = A[i] * B[j];
            s = fmaf(A[3 + i], B[3 + j], s);
            s = fmaf(A[6 + i], B[6 + j], s);
            C[i * 3 + j] = s;
        }
}

// Everything the backward needs from the forward of one point's head stage.
struct PointHead {
    float R[9];
    float theta, w[3], sn, cs, K[9];                 // axis-angle
    float Mx[9], My[9], Mz[9], A[9], se[3], ce[3];   // euler
    float rx[3], s;
};

__device__ __forceinline__ void rot_fwd(int rotfmt, const float *r, PointHead &c) {
    if (rotfmt == NDP_ROT_AXIS_ANGLE) {
        // nets.py:150-153 + rigid_body.py:113-119
        float t2 = r[0] * r[0];
        t2 = fmaf(r[1], r[1], t2);
        t2 = fmaf(r[2], r[2], t2);
        c.theta = sqrtf(t2);
#pragma unroll
        for (int i = 0; i < 3; ++i) c.w[i] = r[i] / c.theta;
        c.sn = sinf(c.theta);
        c.cs = cosf(c.theta);
        c.K[0] = 0.f;      c.K[1] = -c.w[2]; c.K[2] = c.w[1];
        c.K[3] = c.w[2];   c.K[4] = 0.f;     c.K[5] = -c.w[0];
        c.K[6] = -c.w[1];  c.K[7] = c.w[0];  c.K[8] = 0.f;
        float M[9], P[9];
        const float c1 = 1.0f - c.cs;
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = c1 * c.K[i];
        mat3_mul(M, c.K, P);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float I = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
            c.R[i] = (I + c.sn * c.K[i]) + P[i];
        }
    } else {
        // rigid_body.py:19-56, convention X,Y,Z: R = (Mx My) Mz
#pragma unroll
        for (int i = 0; i < 3; ++i) { c.se[i] = sinf(r[i]); c.ce[i] = cosf(r[i]); }
        const float *s = c.se, *k = c.ce;
        c.Mx[0] = 1; c.Mx[1] = 0; c.Mx[2] = 0; c.Mx[3] = 0; c.Mx[4] = k[0]; c.Mx[5] = -s[0]; c.Mx[6] = 0; c.Mx[7] = s[0]; c.Mx[8] = k[0];
        c.My[0] = k[1]; c.My[1] = 0; c.My[2] = s[1]; c.My[3] = 0; c.My[4] = 1; c.My[5] = 0; c.My[6] = -s[1]; c.My[7] = 0; c.My[8] = k[1];
        c.Mz[0] = k[2]; c.Mz[1] = -s[2]; c.Mz[2] = 0; c.Mz[3] = s[2]; c.Mz[4] = k[2]; c.Mz[5] = 0; c.Mz[6] = 0; c.Mz[7] = 0; c.Mz[8] = 1;
        mat3_mul(c.Mx, c.My, c.A);
        mat3_mul(c.A, c.Mz, c.R);
    }
}

__device__ __forceinline__ void rot_bwd(int rotfmt, const PointHead &c, const float *G, float *dr) {
    if (rotfmt == NDP_ROT_AXIS_ANGLE) {
        const float *K = c.K;
        const float c1 = 1.0f - c.cs;
        float GKt[9], MtG[9], M[9], dK[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = c1 * K[i];
        mat3_mul_nt(G, K, GKt);
        mat3_mul_tn(M, G, MtG);
        float d_sn = 0.f, d_c1 = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            d_sn = fmaf(G[i], K[i], d_sn);
            d_c1 = fmaf(GKt[i], K[i], d_c1);
            dK[i] = fmaf(c.sn, G[i], fmaf(c1, GKt[i], MtG[i]));
        }
        const float dw[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
        const float th = c.theta;
        float dth = fmaf(d_sn, c.cs, d_c1 * c.sn);
        float wdotdw = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) wdotdw = fmaf(dw[i], c.w[i], wdotdw);
        dth -= wdotdw / th;
#pragma unroll
        for (int i = 0; i < 3; ++i) dr[i] = fmaf(dth, c.w[i], dw[i] / th);
    } else {
        float dA[9], dMz[9], dMx[9], dMy[9];
        mat3_mul_nt(G, c.Mz, dA);
        mat3_mul_tn(c.A, G, dMz);
        mat3_mul_nt(dA, c.My, dMx);
        mat3_mul_tn(c.Mx, dA, dMy);
        const float *s = c.se, *k = c.ce;
        dr[0] = (dMx[7] - dMx[5]) * k[0] - (dMx[4] + dMx[8]) * s[0];
        dr[1] = (dMy[2] - dMy[6]) * k[1] - (dMy[0] + dMy[8]) * s[1];
        dr[2] = (dMz[3] - dMz[1]) * k[2] - (dMz[0] + dMz[4]) * s[2];
    }
}

// o: scaled head outputs (rot.., scale, trn).  x -> out.   nets.py:117-129
__device__ __forceinline__ void head_warp_fwd(const HeadCfg &hc, const float *o, const float *x,
                                              PointHead &c, float *out) {
    const float *t = o + hc.row_trn;
    if (hc.motion == NDP_MOTION_SFLOW) {
#pragma unroll
        for (int a = 0; a < 3; ++a) out[a] = x[a] + t[a];
        return;
    }
    rot_fwd(hc.rotfmt, o, c);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float s = c.R[a * 3] * x[0];
        s = fmaf(c.R[a * 3 + 1], x[1], s);
        s = fmaf(c.R[a * 3 + 2], x[2], s);
        c.rx[a] = s;
    }
    if (hc.motion == NDP_MOTION_SIM3) {
        c.s = o[hc.row_scale] + 1.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) out[a] = fmaf(c.s, c.rx[a], t[a]);
    } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) out[a] = c.rx[a] + t[a];
    }
}

// g = dL/dout -> d_o = dL/d(scaled head outputs).  d_o points at NDP_NHMAX floats in LDS (rows are
// addressed with run-time offsets, which registers cannot do); unused rows are zeroed.
__device__ __forceinline__ void head_warp_bwd(const HeadCfg &hc, const float *x, const PointHead &c,
                                              const float *g, float *d_o) {
#pragma unroll
    for (int j = 0; j < NDP_NHMAX; j += 4) *reinterpret_cast<float4 *>(d_o + j) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 3; ++a) d_o[hc.row_trn + a] = g[a];
    if (hc.motion == NDP_MOTION_SFLOW) return;
    float grx[3] = {g[0], g[1], g[2]};
    if (hc.motion == NDP_MOTION_SIM3) {
        float ds = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) ds = fmaf(g[a], c.rx[a], ds);
        d_o[hc.row_scale] = ds;
#pragma unroll
        for (int a = 0; a < 3; ++a) grx[a] = g[a] * c.s;
    }
    float G[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) G[a * 3 + b] = grx[a] * x[b];
    float dr[3];
    rot_bwd(hc.rotfmt, c, G, dr);
#pragma unroll
    for (int a = 0; a < 3; ++a) d_o[a] = dr[a];
}

// deterministic block-wide sum (256 threads); every thread gets the result
__device__ __forceinline__ float block_sum_256(float v, float *scratch /* >= 256 floats */) {
    const int t = threadIdx.x;
    scratch[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) scratch[t] = scratch[t] + scratch[t + s];
        __syncthreads();
    }
    const float r = scratch[0];
    __syncthreads();
    return r;
}

// Per-point head backward for the level kernels: recompute the head stage from the saved scaled head
// outputs, push g = dL/dx_out through it, and emit dO = mlp_scale * dL/d(scaled outputs) (16 floats).
// lds_row: NDP_NHMAX floats of LDS private to the calling thread (run-time row offsets live there).
__device__ __forceinline__ void point_head_bwd(const HeadCfg &hc, const float *heads_row /*global, NDP_HROW*/,
                                               const float *x, const float *g, float *lds_row, float *dO_row /*global*/) {
#pragma unroll
    for (int j = 0; j < NDP_NHMAX; j += 4)
        *reinterpret_cast<float4 *>(lds_row + j) = *reinterpret_cast<const float4 *>(heads_row + j);
    PointHead c;
    float out[3];
    head_warp_fwd(hc, lds_row, x, c, out);
    head_warp_bwd(hc, x, c, g, lds_row);
#pragma unroll
    for (int j = 0; j < NDP_NHMAX; j += 4) {
        float4 v = *reinterpret_cast<const float4 *>(lds_row + j);
        v.x *= hc.mlp_scale; v.y *= hc.mlp_scale; v.z *= hc.mlp_scale; v.w *= hc.mlp_scale;
        *reinterpret_cast<float4 *>(dO_row + j) = v;
    }
}
