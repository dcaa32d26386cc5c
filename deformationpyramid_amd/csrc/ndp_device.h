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
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HeadCfg {
    int motion, rotfmt, n_rot, row_scale, row_trn, nh;
    int nonrig, row_nr;          // nonrigidity gate head present (nets.py:100-103) and its row
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
    h.nonrig = d.nonrigidity ? 1 : 0;
    h.row_nr = ndp_head_row_nr(&d);
    h.mlp_scale = d.mlp_scale;
    return h;
}

// Engine / pyramid descriptors carry nonrigidity = 1 to mean "every level but the first has the gate"
// (Deformation_Pyramid builds level i with nonrigidity_est & (i != 0), nets.py:26).
__host__ __device__ inline ndp_layer_desc desc_at_level(const ndp_layer_desc &d, int level) {
    ndp_layer_desc r = d;
    r.nonrigidity = (d.nonrigidity && level > 0) ? 1 : 0;
    return r;
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
    float q[4], qd, ts;                              // quaternion
    float b1[3], b2[3], n1, nu, cdot;                // 6D
    float rx[3], s;
    float nr, xw[3];                                 // gate value and the warp before gating
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
    } else if (rotfmt == NDP_ROT_EULER) {
        // rigid_body.py:19-56, convention X,Y,Z: R = (Mx My) Mz
#pragma unroll
        for (int i = 0; i < 3; ++i) { c.se[i] = sinf(r[i]); c.ce[i] = cosf(r[i]); }
        const float *s = c.se, *k = c.ce;
        c.Mx[0] = 1; c.Mx[1] = 0; c.Mx[2] = 0; c.Mx[3] = 0; c.Mx[4] = k[0]; c.Mx[5] = -s[0]; c.Mx[6] = 0; c.Mx[7] = s[0]; c.Mx[8] = k[0];
        c.My[0] = k[1]; c.My[1] = 0; c.My[2] = s[1]; c.My[3] = 0; c.My[4] = 1; c.My[5] = 0; c.My[6] = -s[1]; c.My[7] = 0; c.My[8] = k[1];
        c.Mz[0] = k[2]; c.Mz[1] = -s[2]; c.Mz[2] = 0; c.Mz[3] = s[2]; c.Mz[4] = k[2]; c.Mz[5] = 0; c.Mz[6] = 0; c.Mz[7] = 0; c.Mz[8] = 1;
        mat3_mul(c.Mx, c.My, c.A);
        mat3_mul(c.A, c.Mz, c.R);
    } else if (rotfmt == NDP_ROT_QUATERNION) {
        // nets.py:155-157 + rigid_body.py:58-85
        float s2 = r[0] * r[0];
        s2 = fmaf(r[1], r[1], s2); s2 = fmaf(r[2], r[2], s2); s2 = fmaf(r[3], r[3], s2);
        const float nrm = sqrtf(s2);
        c.qd = (r[0] < 0.f) ? -nrm : nrm;
#pragma unroll
        for (int i = 0; i < 4; ++i) c.q[i] = r[i] / c.qd;
        const float qr = c.q[0], qi = c.q[1], qj = c.q[2], qk = c.q[3];
        float n2 = qr * qr;
        n2 = fmaf(qi, qi, n2); n2 = fmaf(qj, qj, n2); n2 = fmaf(qk, qk, n2);
        c.ts = 2.0f / n2;
        const float ts = c.ts;
        c.R[0] = 1.0f - ts * (qj * qj + qk * qk); c.R[1] = ts * (qi * qj - qk * qr); c.R[2] = ts * (qi * qk + qj * qr);
        c.R[3] = ts * (qi * qj + qk * qr); c.R[4] = 1.0f - ts * (qi * qi + qk * qk); c.R[5] = ts * (qj * qk - qi * qr);
        c.R[6] = ts * (qi * qk - qj * qr); c.R[7] = ts * (qj * qk + qi * qr); c.R[8] = 1.0f - ts * (qi * qi + qj * qj);
    } else {
        // rigid_body.py:5-16 (6D): Gram-Schmidt on (a1, a2), rows (b1, b2, b1 x b2)
        const float *a1 = r, *a2 = r + 3;
        float n1 = a1[0] * a1[0];
        n1 = fmaf(a1[1], a1[1], n1); n1 = fmaf(a1[2], a1[2], n1);
        n1 = sqrtf(n1);
        c.n1 = n1 > 1e-12f ? n1 : 1e-12f;
#pragma unroll
        for (int i = 0; i < 3; ++i) c.b1[i] = a1[i] / c.n1;
        float cd = c.b1[0] * a2[0];
        cd = fmaf(c.b1[1], a2[1], cd); cd = fmaf(c.b1[2], a2[2], cd);
        c.cdot = cd;
        float u[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) u[i] = a2[i] - cd * c.b1[i];
        float nu = u[0] * u[0];
        nu = fmaf(u[1], u[1], nu); nu = fmaf(u[2], u[2], nu);
        nu = sqrtf(nu);
        c.nu = nu > 1e-12f ? nu : 1e-12f;
#pragma unroll
        for (int i = 0; i < 3; ++i) c.b2[i] = u[i] / c.nu;
        const float *b1 = c.b1, *b2 = c.b2;
        c.R[0] = b1[0]; c.R[1] = b1[1]; c.R[2] = b1[2];
        c.R[3] = b2[0]; c.R[4] = b2[1]; c.R[5] = b2[2];
        c.R[6] = b1[1] * b2[2] - b1[2] * b2[1];
        c.R[7] = b1[2] * b2[0] - b1[0] * b2[2];
        c.R[8] = b1[0] * b2[1] - b1[1] * b2[0];
    }
}

__device__ __forceinline__ void rot_bwd(int rotfmt, const float *r, const PointHead &c, const float *G, float *dr) {
    const float *a2in = r + 3;       // 6D only: the raw second vector
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
    } else if (rotfmt == NDP_ROT_EULER) {
        float dA[9], dMz[9], dMx[9], dMy[9];
        mat3_mul_nt(G, c.Mz, dA);
        mat3_mul_tn(c.A, G, dMz);
        mat3_mul_nt(dA, c.My, dMx);
        mat3_mul_tn(c.Mx, dA, dMy);
        const float *s = c.se, *k = c.ce;
        dr[0] = (dMx[7] - dMx[5]) * k[0] - (dMx[4] + dMx[8]) * s[0];
        dr[1] = (dMy[2] - dMy[6]) * k[1] - (dMy[0] + dMy[8]) * s[1];
        dr[2] = (dMz[3] - dMz[1]) * k[2] - (dMz[0] + dMz[4]) * s[2];
    } else if (rotfmt == NDP_ROT_QUATERNION) {
        const float qr = c.q[0], qi = c.q[1], qj = c.q[2], qk = c.q[3], ts = c.ts;
        const float M[9] = {-(qj * qj + qk * qk), qi * qj - qk * qr, qi * qk + qj * qr,
                            qi * qj + qk * qr, -(qi * qi + qk * qk), qj * qk - qi * qr,
                            qi * qk - qj * qr, qj * qk + qi * qr, -(qi * qi + qj * qj)};
        float dts = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) dts = fmaf(G[i], M[i], dts);
        float dq[4];
        dq[0] = ts * (-qk * G[1] + qj * G[2] + qk * G[3] - qi * G[5] - qj * G[6] + qi * G[7]);
        dq[1] = ts * (qj * G[1] + qk * G[2] + qj * G[3] - 2.f * qi * G[4] - qr * G[5] + qk * G[6] + qr * G[7] - 2.f * qi * G[8]);
        dq[2] = ts * (-2.f * qj * G[0] + qi * G[1] + qr * G[2] + qi * G[3] + qk * G[5] - qr * G[6] + qk * G[7] - 2.f * qj * G[8]);
        dq[3] = ts * (-2.f * qk * G[0] - qr * G[1] + qi * G[2] + qr * G[3] - 2.f * qk * G[4] + qj * G[5] + qi * G[6] + qj * G[7]);
        float n2 = qr * qr;
        n2 = fmaf(qi, qi, n2); n2 = fmaf(qj, qj, n2); n2 = fmaf(qk, qk, n2);
        const float dn2 = dts * (-2.0f / (n2 * n2));
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[i] = fmaf(dn2, 2.0f * c.q[i], dq[i]);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) dot = fmaf(dq[i], c.q[i], dot);
#pragma unroll
        for (int i = 0; i < 4; ++i) dr[i] = (dq[i] - dot * c.q[i]) / c.qd;
    } else {
        const float *b1 = c.b1, *b2 = c.b2;
        const float *g1 = G, *g2 = G + 3, *g3 = G + 6;
        float db1[3] = {g1[0] + (b2[1] * g3[2] - b2[2] * g3[1]), g1[1] + (b2[2] * g3[0] - b2[0] * g3[2]),
                        g1[2] + (b2[0] * g3[1] - b2[1] * g3[0])};
        float db2[3] = {g2[0] + (g3[1] * b1[2] - g3[2] * b1[1]), g2[1] + (g3[2] * b1[0] - g3[0] * b1[2]),
                        g2[2] + (g3[0] * b1[1] - g3[1] * b1[0])};
        float d2b = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) d2b = fmaf(db2[i], b2[i], d2b);
        float du[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) du[i] = (db2[i] - d2b * b2[i]) / c.nu;
        float dub1 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) dub1 = fmaf(du[i], b1[i], dub1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dr[3 + i] = du[i] - dub1 * b1[i];
            db1[i] = db1[i] - c.cdot * du[i] - dub1 * a2in[i];
        }
        float d1b = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) d1b = fmaf(db1[i], b1[i], d1b);
#pragma unroll
        for (int i = 0; i < 3; ++i) dr[i] = (db1[i] - d1b * b1[i]) / c.n1;
    }
}

// o: scaled head outputs (rot.., scale, trn, nr).  x -> out.   nets.py:117-135
__device__ __forceinline__ void head_warp_fwd(const HeadCfg &hc, const float *o, const float *x,
                                              PointHead &c, float *out) {
    const float *t = o + hc.row_trn;
    if (hc.motion == NDP_MOTION_SFLOW) {
#pragma unroll
        for (int a = 0; a < 3; ++a) c.xw[a] = x[a] + t[a];
    } else {
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
            for (int a = 0; a < 3; ++a) c.xw[a] = fmaf(c.s, c.rx[a], t[a]);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) c.xw[a] = c.rx[a] + t[a];
        }
    }
    if (hc.nonrig) {                                  // nets.py:132-135
        c.nr = 1.0f / (1.0f + expf(-o[hc.row_nr]));
#pragma unroll
        for (int a = 0; a < 3; ++a) out[a] = fmaf(c.nr, c.xw[a] - x[a], x[a]);
    } else {
        c.nr = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) out[a] = c.xw[a];
    }
}

// g = dL/dout -> d_o = dL/d(scaled head outputs).  d_o points at NDP_NHMAX floats in LDS (rows are
// addressed with run-time offsets, which registers cannot do); unused rows are zeroed.
__device__ __forceinline__ void head_warp_bwd(const HeadCfg &hc, const float *x, const PointHead &c,
                                              const float *g_in, float g_nr, float *d_o) {
    // the 6D backward needs the raw rot outputs: read them before the row is reused for the gradient
    float rraw[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) rraw[a] = d_o[a];
#pragma unroll
    for (int j = 0; j < NDP_NHMAX; j += 4) *reinterpret_cast<float4 *>(d_o + j) = make_float4(0.f, 0.f, 0.f, 0.f);
    float g[3] = {g_in[0], g_in[1], g_in[2]};
    if (hc.nonrig) {
        float dnr = g_nr;
#pragma unroll
        for (int a = 0; a < 3; ++a) dnr = fmaf(g[a], c.xw[a] - x[a], dnr);
        d_o[hc.row_nr] = dnr * (c.nr * (1.0f - c.nr));
#pragma unroll
        for (int a = 0; a < 3; ++a) g[a] *= c.nr;
    }
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
    float dr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    rot_bwd(hc.rotfmt, rraw, c, G, dr);
#pragma unroll
    for (int a = 0; a < 6; ++a)
        if (a < hc.n_rot) d_o[a] = dr[a];
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
                                               const float *x, const float *g, float g_nr, float *lds_row,
                                               float *dO_row /*global*/, float *amax = nullptr /* max |dO| of the row */) {
    if (heads_row) {                                 // (nullptr: the caller has already brought the row into lds_row)
#pragma unroll
        for (int j = 0; j < NDP_NHMAX; j += 4)
            *reinterpret_cast<float4 *>(lds_row + j) = *reinterpret_cast<const float4 *>(heads_row + j);
    }
    PointHead c;
    float out[3];
    head_warp_fwd(hc, lds_row, x, c, out);
    head_warp_bwd(hc, x, c, g, g_nr, lds_row);
#pragma unroll
    for (int j = 0; j < NDP_NHMAX; j += 4) {
        float4 v = *reinterpret_cast<const float4 *>(lds_row + j);
        v.x *= hc.mlp_scale; v.y *= hc.mlp_scale; v.z *= hc.mlp_scale; v.w *= hc.mlp_scale;
        *reinterpret_cast<float4 *>(dO_row + j) = v;
        if (amax) *amax = fmaxf(fmaxf(*amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
}
