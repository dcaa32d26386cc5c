/*
 * ndp_oracle.c -- CPU ORACLE (test infrastructure only; see ndp_oracle.h).
 *
 * From-scratch restatement of the NDP hot path of rabbityl/DeformationPyramid.  Each function
 * names the reference lines it follows.  Arithmetic is IEEE float32 with the operation order
 * written out (compile with -ffp-contract=off: every fused multiply-add below is an explicit
 * fmaf); the early-stop rule runs in double like the reference's Python floats.
 *
 * Build:  make -C oracle        (gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC)
 */
#include "ndp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXW 256
#define MAXH 12

static float level_freq(int level, int k0) {
    /* nets.py:168: mul_term = 2 ** (self.m + k0) with self.m = level + 1 (nets.py:20-24) */
    return ldexpf(1.0f, level + 1 + k0);
}

/* ---------------------------------------------------------------- rotation parameterisations */

/* rigid_body.py:89-95 */
static void skew3(const float w[3], float K[9]) {
    K[0] = 0.f;   K[1] = -w[2]; K[2] = w[1];
    K[3] = w[2];  K[4] = 0.f;   K[5] = -w[0];
    K[6] = -w[1]; K[7] = w[0];  K[8] = 0.f;
}

static void mat3_mul(const float A[9], const float B[9], float C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = A[i * 3] * B[j];
            s = fmaf(A[i * 3 + 1], B[3 + j], s);
            s = fmaf(A[i * 3 + 2], B[6 + j], s);
            C[i * 3 + j] = s;
        }
}
/* C = A * B^T */
static void mat3_mul_nt(const float A[9], const float B[9], float C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = A[i * 3] * B[j * 3];
            s = fmaf(A[i * 3 + 1], B[j * 3 + 1], s);
            s = fmaf(A[i * 3 + 2], B[j * 3 + 2], s);
            C[i * 3 + j] = s;
        }
}
/* C = A^T * B */
static void mat3_mul_tn(const float A[9], const float B[9], float C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = A[i] * B[j];
            s = fmaf(A[3 + i], B[3 + j], s);
            s = fmaf(A[6 + i], B[6 + j], s);
            C[i * 3 + j] = s;
        }
}

typedef struct rot_ctx {
    float R[9];
    /* axis-angle intermediates (nets.py:150-153, rigid_body.py:113-119) */
    float theta, w[3], sn, cs, K[9];
    /* euler intermediates (rigid_body.py:19-56) */
    float Mx[9], My[9], Mz[9], A[9], se[3], ce[3];
    /* quaternion intermediates (nets.py:154-157, rigid_body.py:58-85) */
    float q[4], qd, ts;
    /* 6D intermediates (rigid_body.py:5-16) */
    float b1[3], b2[3], n1, nu, cdot;
} rot_ctx;

static void rot_fwd(int rotfmt, const float *r, rot_ctx *c) {
    if (rotfmt == NDP_ROT_AXIS_ANGLE) {
        /* theta = ||r||, w = r/theta, R = I + sin(theta) [w]x + ((1-cos(theta)) [w]x) @ [w]x */
        float t2 = r[0] * r[0];
        t2 = fmaf(r[1], r[1], t2);
        t2 = fmaf(r[2], r[2], t2);
        c->theta = sqrtf(t2);
        for (int i = 0; i < 3; ++i) c->w[i] = r[i] / c->theta;
        c->sn = sinf(c->theta);
        c->cs = cosf(c->theta);
        skew3(c->w, c->K);
        float M[9], P[9];
        float c1 = 1.0f - c->cs;
        for (int i = 0; i < 9; ++i) M[i] = c1 * c->K[i];
        mat3_mul(M, c->K, P);
        for (int i = 0; i < 9; ++i) {
            float I = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
            c->R[i] = (I + c->sn * c->K[i]) + P[i];
        }
    } else if (rotfmt == NDP_ROT_EULER) { /* R = (Mx My) Mz, convention X,Y,Z (rigid_body.py:19,56) */
        for (int i = 0; i < 3; ++i) { c->se[i] = sinf(r[i]); c->ce[i] = cosf(r[i]); }
        const float *s = c->se, *k = c->ce;
        float Mx[9] = {1, 0, 0, 0, k[0], -s[0], 0, s[0], k[0]};
        float My[9] = {k[1], 0, s[1], 0, 1, 0, -s[1], 0, k[1]};
        float Mz[9] = {k[2], -s[2], 0, s[2], k[2], 0, 0, 0, 1};
        memcpy(c->Mx, Mx, sizeof Mx); memcpy(c->My, My, sizeof My); memcpy(c->Mz, Mz, sizeof Mz);
        mat3_mul(c->Mx, c->My, c->A);
        mat3_mul(c->A, c->Mz, c->R);
    } else if (rotfmt == NDP_ROT_QUATERNION) {
        /* nets.py:155-157: s = sum(r*r); q = r / copysign(sqrt(s), r0); then quaternion_to_SO3 (rigid_body.py:64-85) */
        float s2 = r[0] * r[0];
        s2 = fmaf(r[1], r[1], s2); s2 = fmaf(r[2], r[2], s2); s2 = fmaf(r[3], r[3], s2);
        const float nrm = sqrtf(s2);
        c->qd = (r[0] < 0.f) ? -nrm : nrm;                      /* rigid_body.py:58-60 (_copysign) */
        for (int i = 0; i < 4; ++i) c->q[i] = r[i] / c->qd;
        const float qr = c->q[0], qi = c->q[1], qj = c->q[2], qk = c->q[3];
        float n2 = qr * qr;
        n2 = fmaf(qi, qi, n2); n2 = fmaf(qj, qj, n2); n2 = fmaf(qk, qk, n2);
        c->ts = 2.0f / n2;
        const float ts = c->ts;
        c->R[0] = 1.0f - ts * (qj * qj + qk * qk); c->R[1] = ts * (qi * qj - qk * qr); c->R[2] = ts * (qi * qk + qj * qr);
        c->R[3] = ts * (qi * qj + qk * qr); c->R[4] = 1.0f - ts * (qi * qi + qk * qk); c->R[5] = ts * (qj * qk - qi * qr);
        c->R[6] = ts * (qi * qk - qj * qr); c->R[7] = ts * (qj * qk + qi * qr); c->R[8] = 1.0f - ts * (qi * qi + qj * qj);
    } else { /* NDP_ROT_6D, rigid_body.py:5-16: Gram-Schmidt on (a1, a2), rows (b1, b2, b1 x b2) */
        const float *a1 = r, *a2 = r + 3;
        float n1 = a1[0] * a1[0];
        n1 = fmaf(a1[1], a1[1], n1); n1 = fmaf(a1[2], a1[2], n1);
        n1 = sqrtf(n1);
        c->n1 = n1 > 1e-12f ? n1 : 1e-12f;                       /* F.normalize eps */
        for (int i = 0; i < 3; ++i) c->b1[i] = a1[i] / c->n1;
        float cd = c->b1[0] * a2[0];
        cd = fmaf(c->b1[1], a2[1], cd); cd = fmaf(c->b1[2], a2[2], cd);
        c->cdot = cd;
        float u[3];
        for (int i = 0; i < 3; ++i) u[i] = a2[i] - cd * c->b1[i];
        float nu = u[0] * u[0];
        nu = fmaf(u[1], u[1], nu); nu = fmaf(u[2], u[2], nu);
        nu = sqrtf(nu);
        c->nu = nu > 1e-12f ? nu : 1e-12f;
        for (int i = 0; i < 3; ++i) c->b2[i] = u[i] / c->nu;
        const float *b1 = c->b1, *b2 = c->b2;
        c->R[0] = b1[0]; c->R[1] = b1[1]; c->R[2] = b1[2];
        c->R[3] = b2[0]; c->R[4] = b2[1]; c->R[5] = b2[2];
        c->R[6] = b1[1] * b2[2] - b1[2] * b2[1];
        c->R[7] = b1[2] * b2[0] - b1[0] * b2[2];
        c->R[8] = b1[0] * b2[1] - b1[1] * b2[0];
    }
}

/* G = dL/dR (row-major 3x3) -> dr = dL/d(rot parameters), chain rule op by op (what autograd does
 * on the reference's forward; avoids the cancellation of the closed-form A',B' at small theta). */
static void rot_bwd(int rotfmt, const float *r, const rot_ctx *c, const float G[9], float *dr) {
    if (rotfmt == NDP_ROT_AXIS_ANGLE) {
        const float *K = c->K;
        float c1 = 1.0f - c->cs;
        float GKt[9], MtG[9], M[9], dK[9];
        for (int i = 0; i < 9; ++i) M[i] = c1 * K[i];
        mat3_mul_nt(G, K, GKt);   /* dL/dM = G K^T   (P = M K) */
        mat3_mul_tn(M, G, MtG);   /* dL/dK (via P) = M^T G     */
        float d_sn = 0.f, d_c1 = 0.f;
        for (int i = 0; i < 9; ++i) {
            d_sn = fmaf(G[i], K[i], d_sn);
            d_c1 = fmaf(GKt[i], K[i], d_c1);
            dK[i] = fmaf(c->sn, G[i], fmaf(c1, GKt[i], MtG[i]));
        }
        float dw[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
        /* theta enters through sin, 1-cos and w = r/theta */
        float th = c->theta;
        float dth = fmaf(d_sn, c->cs, d_c1 * c->sn);
        float wdotdw = 0.f;
        for (int i = 0; i < 3; ++i) wdotdw = fmaf(dw[i], c->w[i], wdotdw);
        dth -= wdotdw / th;                      /* d(r/theta)/dtheta = -r/theta^2 = -w/theta */
        for (int i = 0; i < 3; ++i) dr[i] = fmaf(dth, c->w[i], dw[i] / th);   /* dtheta/dr = r/theta = w */
    } else if (rotfmt == NDP_ROT_EULER) {
        float dA[9], dMz[9], dMx[9], dMy[9];
        mat3_mul_nt(G, c->Mz, dA);
        mat3_mul_tn(c->A, G, dMz);
        mat3_mul_nt(dA, c->My, dMx);
        mat3_mul_tn(c->Mx, dA, dMy);
        const float *s = c->se, *k = c->ce;
        dr[0] = (dMx[7] - dMx[5]) * k[0] - (dMx[4] + dMx[8]) * s[0];
        dr[1] = (dMy[2] - dMy[6]) * k[1] - (dMy[0] + dMy[8]) * s[1];
        dr[2] = (dMz[3] - dMz[1]) * k[2] - (dMz[0] + dMz[4]) * s[2];
        (void)r;
    } else if (rotfmt == NDP_ROT_QUATERNION) {
        const float qr = c->q[0], qi = c->q[1], qj = c->q[2], qk = c->q[3], ts = c->ts;
        /* R = I-part + ts * M(q);  dL/dts = <G, M> */
        const float M[9] = {-(qj * qj + qk * qk), qi * qj - qk * qr, qi * qk + qj * qr,
                            qi * qj + qk * qr, -(qi * qi + qk * qk), qj * qk - qi * qr,
                            qi * qk - qj * qr, qj * qk + qi * qr, -(qi * qi + qj * qj)};
        float dts = 0.f;
        for (int i = 0; i < 9; ++i) dts = fmaf(G[i], M[i], dts);
        float dq[4];
        dq[0] = ts * (-qk * G[1] + qj * G[2] + qk * G[3] - qi * G[5] - qj * G[6] + qi * G[7]);
        dq[1] = ts * (qj * G[1] + qk * G[2] + qj * G[3] - 2.f * qi * G[4] - qr * G[5] + qk * G[6] + qr * G[7] - 2.f * qi * G[8]);
        dq[2] = ts * (-2.f * qj * G[0] + qi * G[1] + qr * G[2] + qi * G[3] + qk * G[5] - qr * G[6] + qk * G[7] - 2.f * qj * G[8]);
        dq[3] = ts * (-2.f * qk * G[0] - qr * G[1] + qi * G[2] + qr * G[3] - 2.f * qk * G[4] + qj * G[5] + qi * G[6] + qj * G[7]);
        /* ts = 2 / (q.q) */
        float n2 = qr * qr;
        n2 = fmaf(qi, qi, n2); n2 = fmaf(qj, qj, n2); n2 = fmaf(qk, qk, n2);
        const float dn2 = dts * (-2.0f / (n2 * n2));
        for (int i = 0; i < 4; ++i) dq[i] = fmaf(dn2, 2.0f * c->q[i], dq[i]);
        /* q = r / d, d = sgn * sqrt(r.r)  =>  dr = (dq - (dq.q) q) / d */
        float dot = 0.f;
        for (int i = 0; i < 4; ++i) dot = fmaf(dq[i], c->q[i], dot);
        for (int i = 0; i < 4; ++i) dr[i] = (dq[i] - dot * c->q[i]) / c->qd;
        (void)r;
    } else {
        const float *b1 = c->b1, *b2 = c->b2;
        const float *g1 = G, *g2 = G + 3, *g3 = G + 6;            /* dL/d(rows of R) */
        /* b3 = b1 x b2 */
        float db1[3] = {g1[0] + (b2[1] * g3[2] - b2[2] * g3[1]), g1[1] + (b2[2] * g3[0] - b2[0] * g3[2]),
                        g1[2] + (b2[0] * g3[1] - b2[1] * g3[0])};
        float db2[3] = {g2[0] + (g3[1] * b1[2] - g3[2] * b1[1]), g2[1] + (g3[2] * b1[0] - g3[0] * b1[2]),
                        g2[2] + (g3[0] * b1[1] - g3[1] * b1[0])};
        /* b2 = u / |u| */
        float d2b = 0.f;
        for (int i = 0; i < 3; ++i) d2b = fmaf(db2[i], b2[i], d2b);
        float du[3];
        for (int i = 0; i < 3; ++i) du[i] = (db2[i] - d2b * b2[i]) / c->nu;
        /* u = a2 - (b1.a2) b1 */
        float dub1 = 0.f;
        for (int i = 0; i < 3; ++i) dub1 = fmaf(du[i], b1[i], dub1);
        const float *a2 = r + 3;
        for (int i = 0; i < 3; ++i) {
            dr[3 + i] = du[i] - dub1 * b1[i];
            db1[i] = db1[i] - c->cdot * du[i] - dub1 * a2[i];
        }
        /* b1 = a1 / |a1| */
        float d1b = 0.f;
        for (int i = 0; i < 3; ++i) d1b = fmaf(db1[i], b1[i], d1b);
        for (int i = 0; i < 3; ++i) dr[i] = (db1[i] - d1b * b1[i]) / c->n1;
    }
}

/* ---------------------------------------------------------------- one point through one level */

typedef struct pt_ctx {
    float pe[6];
    float h[4][MAXW];     /* h[0] = h0, h[i] = output of hidden layer i (post-ReLU) */
    float o[MAXH];        /* scaled head outputs */
    rot_ctx rc;
    float rx[3], s, nr, xw[3];   /* R x, Sim3 scale, gate, warped-before-gate */
} pt_ctx;

static void point_fwd(const ndp_layer_desc *d, const float *P, float f, const float x[3],
                      pt_ctx *c, float out[3]) {
    const int W = d->width, NH = ndp_n_heads(d);
    /* nets.py:164-177: [sin fx, cos fx, sin fy, cos fy, sin fz, cos fz], no pi factor */
    for (int a = 0; a < 3; ++a) {
        float ph = x[a] * f;
        c->pe[2 * a] = sinf(ph);
        c->pe[2 * a + 1] = cosf(ph);
    }
    const float *W0 = P + ndp_off_W0(d), *b0 = P + ndp_off_b0(d);
    for (int o = 0; o < W; ++o) {
        float acc = b0[o];
        for (int k = 0; k < 6; ++k) acc = fmaf(W0[o * 6 + k], c->pe[k], acc);
        c->h[0][o] = acc > 0.f ? acc : 0.f;
    }
    for (int l = 1; l <= d->n_hidden; ++l) {
        const float *Wl = P + ndp_off_Wi(d, l), *bl = P + ndp_off_bi(d, l);
        for (int o = 0; o < W; ++o) {
            float acc = bl[o];
            for (int k = 0; k < W; ++k) acc = fmaf(Wl[o * W + k], c->h[l - 1][k], acc);
            c->h[l][o] = acc > 0.f ? acc : 0.f;
        }
    }
    const float *hl = c->h[d->n_hidden];
    const float *Wh = P + ndp_off_Wh(d), *bh = P + ndp_off_bh(d);
    for (int j = 0; j < NH; ++j) {
        float acc = bh[j];
        for (int k = 0; k < W; ++k) acc = fmaf(Wh[j * W + k], hl[k], acc);
        c->o[j] = d->mlp_scale * acc;                      /* nets.py:117,125,133,146 */
    }
    const float *t = c->o + ndp_head_row_trn(d);
    if (d->motion == NDP_MOTION_SFLOW) {
        for (int a = 0; a < 3; ++a) c->xw[a] = x[a] + t[a];                       /* nets.py:128-129 */
    } else {
        rot_fwd(d->rotfmt, c->o, &c->rc);
        for (int a = 0; a < 3; ++a) {
            float s = c->rc.R[a * 3] * x[0];
            s = fmaf(c->rc.R[a * 3 + 1], x[1], s);
            s = fmaf(c->rc.R[a * 3 + 2], x[2], s);
            c->rx[a] = s;
        }
        if (d->motion == NDP_MOTION_SIM3) {
            c->s = c->o[ndp_head_row_scale(d)] + 1.0f;                            /* nets.py:125 */
            for (int a = 0; a < 3; ++a) c->xw[a] = fmaf(c->s, c->rx[a], t[a]);   /* nets.py:126 */
        } else {
            for (int a = 0; a < 3; ++a) c->xw[a] = c->rx[a] + t[a];               /* nets.py:121 */
        }
    }
    if (d->nonrigidity) {                                                         /* nets.py:132-135 */
        c->nr = 1.0f / (1.0f + expf(-c->o[ndp_head_row_nr(d)]));
        for (int a = 0; a < 3; ++a) out[a] = fmaf(c->nr, c->xw[a] - x[a], x[a]);
    } else {
        for (int a = 0; a < 3; ++a) out[a] = c->xw[a];
    }
}

/* backward of one point; grads (P floats) accumulated */
static void point_bwd(const ndp_layer_desc *d, const float *P, const float x[3], const pt_ctx *c,
                      const float g_in[3], float g_nr, float *grads) {
    const int W = d->width, NH = ndp_n_heads(d);
    float g[3] = {g_in[0], g_in[1], g_in[2]};
    float d_o[MAXH];
    for (int j = 0; j < NH; ++j) d_o[j] = 0.f;
    if (d->nonrigidity) {
        float dnr = g_nr;
        for (int a = 0; a < 3; ++a) dnr = fmaf(g[a], c->xw[a] - x[a], dnr);
        d_o[ndp_head_row_nr(d)] = dnr * (c->nr * (1.0f - c->nr));
        for (int a = 0; a < 3; ++a) g[a] *= c->nr;
    }
    float *d_t = d_o + ndp_head_row_trn(d);
    for (int a = 0; a < 3; ++a) d_t[a] = g[a];
    if (d->motion != NDP_MOTION_SFLOW) {
        float grx[3] = {g[0], g[1], g[2]};
        if (d->motion == NDP_MOTION_SIM3) {
            float ds = 0.f;
            for (int a = 0; a < 3; ++a) ds = fmaf(g[a], c->rx[a], ds);
            d_o[ndp_head_row_scale(d)] = ds;
            for (int a = 0; a < 3; ++a) grx[a] = g[a] * c->s;
        }
        float G[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) G[a * 3 + b] = grx[a] * x[b];
        rot_bwd(d->rotfmt, c->o, &c->rc, G, d_o);
    }
    /* heads: o_j = mlp_scale * (Wh_j . h + bh_j) */
    const float *hl = c->h[d->n_hidden];
    const float *Wh = P + ndp_off_Wh(d);
    float *gWh = grads + ndp_off_Wh(d), *gbh = grads + ndp_off_bh(d);
    float dh[MAXW], dz[MAXW];
    for (int k = 0; k < W; ++k) dh[k] = 0.f;
    for (int j = 0; j < NH; ++j) {
        float dj = d->mlp_scale * d_o[j];
        gbh[j] += dj;
        for (int k = 0; k < W; ++k) {
            gWh[j * W + k] = fmaf(dj, hl[k], gWh[j * W + k]);
            dh[k] = fmaf(dj, Wh[j * W + k], dh[k]);
        }
    }
    for (int l = d->n_hidden; l >= 1; --l) {
        const float *Wl = P + ndp_off_Wi(d, l);
        float *gWl = grads + ndp_off_Wi(d, l), *gbl = grads + ndp_off_bi(d, l);
        const float *hin = c->h[l - 1], *hout = c->h[l];
        for (int o = 0; o < W; ++o) dz[o] = hout[o] > 0.f ? dh[o] : 0.f;
        for (int k = 0; k < W; ++k) dh[k] = 0.f;
        for (int o = 0; o < W; ++o) {
            float z = dz[o];
            gbl[o] += z;
            if (z == 0.f) continue;   /* exact: fma(0, finite, acc) == acc */
            for (int k = 0; k < W; ++k) {
                gWl[o * W + k] = fmaf(z, hin[k], gWl[o * W + k]);
                dh[k] = fmaf(z, Wl[o * W + k], dh[k]);
            }
        }
    }
    float *gW0 = grads + ndp_off_W0(d), *gb0 = grads + ndp_off_b0(d);
    for (int o = 0; o < W; ++o) {
        float z = c->h[0][o] > 0.f ? dh[o] : 0.f;
        gb0[o] += z;
        for (int k = 0; k < 6; ++k) gW0[o * 6 + k] = fmaf(z, c->pe[k], gW0[o * 6 + k]);
    }
}

/* ---------------------------------------------------------------- public: level fwd / bwd */

void ndp_o_level_fwd(const ndp_layer_desc *d, const float *params, int level, int k0,
                     const float *x, int n, float *x_out, float *nonrig_out, int nthreads) {
    const float f = level_freq(level, k0);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        pt_ctx *c = (pt_ctx *)malloc(sizeof(pt_ctx));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int i = 0; i < n; ++i) {
            point_fwd(d, params, f, x + 3 * i, c, x_out + 3 * i);
            if (nonrig_out) nonrig_out[i] = d->nonrigidity ? c->nr : 0.f;
        }
        free(c);
    }
    (void)nthreads;
}

void ndp_o_level_bwd(const ndp_layer_desc *d, const float *params, int level, int k0,
                     const float *x, int n, const float *g, const float *g_nr,
                     float *grads, int nthreads) {
    const float f = level_freq(level, k0);
    const int P = ndp_param_count(d);
    int nt = nthreads > 0 ? nthreads : 1;
#ifndef _OPENMP
    nt = 1;
#endif
    float *part = (float *)calloc((size_t)nt * P, sizeof(float));
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        pt_ctx *c = (pt_ctx *)malloc(sizeof(pt_ctx));
        float *gl = part + (size_t)tid * P;
        float tmp[3];
        /* contiguous chunks in thread order -> deterministic for a fixed thread count */
        int lo = (int)((long long)n * tid / nt), hi = (int)((long long)n * (tid + 1) / nt);
        for (int i = lo; i < hi; ++i) {
            point_fwd(d, params, f, x + 3 * i, c, tmp);
            point_bwd(d, params, x + 3 * i, c, g + 3 * i, g_nr ? g_nr[i] : 0.f, gl);
        }
        free(c);
    }
    for (int k = 0; k < P; ++k) {
        float s = part[k];
        for (int t = 1; t < nt; ++t) s += part[(size_t)t * P + k];
        grads[k] = s;
    }
    free(part);
}

void ndp_o_pyramid_fwd(const ndp_layer_desc *descs, int m, int k0, const float *params_all,
                       const float *x, int n, float *x_out, int nthreads) {
    float *a = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    memcpy(a, x, sizeof(float) * 3 * (size_t)n);
    size_t off = 0;
    for (int l = 0; l < m; ++l) {
        ndp_o_level_fwd(&descs[l], params_all + off, l, k0, a, n, x_out, NULL, nthreads);
        memcpy(a, x_out, sizeof(float) * 3 * (size_t)n);
        off += (size_t)ndp_param_count(&descs[l]);
    }
    if (m == 0) memcpy(x_out, x, sizeof(float) * 3 * (size_t)n);
    free(a);
}

/* ---------------------------------------------------------------- Chamfer / landmarks */

static void nn_search(const float *q, int nq, const float *r, int nr, float *d2, int *idx, int nthreads) {
    /* pytorch3d knn_points(K=1) semantics (call sites loss.py:177-178): exact brute force,
     * squared L2 accumulated as dist = fma(diff, diff, dist) over x,y,z; first minimum kept. */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int i = 0; i < nq; ++i) {
        float best = INFINITY;
        int bi = -1;
        const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
        for (int j = 0; j < nr; ++j) {
            float dx = qx - r[3 * j], dy = qy - r[3 * j + 1], dz = qz - r[3 * j + 2];
            float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (dd < best) { best = dd; bi = j; }
        }
        d2[i] = best;
        idx[i] = bi;
    }
    (void)nthreads;
}

float ndp_o_chamfer(const float *x, int S, const float *y, int T, float trunc,
                    float *d2x, int *idx_x, float *d2y, int *idx_y, float *gx, int nthreads) {
    nn_search(x, S, y, T, d2x, idx_x, nthreads);
    nn_search(y, T, x, S, d2y, idx_y, nthreads);
    /* loss.py:185-188 (>= trunc -> 0), :227-228 (sqrt, sum), :233-235 (/ lengths), :255 (x + y) */
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < S; ++i) sx += (d2x[i] >= trunc) ? 0.f : sqrtf(d2x[i]);
    for (int j = 0; j < T; ++j) sy += (d2y[j] >= trunc) ? 0.f : sqrtf(d2y[j]);
    float loss = sx / (float)S + sy / (float)T;
    if (gx) {
        /* d/dx_i: (x_i - y_nn(i)) / (S ||.||)  +  sum_{j: nn'(j)=i} (x_i - y_j) / (T ||.||), j ascending */
        for (int i = 0; i < S; ++i) {
            float gi[3] = {0.f, 0.f, 0.f};
            if (!(d2x[i] >= trunc)) {
                const float *yy = y + 3 * idx_x[i];
                float inv = 1.0f / ((float)S * sqrtf(d2x[i]));
                for (int a = 0; a < 3; ++a) gi[a] = (x[3 * i + a] - yy[a]) * inv;
            }
            for (int a = 0; a < 3; ++a) gx[3 * i + a] = gi[a];
        }
        for (int j = 0; j < T; ++j) {
            if (d2y[j] >= trunc) continue;
            int i = idx_y[j];
            float inv = 1.0f / ((float)T * sqrtf(d2y[j]));
            for (int a = 0; a < 3; ++a) gx[3 * i + a] = fmaf(x[3 * i + a] - y[3 * j + a], inv, gx[3 * i + a]);
        }
    }
    return loss;
}

float ndp_o_landmark(const float *x, const float *t, int K, float *gx) {
    /* registration.py:203: torch.mean(torch.sum((warped - tgt)**2, dim=-1)) */
    float s = 0.f;
    const float invK = 1.0f / (float)K;
    for (int k = 0; k < K; ++k) {
        float e0 = x[3 * k] - t[3 * k], e1 = x[3 * k + 1] - t[3 * k + 1], e2 = x[3 * k + 2] - t[3 * k + 2];
        s += fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
        if (gx) { gx[3 * k] = 2.0f * e0 * invK; gx[3 * k + 1] = 2.0f * e1 * invK; gx[3 * k + 2] = 2.0f * e2 * invK; }
    }
    return s * invK;
}

/* ---------------------------------------------------------------- Adam / early stop */

void ndp_o_adam(float *p, const float *g, float *m, float *v, int P, int t,
                double lr, double b1, double b2, double eps) {
    /* torch/optim/adam.py _single_tensor_adam (torch 2.10, CPU defaults: no foreach/fused,
     * amsgrad False, weight_decay 0), Python-double scalars cast to float at each tensor op. */
    const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
    const float w1 = (float)(1.0 - b1);            /* exp_avg.lerp_(grad, 1 - beta1)              */
    const float fb2 = (float)b2, w2 = (float)(1.0 - b2);
    const float neg_step = (float)(-(lr / bc1));   /* param.addcdiv_(exp_avg, denom, value=-step_size) */
    const float bc2s = (float)sqrt(bc2), feps = (float)eps;
    for (int i = 0; i < P; ++i) {
        float gi = g[i];
        float mi = m[i] + w1 * (gi - m[i]);        /* lerp, weight < 0.5 branch                    */
        float vi = v[i] * fb2;                     /* exp_avg_sq.mul_(beta2)                       */
        vi = vi + (w2 * gi) * gi;                  /* .addcmul_(grad, grad, value=1-beta2)         */
        float denom = sqrtf(vi) / bc2s + feps;     /* (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps) */
        p[i] = p[i] + (neg_step * mi) / denom;
        m[i] = mi;
        v[i] = vi;
    }
}

int ndp_o_stop_check(double loss, int *break_counter, double *loss_prev,
                     int max_break_count, double break_threshold_ratio) {
    /* registration.py:226-232 */
    if (loss < 1e-4) return 1;
    if (fabs(*loss_prev - loss) < *loss_prev * break_threshold_ratio) *break_counter += 1;
    if (*break_counter >= max_break_count) return 1;
    *loss_prev = loss;
    return 0;
}

/* ---------------------------------------------------------------- the level loop */

int ndp_o_optimize(const ndp_layer_desc *descs, const ndp_o_opt_cfg *cfg, float *params_all,
                   float *pts, int K, int S, const float *ldmk_t, const float *tgt, int T,
                   int *iters_per_level, double *loss_trace, int trace_cap, int nthreads) {
    const int n = K + S;
    const int use_cd = (S > 0) && (K == 0 || cfg->w_cd > 0.f);
    int maxP = 0;
    for (int l = 0; l < cfg->m; ++l) { int p = ndp_param_count(&descs[l]); if (p > maxP) maxP = p; }
    float *warped = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    float *g = (float *)calloc(3 * (size_t)(n > 0 ? n : 1), sizeof(float));
    float *gcd = (float *)malloc(sizeof(float) * 3 * (size_t)(S > 0 ? S : 1));
    float *d2x = (float *)malloc(sizeof(float) * (size_t)(S > 0 ? S : 1));
    float *d2y = (float *)malloc(sizeof(float) * (size_t)(T > 0 ? T : 1));
    int *ix = (int *)malloc(sizeof(int) * (size_t)(S > 0 ? S : 1));
    int *iy = (int *)malloc(sizeof(int) * (size_t)(T > 0 ? T : 1));
    float *grads = (float *)malloc(sizeof(float) * (size_t)maxP);
    float *am = (float *)malloc(sizeof(float) * (size_t)maxP);
    float *av = (float *)malloc(sizeof(float) * (size_t)maxP);
    float *nrv = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float *gnr = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int total_steps = 0, ntrace = 0;
    size_t off = 0;
    for (int level = 0; level < cfg->m; ++level) {
        const ndp_layer_desc *d = &descs[level];
        const int P = ndp_param_count(d);
        float *params = params_all + off;
        memset(am, 0, sizeof(float) * (size_t)P);    /* fresh Adam per level: registration.py:176 */
        memset(av, 0, sizeof(float) * (size_t)P);
        int break_counter = 0, t = 0, evals = 0;
        double loss_prev = 1e6;                       /* registration.py:179-180 */
        for (int it = 0; it < cfg->iters; ++it) {
            const int use_reg = cfg->w_reg > 0.f && level > 0 && d->nonrigidity;          /* :216 */
            ndp_o_level_fwd(d, params, level, cfg->k0, pts, n, warped, use_reg ? nrv : NULL, nthreads);   /* :208 */
            float loss = 0.f;
            if (K > 0) {
                loss = ndp_o_landmark(warped, ldmk_t, K, g);                               /* :193,:203 */
            }
            if (use_cd) {
                float lcd = ndp_o_chamfer(warped + 3 * K, S, tgt, T, cfg->trunc, d2x, ix, d2y, iy, gcd, nthreads);
                if (K > 0) {
                    loss = loss + cfg->w_cd * lcd;                                         /* :197 */
                    for (int i = 0; i < 3 * S; ++i) g[3 * K + i] = cfg->w_cd * gcd[i];
                } else {
                    loss = lcd;                                                            /* :212 */
                    memcpy(g, gcd, sizeof(float) * 3 * (size_t)S);
                }
            } else if (S > 0) {
                memset(g + 3 * K, 0, sizeof(float) * 3 * (size_t)S);
            }
            if (use_reg) {
                /* registration.py:216-220: loss += w_reg * BCELoss(nonrigidity, 0), mean over all warped points;
                 * torch clamps log at -100 and the backward divides by max((1-x) x, 1e-12)                    */
                float acc = 0.f;
                const float invn = 1.0f / (float)n;
                for (int i = 0; i < n; ++i) {
                    float l1 = logf(1.0f - nrv[i]);
                    if (l1 < -100.0f) l1 = -100.0f;
                    acc += -l1;
                    const float den = (1.0f - nrv[i]) * nrv[i];
                    gnr[i] = cfg->w_reg * (invn * (nrv[i] / (den > 1e-12f ? den : 1e-12f)));
                }
                loss = loss + cfg->w_reg * (acc * invn);
            }
            ++evals;
            if (loss_trace && ntrace < trace_cap) loss_trace[ntrace++] = (double)loss;
            if (cfg->early_stop &&
                ndp_o_stop_check((double)loss, &break_counter, &loss_prev, cfg->max_break_count,
                                 cfg->break_threshold_ratio))
                break;
            ndp_o_level_bwd(d, params, level, cfg->k0, pts, n, g, use_reg ? gnr : NULL, grads, nthreads); /* :236 */
            ndp_o_adam(params, grads, am, av, P, ++t, cfg->lr, 0.9, 0.999, 1e-8);          /* :237 */
            ++total_steps;
        }
        /* registration.py:242-249: next level starts from this level's (last forward) output.
         * In landmark-only mode the samples are not warped (w_cd == 0 keeps s_sample, :245-246). */
        if (K > 0) memcpy(pts, warped, sizeof(float) * 3 * (size_t)K);
        if (use_cd) memcpy(pts + 3 * K, warped + 3 * K, sizeof(float) * 3 * (size_t)S);
        if (iters_per_level) iters_per_level[level] = evals;
        off += (size_t)P;
    }
    free(warped); free(g); free(gcd); free(d2x); free(d2y); free(ix); free(iy);
    free(grads); free(am); free(av); free(nrv); free(gnr);
    return total_steps;
}

/* ---------------------------------------------------------------- NSFP baseline
 * Neural_Prior (nets.py:256-292): x -> L9(relu(L8(... relu(L1 x)))) ; optimize_neural_SFlow (registration.py:470-540). */

static void nsfp_point_fwd(const float *P, const float x[3], float h[8][NDP_NSFP_W], float flow[3]) {
    const float *W1 = P + ndp_nsfp_off_W(1), *b1 = P + ndp_nsfp_off_b(1);
    for (int o = 0; o < NDP_NSFP_W; ++o) {
        float z = fmaf(W1[3 * o + 2], x[2], fmaf(W1[3 * o + 1], x[1], fmaf(W1[3 * o], x[0], b1[o])));
        h[0][o] = z > 0.f ? z : 0.f;
    }
    for (int l = 2; l <= NDP_NSFP_LAYERS - 1; ++l) {
        const float *W = P + ndp_nsfp_off_W(l), *b = P + ndp_nsfp_off_b(l);
        for (int o = 0; o < NDP_NSFP_W; ++o) {
            float z = b[o];
            for (int k = 0; k < NDP_NSFP_W; ++k) z = fmaf(W[o * NDP_NSFP_W + k], h[l - 2][k], z);
            h[l - 1][o] = z > 0.f ? z : 0.f;
        }
    }
    const float *W9 = P + ndp_nsfp_off_W(NDP_NSFP_LAYERS), *b9 = P + ndp_nsfp_off_b(NDP_NSFP_LAYERS);
    for (int j = 0; j < 3; ++j) {
        float z = b9[j];
        for (int k = 0; k < NDP_NSFP_W; ++k) z = fmaf(W9[j * NDP_NSFP_W + k], h[7][k], z);
        flow[j] = z;
    }
}

void ndp_o_nsfp_fwd(const float *params, const float *x, int n, float *x_out, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int p = 0; p < n; ++p) {
        float h[8][NDP_NSFP_W], flow[3];
        nsfp_point_fwd(params, x + 3 * p, h, flow);
        for (int a = 0; a < 3; ++a) x_out[3 * p + a] = x[3 * p + a] + flow[a];       /* registration.py:507 */
    }
}

/* grads (P floats, overwritten) of a scalar loss given g = dL/dx_out [n][3]; points in index order */
void ndp_o_nsfp_bwd(const float *params, const float *x, int n, const float *g, float *grads) {
    const int P = ndp_nsfp_param_count();
    memset(grads, 0, sizeof(float) * (size_t)P);
    for (int p = 0; p < n; ++p) {
        float h[8][NDP_NSFP_W], flow[3], dz[NDP_NSFP_W], dzn[NDP_NSFP_W];
        nsfp_point_fwd(params, x + 3 * p, h, flow);
        const float *gp = g + 3 * p;
        const float *W9 = params + ndp_nsfp_off_W(NDP_NSFP_LAYERS);
        float *gW9 = grads + ndp_nsfp_off_W(NDP_NSFP_LAYERS), *gb9 = grads + ndp_nsfp_off_b(NDP_NSFP_LAYERS);
        for (int j = 0; j < 3; ++j) {
            gb9[j] += gp[j];
            for (int k = 0; k < NDP_NSFP_W; ++k) gW9[j * NDP_NSFP_W + k] = fmaf(gp[j], h[7][k], gW9[j * NDP_NSFP_W + k]);
        }
        for (int k = 0; k < NDP_NSFP_W; ++k) {
            float s = 0.f;
            for (int j = 0; j < 3; ++j) s = fmaf(gp[j], W9[j * NDP_NSFP_W + k], s);
            dz[k] = h[7][k] > 0.f ? s : 0.f;
        }
        for (int l = NDP_NSFP_LAYERS - 1; l >= 2; --l) {
            const float *W = params + ndp_nsfp_off_W(l);
            float *gW = grads + ndp_nsfp_off_W(l), *gb = grads + ndp_nsfp_off_b(l);
            for (int o = 0; o < NDP_NSFP_W; ++o) {
                gb[o] += dz[o];
                for (int k = 0; k < NDP_NSFP_W; ++k) gW[o * NDP_NSFP_W + k] = fmaf(dz[o], h[l - 2][k], gW[o * NDP_NSFP_W + k]);
            }
            for (int k = 0; k < NDP_NSFP_W; ++k) {
                float s = 0.f;
                for (int o = 0; o < NDP_NSFP_W; ++o) s = fmaf(dz[o], W[o * NDP_NSFP_W + k], s);
                dzn[k] = h[l - 2][k] > 0.f ? s : 0.f;
            }
            memcpy(dz, dzn, sizeof dz);
        }
        float *gW1 = grads + ndp_nsfp_off_W(1), *gb1 = grads + ndp_nsfp_off_b(1);
        for (int o = 0; o < NDP_NSFP_W; ++o) {
            gb1[o] += dz[o];
            for (int a = 0; a < 3; ++a) gW1[3 * o + a] = fmaf(dz[o], x[3 * p + a], gW1[3 * o + a]);
        }
    }
}

/* registration.py:504-529.  params updated in place; returns the number of Adam steps; loss_trace gets every
 * evaluated loss; warped [S][3] = s_sample + flow of the LAST forward.                                          */
int ndp_o_nsfp_optimize(float *params, const float *s_sample, int S, const float *t_sample, int T, int iters,
                        int max_break_count, double ratio, double lr, int early_stop,
                        float *warped, double *loss_trace, int trace_cap, int nthreads) {
    const int P = ndp_nsfp_param_count();
    float *g = (float *)malloc(sizeof(float) * 3 * (size_t)S), *grads = (float *)malloc(sizeof(float) * (size_t)P);
    float *am = (float *)calloc((size_t)P, sizeof(float)), *av = (float *)calloc((size_t)P, sizeof(float));
    float *d2x = (float *)malloc(sizeof(float) * (size_t)S), *d2y = (float *)malloc(sizeof(float) * (size_t)T);
    int *ix = (int *)malloc(sizeof(int) * (size_t)S), *iy = (int *)malloc(sizeof(int) * (size_t)T);
    int break_counter = 0, steps = 0, ntrace = 0;
    double loss_prev = 1e6;
    for (int i = 0; i < iters; ++i) {
        ndp_o_nsfp_fwd(params, s_sample, S, warped, nthreads);
        const float loss = ndp_o_chamfer(warped, S, t_sample, T, 1e9f, d2x, ix, d2y, iy, g, nthreads);
        if (loss_trace && ntrace < trace_cap) loss_trace[ntrace++] = (double)loss;
        if (early_stop && ndp_o_stop_check((double)loss, &break_counter, &loss_prev, max_break_count, ratio)) break;
        ndp_o_nsfp_bwd(params, s_sample, S, g, grads);
        ndp_o_adam(params, grads, am, av, P, ++steps, lr, 0.9, 0.999, 1e-8);
    }
    free(g); free(grads); free(am); free(av); free(d2x); free(d2y); free(ix); free(iy);
    return steps;
}
