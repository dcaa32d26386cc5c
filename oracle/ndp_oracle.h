/*
 * ndp_oracle.h -- CPU ORACLE for the NDP per-pair optimisation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a from-scratch plain-C restatement of the reference's
 * algorithm (rabbityl/DeformationPyramid: model/nets.py, model/loss.py, model/rigid_body.py,
 * model/registration.py:126-262).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (deformationpyramid_amd/) never does.
 *
 * Pinning: checked against golden vectors captured from the reference itself (run in the build
 * container by tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.  The nearest
 * neighbour arithmetic of the reference lives in pytorch3d (un-vendored, un-pinned upstream);
 * its published semantics (exact brute-force K=1, squared L2) are restated in ndp_o_chamfer().
 */
#ifndef NDP_ORACLE_H
#define NDP_ORACLE_H

#include "../include/ndp_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One level forward on n points: x [n][3] -> x_out [n][3].  level is 0-based; f = 2^(level+1+k0).
 * nonrig_out (may be NULL): sigmoid gate per point when desc->nonrigidity.      (nets.py:111-140) */
void ndp_o_level_fwd(const ndp_layer_desc *d, const float *params, int level, int k0,
                     const float *x, int n, float *x_out, float *nonrig_out, int nthreads);

/* Backward of one level: given g = dL/dx_out [n][3] (and g_nr = dL/dnonrig [n] or NULL),
 * accumulate dL/dparams into grads (P floats, overwritten).  x is a detached input
 * (registration.py:243-249), so no dL/dx is produced.                                          */
void ndp_o_level_bwd(const ndp_layer_desc *d, const float *params, int level, int k0,
                     const float *x, int n, const float *g, const float *g_nr,
                     float *grads, int nthreads);

/* Whole pyramid, levels 0..m-1 chained, forward only (registration.py:254-255; nets.py:36-48).
 * params_all: the m levels' blocks back to back (level l at offset sum_{j<l} P_j).            */
void ndp_o_pyramid_fwd(const ndp_layer_desc *descs, int m, int k0, const float *params_all,
                       const float *x, int n, float *x_out, int nthreads);

/* Truncated L1 Chamfer on 1-NN (loss.py:94-258): exact brute-force nearest neighbours in both
 * directions (lowest index wins ties), d2 = fma(dz,dz,fma(dy,dy,dx*dx)); entries with d2 >= trunc
 * contribute 0; loss = sum_i sqrt(d2x_i)/S + sum_j sqrt(d2y_j)/T.  gx (may be NULL) receives
 * dL/dx [S][3].  Returns the loss.                                                               */
float ndp_o_chamfer(const float *x, int S, const float *y, int T, float trunc,
                    float *d2x, int *idx_x, float *d2y, int *idx_y, float *gx, int nthreads);

/* Landmark loss mean_k sum_xyz (x_k - t_k)^2 and its gradient 2 (x_k - t_k)/K  (registration.py:201-203). */
float ndp_o_landmark(const float *x, const float *t, int K, float *gx);

/* torch.optim.Adam single-tensor step (lr, betas (0.9,0.999) by default, eps 1e-8, no weight
 * decay), step count t = 1,2,...   (registration.py:176,237)                                    */
void ndp_o_adam(float *p, const float *g, float *m, float *v, int P, int t,
                double lr, double b1, double b2, double eps);

/* Early-stop rule of registration.py:226-232 evaluated on a Python-float view of the fp32 loss.
 * state: break_counter and loss_prev (reset to 0 / 1e6 by the caller at each level start).
 * Returns 1 if the level loop must break BEFORE the backward of this iteration.                 */
int ndp_o_stop_check(double loss, int *break_counter, double *loss_prev,
                     int max_break_count, double break_threshold_ratio);

typedef struct ndp_o_opt_cfg {
    int m, k0;
    int iters;                 /* config.iters                                   */
    int max_break_count;       /* config.max_break_count                         */
    double break_threshold_ratio;
    double lr;
    float w_cd;                /* weight on the Chamfer term (1 when no landmarks) */
    float trunc;               /* truncation in squared units (1e9 = off)        */
    float w_reg;               /* nonrigidity BCE weight (0 = off)               */
    int early_stop;            /* 0 disables the three stop tests (fixed-work runs) */
} ndp_o_opt_cfg;

/* The level loop + Adam loop + early stop of optimize_deformation_pyramid (registration.py:170-249).
 * pts [K+S][3]: landmarks first (K may be 0), then Chamfer samples (S may be 0); overwritten with
 * the points warped through every optimised level.  ldmk_t [K][3], tgt [T][3].
 * params_all is updated in place.  iters_per_level[m] = loss evaluations per level;
 * loss_trace (cap entries, may be NULL) receives every evaluated loss.  Returns total Adam steps. */
int ndp_o_optimize(const ndp_layer_desc *descs, const ndp_o_opt_cfg *cfg, float *params_all,
                   float *pts, int K, int S, const float *ldmk_t, const float *tgt, int T,
                   int *iters_per_level, double *loss_trace, int trace_cap, int nthreads);

/* ---- NSFP baseline: Neural_Prior (nets.py:256-292), optimize_neural_SFlow (registration.py:470-540).
 * Parameter layout: ndp_types.h (ndp_nsfp_off_W / ndp_nsfp_off_b).  x_out = x + MLP(x).                                  */
void ndp_o_nsfp_fwd(const float *params, const float *x, int n, float *x_out, int nthreads);
void ndp_o_nsfp_bwd(const float *params, const float *x, int n, const float *g, float *grads);
int ndp_o_nsfp_optimize(float *params, const float *s_sample, int S, const float *t_sample, int T, int iters,
                        int max_break_count, double ratio, double lr, int early_stop,
                        float *warped, double *loss_trace, int trace_cap, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
