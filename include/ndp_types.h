/*
 * ndp_types.h -- plain-C description of one Neural-Deformation-Pyramid (NDP) level and of the
 * flat parameter layout shared by the HIP library (include/ndp_hip.h) and the CPU oracle
 * (oracle/ndp_oracle.h).  No torch / HIP types appear here.
 *
 * One pyramid level is the reference's NDPLayer (model/nets.py:65-184):
 *     pe  = [sin f x0, cos f x0, sin f x1, cos f x1, sin f x2, cos f x2],  f = 2^(level+1+k0)   (nets.py:164-177)
 *     h0  = relu(W0 pe + b0)                   W0 [W][6]                                         (nets.py:75,114)
 *     h_i = relu(W_i h_{i-1} + b_i)            W_i [W][W], i = 1..n_hidden (= depth-1)           (nets.py:295-304)
 *     head outputs o_j = mlp_scale * (Wh_j . h + bh_j)                                           (nets.py:117,125,146,133)
 *     warp by motion type                                                                        (nets.py:119-135)
 *
 * Flat layout of one level's parameters (float32, P = ndp_param_count() values):
 *     [ W0 (W*6) | b0 (W) | W1 (W*W) | b1 (W) | ... | Wh (NH*W) | bh (NH) ]
 * Head rows, in the reference's module-registration order (nets.py:82-102):
 *     rot rows (n_rot = 3 axis_angle/euler, 4 quaternion, 6 "6D"; none for sflow),
 *     scale row (Sim3 only), 3 translation rows, nonrigidity row (if enabled).
 * For W=128, depth=3: SE3/axis-angle P = 34694, Sim3 P = 34823.
 */
#ifndef NDP_TYPES_H
#define NDP_TYPES_H

#if defined(__HIPCC__)
#define NDP_HD __host__ __device__
#else
#define NDP_HD
#endif

#ifdef __cplusplus
extern "C" {
#endif

enum { NDP_MOTION_SE3 = 0, NDP_MOTION_SIM3 = 1, NDP_MOTION_SFLOW = 2 };
enum { NDP_ROT_AXIS_ANGLE = 0, NDP_ROT_EULER = 1, NDP_ROT_QUATERNION = 2, NDP_ROT_6D = 3 };

typedef struct ndp_layer_desc {
    int width;        /* W: hidden width (reference default 128)                         */
    int n_hidden;     /* depth-1: number of W x W layers (reference default 2)            */
    int motion;       /* NDP_MOTION_*   (config motion_type, nets.py:17)                  */
    int rotfmt;       /* NDP_ROT_*      (config rotation_format, nets.py:84-89)           */
    int nonrigidity;  /* 1 if the level carries the nr_branch gate (nets.py:100-103)      */
    float mlp_scale;  /* 0.001 (nets.py:107)                                              */
} ndp_layer_desc;

NDP_HD static inline int ndp_n_rot(const ndp_layer_desc *d) {
    if (d->motion == NDP_MOTION_SFLOW) return 0;
    return d->rotfmt == NDP_ROT_QUATERNION ? 4 : (d->rotfmt == NDP_ROT_6D ? 6 : 3);
}
NDP_HD static inline int ndp_n_heads(const ndp_layer_desc *d) {
    return ndp_n_rot(d) + (d->motion == NDP_MOTION_SIM3 ? 1 : 0) + 3 + (d->nonrigidity ? 1 : 0);
}
NDP_HD static inline int ndp_head_row_scale(const ndp_layer_desc *d) { return ndp_n_rot(d); }
NDP_HD static inline int ndp_head_row_trn(const ndp_layer_desc *d) {
    return ndp_n_rot(d) + (d->motion == NDP_MOTION_SIM3 ? 1 : 0);
}
NDP_HD static inline int ndp_head_row_nr(const ndp_layer_desc *d) { return ndp_head_row_trn(d) + 3; }

/* offsets (in floats) inside one level's flat parameter block */
NDP_HD static inline int ndp_off_W0(const ndp_layer_desc *d) { (void)d; return 0; }
NDP_HD static inline int ndp_off_b0(const ndp_layer_desc *d) { return d->width * 6; }
NDP_HD static inline int ndp_off_Wi(const ndp_layer_desc *d, int i /*1-based*/) {
    return d->width * 7 + (i - 1) * (d->width * d->width + d->width);
}
NDP_HD static inline int ndp_off_bi(const ndp_layer_desc *d, int i) { return ndp_off_Wi(d, i) + d->width * d->width; }
NDP_HD static inline int ndp_off_Wh(const ndp_layer_desc *d) { return ndp_off_Wi(d, d->n_hidden + 1); }
NDP_HD static inline int ndp_off_bh(const ndp_layer_desc *d) { return ndp_off_Wh(d) + ndp_n_heads(d) * d->width; }
NDP_HD static inline int ndp_param_count(const ndp_layer_desc *d) { return ndp_off_bh(d) + ndp_n_heads(d); }

/* ---- Neural scene-flow prior baseline (/root/reference/model/nets.py:256-292) ----------------
 * flow(x) = L9(relu(L8(... relu(L1(x)) ...))), L1: 3 -> 128, L2..L8: 128 -> 128, L9: 128 -> 3.  Flat layout, layers in
 * construction order, weight (row-major [out][in]) then bias:
 *     [ W1 (128*3) | b1 (128) | W2 (128*128) | b2 (128) | ... | W8 | b8 | W9 (3*128) | b9 (3) ]         P = 116 483      */
#define NDP_NSFP_W 128
#define NDP_NSFP_LAYERS 9
NDP_HD static inline int ndp_nsfp_off_W(int l /*1..9*/) {
    return l == 1 ? 0 : NDP_NSFP_W * 4 + (l - 2) * (NDP_NSFP_W * NDP_NSFP_W + NDP_NSFP_W);
}
NDP_HD static inline int ndp_nsfp_off_b(int l) {
    return ndp_nsfp_off_W(l) + (l == 1 ? NDP_NSFP_W * 3 : (l == NDP_NSFP_LAYERS ? 3 * NDP_NSFP_W : NDP_NSFP_W * NDP_NSFP_W));
}
NDP_HD static inline int ndp_nsfp_param_count(void) { return ndp_nsfp_off_b(NDP_NSFP_LAYERS) + 3; }

#ifdef __cplusplus
}
#endif
#endif /* NDP_TYPES_H */
