/*
 * ndp_hip.h -- C ABI of libndp_hip.so, the MI355X (gfx950) implementation of the NDP per-pair
 * optimisation hot path.  Plain pointers and sizes only: every `float*`/`int*` below is a DEVICE
 * pointer to contiguous memory owned by the caller, `stream` is a hipStream_t passed as void*,
 * nothing is allocated inside, there is no global state, and every entry point returns
 *     0  success,   <0  invalid argument (NDP_E_*),   >0  a hipError_t from the launch.
 *
 * The reference has no C ABI: its "operator API" for this path is the Python surface of
 * model/registration.py, model/nets.py and model/loss.py.  Each entry point names the reference
 * code it replaces; INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Parameter layout of one level: include/ndp_types.h.  Points are float32 [n][3] row-major.
 * width = 128, depth = 3 (both shipped configs: NDP.yaml:24-25, LNDP.yaml:45-46) run on the MFMA kernels.  Every other
 * 1 <= width <= 256, 1 <= depth <= 4 the YAML can name (the reference builds any: nets.py:75,295-304) runs behind the SAME
 * entry points on generic fp32 kernels (csrc/ndp_generic.inc: the oracle's chains on the fp32 matrix instruction, about a third
 * of the 128 / 3 kernels' rate per layer; gemm_mode and the `split` entries select nothing there).  Their activation store is
 * [depth][n_cap][width] fp32 rows wherever this header says [3][n_cap][128].  Shapes beyond return NDP_E_UNSUPPORTED.
 */
#ifndef NDP_HIP_H
#define NDP_HIP_H

#include "ndp_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NDP_E_INVALID     (-1)
#define NDP_E_UNSUPPORTED (-2)
#define NDP_MAX_LEVELS    16
#define NDP_TILE          64      /* points per tile; point capacities are multiples of this */
#define NDP_HROW          24      /* floats per point in the saved head record                 */

int ndp_version(void);                 /* 100*major + minor */
const char *ndp_last_error(void);      /* text of the last non-zero return on this thread */
/* Digest of the sources + flags the library was built from (the loader compares it with the tree and rebuilds or
 * refuses a stale library), and sizeof of the six structs that cross the ABI by value, in declaration order:
 * ndp_layer_desc, ndp_pair_geom, ndp_pair_state, ndp_engine, ndp_warp_job, ndp_load_job (the binding checks its mirror). */
const char *ndp_build_id(void);
int ndp_abi_sizes(int *out6);

/* ------------------------------------------------------------------ single-pair operators */

/* NDPLayer.forward for one level on n points (nets.py:111-140; posenc :164-177; MLP :295-304;
 * get_Rotation :144-161; rigid_body.py:19-56,89-119).
 *   x [n][3] -> x_out [n][3].
 *   act  (may be NULL): [3][n_cap][128] saved post-ReLU activations h0,h1,h2 for ndp_level_bwd ([depth][n_cap][width] in general)
 *   heads(may be NULL): [n_cap][NDP_HROW] per-point record: 16 scaled head outputs (rot.., scale,
 *                       trn, nr) followed by the 6 positional-encoding values and 2 pad floats
 *   nonrig_out (may be NULL): [n] gate values sigmoid(0.001 nr_branch(h)) when desc->nonrigidity
 *                       (the second element NDPLayer.forward returns, nets.py:133-140)
 * n_cap = n rounded up to NDP_TILE (row count of act/heads).                                   */
int ndp_level_fwd(const ndp_layer_desc *desc, const float *params, int level, int k0,
                  const float *x, int n, float *x_out, float *act, float *heads, float *nonrig_out,
                  void *stream);

/* Backward of one level wrt its parameters given g = dL/dx_out [n][3] (autograd of nets.py:111-140;
 * x is a detached input, registration.py:243-249).  act/heads come from ndp_level_fwd on the same
 * x and params; act is CONSUMED at 128 / 3 (its h2 plane is overwritten with an intermediate; the generic kernels leave it).  dO_work is
 * scratch of [n_cap][16] floats; g_nr (may be NULL) = dL/d(gate) [n] when desc->nonrigidity.  grads_part [n_part][P_stride] receives n_part partial sums
 * (deterministic: workgroup g sums tiles g, g+n_part, ...); ndp_grad_reduce folds them in index order. */
int ndp_level_bwd(const ndp_layer_desc *desc, const float *params, int level, int k0,
                  const float *x, int n, float *act, const float *heads, const float *g, const float *g_nr,
                  float *dO_work, float *grads_part, int n_part, int p_stride, void *stream);

/* grads[P] = sum_{g<n_part} grads_part[g][:]  (fixed order). */
int ndp_grad_reduce(const float *grads_part, int n_part, int p_stride, int P, float *grads, void *stream);

/* Whole pyramid forward, levels 0..m-1 (Deformation_Pyramid.warp, nets.py:36-48; the final
 * inference warp of registration.py:254-255).  params_all: level l at params_all + l*p_stride.
 * desc->nonrigidity = 1 means "every level but the first carries the gate" (nets.py:26).        */
int ndp_pyramid_fwd(const ndp_layer_desc *desc, int m, int k0, const float *params_all, int p_stride,
                    const float *x, int n, float *x_out, void *stream);

/* Batched, single-launch form of the final inference warp: every job is one cloud pushed through all m
 * levels inside ONE kernel (grid = tiles x jobs; a workgroup keeps its 64 points in LDS from level to
 * level, so there is no intermediate HBM traffic and one launch fills the chip with several pairs).
 * Folds the centring and the un-centring of registration.py:150-153 / :258:
 *     x_out = pyramid(x - shift_in) + shift_out          (either shift may be NULL).
 * `jobs` is a HOST array (copied into the kernel arguments); all jobs share desc / m / k0 / p_stride. */
typedef struct ndp_warp_job {
    const float *params;             /* [m][p_stride] */
    const float *x;                  /* [n][3] */
    float *x_out;                    /* [n][3] (may alias x) */
    const float *shift_in;           /* [3] device, or NULL */
    const float *shift_out;          /* [3] device, or NULL */
    int n, pad;
} ndp_warp_job;
#define NDP_MAX_WARP_JOBS 32
int ndp_pyramid_fwd_batch(const ndp_layer_desc *desc, int m, int k0, int p_stride,
                          const ndp_warp_job *jobs, int n_jobs, void *stream);
/* The same warp with the engine's split arithmetic (ndp_engine.gemm_mode & 1): 128-wide contractions and layer 0 as two-way fp16
 * splits in the one-accumulator form (2^6 x = hi + lo, three partial products into one fp32 accumulator) on the fp16 MFMA; 256 points
 * per workgroup carried through all m levels in LDS.  fp32-level accuracy (within 1e-5 of ndp_pyramid_fwd_batch on warped
 * coordinates), not bitwise the fma chain; the 2^6 pre-scale means an activation or weight beyond about 1023 (65504 / 64)
 * saturates (layer 0's pre-activation at 65504).                                                                                  */
int ndp_pyramid_fwd_batch_split(const ndp_layer_desc *desc, int m, int k0, int p_stride,
                                const ndp_warp_job *jobs, int n_jobs, void *stream);
/* the same with `tiles` (1..8) 64-point tiles per workgroup instead of 4: fewer weight prologues per cloud, longer workgroups (the
   batched engine's throughput shape; identical results) */
int ndp_pyramid_fwd_batch_split_tiles(const ndp_layer_desc *desc, int m, int k0, int p_stride,
                                      const ndp_warp_job *jobs, int n_jobs, int tiles, void *stream);

/* Per-cloud means (registration.py:150-153: src_pcd.mean(dim=0), tgt_pcd.mean(dim=0)); means[0..2] = source,
 * means[4..6] = target (means[3], means[7] = 0).  Accumulated in double in a fixed order, rounded once.   */
int ndp_pair_means(const float *src, int n_src, const float *tgt, int n_tgt, float *means, void *stream);

/* ---- Neural scene-flow prior (NSFP) baseline, SURVEY section 8 f3 ----------------------------------------------
 * x_out = x + MLP(x) with the 9-layer ReLU MLP of nets.py:256-292 (parameter layout: ndp_types.h), one launch per
 * layer, the 128x128 layers on the fp32 MFMA with the weight slice stationary in registers.
 * act: [8][cap][128] post-ReLU activations h1..h8 (cap = n rounded up to 64) saved for ndp_nsfp_bwd, or NULL for
 * inference, in which case tmp [2][cap][128] is the ping-pong scratch.  (registration.py:506-507, :534-536)          */
int ndp_nsfp_fwd(const float *params, const float *x, int n, float *x_out, float *act, float *tmp, void *stream);

/* Gradient of a scalar loss wrt all parameters given g = dL/dx_out [n][3] (autograd of registration.py:506-529).
 * act is consumed (h8 is overwritten by the running dz).  dO_work [cap][16]; grads_part [n_part][p_stride]
 * partials, one per workgroup, to be folded by ndp_grad_reduce (deterministic order, no atomics).                 */
int ndp_nsfp_bwd(const float *params, const float *x, int n, float *act, const float *g,
                 float *dO_work, float *grads_part, int n_part, int p_stride, void *stream);

/* ---- Nerfies baseline, SURVEY section 8 f3 (second half) ------------------------------------------------------------
 * Nerfies_Deformation.forward (nets.py:187-253): windowed 39-wide posenc -> 39->128 -> six 128x128 ReLU layers -> w / v
 * heads -> SE(3) exponential warp, plus the per-point Jacobian d warp / d x (carried forward as three tangent rows per
 * point: every buffer holds four planes [primal | d/dx0 | d/dx1 | d/dx2] of cap = n rounded up to 64 rows) and
 * reg = mean_p log(max(sigma_max(J_p), 1e-6))^2 (loss.py:373-379, per point in double).  Parameter layout:
 * [W_in 128x39 | b_in 128 | (W_l 128x128 | b_l 128) l=1..6 | W_h 6x128 (w rows, v rows) | b_h 6], P = 104 966.
 *   window6: HOST array of the six annealing weights of the iteration (nets.py:223-225)
 *   act: [save ? 7 : 2][4 cap][128] activation planes; pe: [cap][40]; heads: [4 cap][8]; work: [cap] doubles.           */
int ndp_nerfies_fwd(const float *params, const float *x, int n, const float *window6, float *x_out, float *J,
                    float *reg, float *act, int save, float *pe, float *heads, double *work, void *stream);
/* Gradient of a scalar loss wrt all parameters given g = dL/d(x_out) [n][3]; the regulariser carries none upstream (its
 * Jacobian is built with create_graph=False).  act (save = 1) is consumed.  dO_work [cap][16]; grads_part
 * [n_part][p_stride] partials for ndp_grad_reduce.                                                                     */
int ndp_nerfies_bwd(const float *params, const float *x, int n, float *act, const float *pe, const float *heads,
                    const float *g, float *dO_work, float *grads_part, int n_part, int p_stride, void *stream);

/* ---- Embedded-deformation N-ICP baseline, SURVEY section 8 f4 --------------------------------------------------------
 * Node graph: n_nodes positions `nodes` [n][3], axis-angle rotations `phi` [n][3], translations `t` [n][3]; every point has
 * six anchors (`anchors` [S][6] int32, -1 = the LAST node with weight 0, as upstream's negative indexing) with skinning
 * `weights` [S][6]; graph edges `edges` [n][K] int32 (-1 padded the same way) with `edge_w` [n][K].
 * ndp_ed_warp: R_work [n][9] <- axis_angle_to_matrix(phi) ; y = ED_warp(x) (geometry.py:37-49).  phi == NULL: R_work is used as it
 *              is (upstream's final warp runs with the R of the last loop iteration, registration.py:376, 449-453).
 * ndp_ed_arap: out[0] = arap_cost (loss.py:261-285) for the R of the last ndp_ed_warp.
 * ndp_ed_grad: grads [phi (3n) | t (3n)] = d/d(phi, t) of <gy, ED_warp(x)> + w_arap * arap  (gy = dL/dy [S][3], e.g. the
 * Chamfer gradient times w_cd): one workgroup per node, fixed summation order.                                          */
int ndp_ed_warp(const float *x, int S, const int *anchors, const float *weights, const float *nodes, int n_nodes,
                const float *phi, const float *t, float *R_work, float *y, void *stream);
int ndp_ed_arap(const float *nodes, int n_nodes, const float *R, const float *t, const int *edges, const float *edge_w,
                int K, float *out, void *stream);
int ndp_ed_grad(const float *x, int S, const int *anchors, const float *weights, const float *gy, const float *nodes,
                int n_nodes, const float *R, const float *t, const float *phi, const int *edges, const float *edge_w,
                int K, float w_arap, float *grads, void *stream);

/* Exact brute-force 1-NN in both directions (pytorch3d knn_points K=1 as called at loss.py:177-178):
 * d2x[i] = min_j |x_i - y_j|^2 (fma chain over x,y,z), idx_x[i] = lowest argmin; same for y->x.  */
int ndp_chamfer_nn_fwd(const float *x, int S, const float *y, int T,
                       float *d2x, int *idx_x, float *d2y, int *idx_y, void *stream);

/* The same result from ONE pass over the S x T squared distances (what the engine runs every tick): each distance is
 * evaluated once and serves both directions -- row minima thread-private, column minima by a cross-lane butterfly per
 * wave and an LDS table per workgroup, exact lowest indices by a re-scan of the winning 128-source block in LDS.
 * Bit-identical to ndp_chamfer_nn_fwd.  ws_row: scratch of ndp_engine_nn_workspace(S rounded up to 64, T rounded up
 * to 64) floats.                                                                                                   */
int ndp_chamfer_nn_onepass(const float *x, int S, const float *y, int T, float *d2x, int *idx_x, float *d2y,
                           int *idx_y, float *ws_row, void *stream);

/* The one-pass result with the S x T distances on the bf16 MATRIX pipe (the engine's nn_mode 2): |x - y|^2 = |x|^2 + |y|^2 - 2 x.y as
 * one contraction per 32 x 32 tile, every term a three-way bf16 split (two v_mfma_f32_32x32x16_bf16 per 1024 distances); that value
 * only SELECTS candidates -- every candidate within the rounding bound is re-evaluated with the exact fma chain, so d2 and the lowest
 * index are bit-identical to ndp_chamfer_nn_fwd.  2048 sources are resident in LDS at a time; more are walked in passes (highest
 * indices first, so the exact comparisons still end on the lowest index): no size limit since round 3
 * (ndp_engine_nn_matrix_fits(n_cap) is kept and returns 1).  ws_row as for ndp_chamfer_nn_onepass.                           */
int ndp_chamfer_nn_matrix(const float *x, int S, const float *y, int T, float *d2x, int *idx_x, float *d2y,
                          int *idx_y, float *ws_row, void *stream);
int ndp_engine_nn_matrix_fits(int n_cap);
/* 1 when the column table of the one-pass VECTOR kernel (engine nn_mode 0 / ndp_chamfer_nn_onepass) fits LDS for n_cap sources;
 * beyond that an engine must use nn_mode 1 (latency shape: no table, no size limit).                                                 */
int ndp_engine_nn_onepass_fits(int n_cap);

/* Truncated L1 Chamfer value and gradient from the NN result (loss.py:185-258 and its autograd):
 * loss[0] = sum_i sqrt(d2x_i)[d2x_i<trunc]/S + sum_j sqrt(d2y_j)[d2y_j<trunc]/T   (point_sum != 0: without the /S, /T --
 * point_reduction="sum", loss.py:233-235) ;
 * gx [S][3] = dloss/dx (contributions of y_j -> x_i added in ascending j).                      */
int ndp_chamfer_l1_bwd(const float *x, int S, const float *y, int T, float trunc,
                       const float *d2x, const int *idx_x, const float *d2y, const int *idx_y,
                       float *loss, float *gx, int point_sum, void *stream);

/* compute_flow_metrics / scene_flow_metrics (loss.py:382-403, 431-471) on the device: flow, flow_gt [n][3]; overlap [n]
 * bytes (0 / 1) or NULL.  out15 (device, 3 x 5 doubles), per subset {all, overlap, not overlap}:
 * {sum of end-point errors, #AccS hits, #AccR hits, #outliers, #points}; the caller divides (an empty subset gives NaN
 * like upstream's mean over nothing).                                                                                 */
int ndp_flow_metrics(const float *flow, const float *flow_gt, const unsigned char *overlap, int n, double *out15, void *stream);

/* Landmark loss mean_k |x_k - t_k|^2 and gradient (registration.py:201-203). */
int ndp_landmark_mse_fwd_bwd(const float *x, const float *t, int K, float *loss, float *gx, void *stream);

/* torch.optim.Adam single-tensor step t (1-based) on P parameters (registration.py:176,237).
 * neg_step = -(lr / (1 - b1^t)), bc2_sqrt = sqrt(1 - b2^t) are computed by the caller in double
 * and passed as float, exactly as torch casts its Python scalars.                               */
int ndp_adam_step(float *params, const float *grads, float *m, float *v, int P,
                  float w1, float b2, float w2, float neg_step, float bc2_sqrt, float eps, void *stream);

/* ------------------------------------------------------------------ batched engine
 * B independent pairs advance together, one "tick" = one iteration of the inner loop of
 * optimize_deformation_pyramid (registration.py:184-238) for every unfinished pair, each pair at
 * its own level / iteration, early stop decided on the device (registration.py:226-232).        */

typedef struct ndp_pair_geom {
    int K;            /* landmark count (0 without landmarks)                        */
    int S;            /* Chamfer source samples (0 in landmark-only mode)            */
    int T;            /* Chamfer target samples                                      */
    int pad;
} ndp_pair_geom;

typedef struct ndp_pair_state {
    int level;                       /* current level; == m when the pair is finished           */
    int iter;                        /* loss evaluations done at this level                     */
    int break_counter;               /* registration.py:179                                     */
    int adam_t;                      /* Adam steps taken at this level                          */
    int cur;                         /* which half of pts[] is the level input                  */
    int decision;                    /* what this tick's update kernel must do (NDP_DEC_*)      */
    int total_steps;                 /* Adam steps over all levels                              */
    int total_evals;                 /* loss evaluations over all levels                        */
    float loss;                      /* last evaluated loss                                     */
    int step_level;                  /* level this tick's Adam step applies to                  */
    int step_t;                      /* Adam step number (1-based) of this tick's step          */
    int pad;
    double loss_prev;                /* registration.py:180                                     */
    int evals_per_level[NDP_MAX_LEVELS];
} ndp_pair_state;

enum { NDP_DEC_STEP = 0, NDP_DEC_ADVANCE = 1, NDP_DEC_STEP_ADVANCE = 2, NDP_DEC_IDLE = 3 };

typedef struct ndp_engine {
    ndp_layer_desc desc;             /* shared by all levels; nonrigidity = 1: levels > 0 gated  */
    int m, k0;
    int P, p_stride;                 /* params per level, padded stride (multiple of 4)         */
    int iters, max_break_count, early_stop;
    int B, G;                        /* pairs; blocks per pair in the level kernels             */
    int n_cap, t_cap;                /* per-pair capacities, multiples of NDP_TILE              */
    double break_threshold_ratio;
    float w_cd, trunc;
    float adam_w1, adam_b2, adam_w2, adam_eps;
    float w_reg, pad_f;              /* nonrigidity BCE weight (registration.py:216-220)         */
    ndp_pair_geom *geom;             /* [B]                                                     */
    ndp_pair_state *state;           /* [2][B] double-buffered by tick parity                   */
    float *pts;                      /* [B][2][n_cap][3]  landmarks first, then samples         */
    float *ldmk_t;                   /* [B][n_cap][3]                                           */
    float *tgt;                      /* [B][t_cap][3]                                           */
    float *params;                   /* [B][m][p_stride]                                        */
    float *gpart;                    /* [B][G][p_stride] gradient partials (G == 1 without gemm_mode bit 1024: the W1 / W2 blocks are not written) */
    float *adam_m, *adam_v;          /* [B][p_stride]                                           */
    float *act;                      /* [B][3][n_cap][128] fp32 rows of h0, h1, h2 ([B][depth][n_cap][width] for other shapes) -- except under the default gemm_mode 7 (fused
                                        split backward), where planes 1 and 2 hold h1 and (since ABI 202) h2 per 64-point tile as a
                                        PLANE IMAGE of the same size: two [64][128] fp16 planes hi = fp16(2^6 h) | lo = fp16(2^6 h - hi),
                                        rows of 256 bytes with their 16-byte granules XOR-swizzled (csrc/ndp_fwd_split.inc: bf_swz) --
                                        the backward's LDS layout, written by the forward and pulled in by LDS-DMA; the ReLU mask is
                                        hi's SIGN BIT: hi = -0 where the pre-activation is <= 0, hi >= +0 where it is positive.  Every activation a split forward stores -- image
                                        or fp32 row -- is the BOUNDED value: it saturates at 65504 / 64 = 1023.5 (the fp16 operand
                                        range of the 2^6-scaled splits); this network's activations are O(1)        */
    float *heads;                    /* [B][n_cap][NDP_HROW]                                    */
    float *d2x; int *idx_x;          /* [B][n_cap]                                              */
    float *d2y; int *idx_y;          /* [B][t_cap]                                              */
    const float *adam_tab;           /* [iters+1][2]: {neg_step, bc2_sqrt} for t = 1..iters     */
    float *dO;                       /* [B][n_cap][16] mlp_scale * dL/d(head outputs), this tick */
    float *nn_row;                   /* one-pass 1-NN row partials, B x ndp_engine_nn_workspace() floats (NULL if w_cd == 0) */
    int nn_mode, gemm_mode;          /* gemm_mode 0: level kernels on the fp32 MFMA, bitwise the oracle's fma chain.  Mask 1 forward,
                                        2 bwd1, 4 bwd2: their 128 x 128 contractions from two-way fp16 splits (x' = hi + lo of operands
                                        pre-scaled by a power of two, three products, fp32 accumulate; the two-launch backward of bit
                                        16: hi + 2^-11 lo) on the 16-bit MFMA -- fp32-level accuracy, not bitwise the chain
                                        (csrc/ndp_*_split.inc); 7 is what Registration uses by default.  With 1 | 2 the forward does
                                        not store h0 (act[b][0] is left untouched): the backward recomputes it from the saved
                                        encoding with the forward's own layer-0 MFMA; bit 8 makes the forward store it all the same
                                        (tests).  With 2 | 4 both backward layers run as ONE launch (k_eng_bwd_f,
                                        csrc/ndp_bwd_fused.inc: dz1 stays in LDS, one accumulator per product on operands pre-scaled
                                        by powers of two -- activations and weights beyond 1023 saturate there); bit 16: as the two
                                        launches k_eng_bwd2_8 + k_eng_bwd1_8 instead; bit 32 (tests): the fused launch also writes
                                        dz1 over the h2 plane of `act`, where the two-launch form leaves it.
                                        Measured variants kept behind bits: 64 the Adam step by the last-arriving
                                        backward workgroup of a pair (no k_eng_update launch; gmax must then be [2 B]); 128 the 4-wave
                                        shapes of the nearest-neighbour kernels; 256 a persistent one-launch tick for a handful of
                                        resident pairs (k_eng_tick_small; gmax [2 B]) -- all bitwise the default, all slower;
                                        512 the per-point warp of the split forward as a launch of its own (k_eng_warp) instead of
                                        behind the forward workgroup's tile loop (same arithmetic, same bits); 1024 the whole Adam
                                        step in k_eng_update -- without it an engine with G == 1 steps the two 128 x 128 matrices
                                        behind the fused backward's tile loop (no gradient partial of them is written: gpart's two
                                        matrix blocks are then undefined) and the rest in k_eng_update_rest: bitwise the same state.
                                        nn_mode 0: one-pass kernel, distances on the vector pipe; 2: the same on the bf16 matrix pipe
                                        with exact re-evaluation (bit-identical, needs ndp_engine_nn_matrix_fits(n_cap)); 1: latency
                                        shape -- two passes in 64-query workgroups, S/64 + T/64 of them per pair -- for a handful of
                                        resident pairs   */
    unsigned int *gmax;              /* [B] bit pattern of max |dO| of the pair this tick (zeroed by the forward stage, raised by
                                        the loss stage): the power-of-two scale that puts the split backward's gradient operands
                                        into fp16's range.  Required when gemm_mode & 6, else may be NULL.                       */
} ndp_engine;

/* Floats PER PAIR of the row-partial buffer of the one-pass nearest-neighbour kernel ({d2, idx} per source and
 * 128-target chunk).                                                                                             */
int ndp_engine_nn_workspace(int n_cap, int t_cap, long long *row_floats);

/* Launch n_ticks ticks starting at tick index tick0 (parity selects the state buffer read).
 * The caller initialises state[tick0 & 1] (level 0, iter 0, break_counter 0, loss_prev 1e6, cur 0).
 * Asynchronous on `stream`; read state[(tick0 + n_ticks) & 1] after synchronising.              */
int ndp_engine_run(const ndp_engine *e, int tick0, int n_ticks, void *stream);

/* Slot (re)fill in one launch for several pairs (registration.py:150-164: centring, sampling by the
 * permutation prefix, landmark centring; :133-140 the freshly initialised pyramid; :176 fresh Adam state):
 *   pts[slot][0][i]   = i < K ? ldmk_s[i] - mean_s : src[perm_s[i-K]] - mean_s      (i < K+S, rest zero)
 *   ldmk_t[slot][k]   = ldmk_t[k] - mean_t ;  tgt[slot][j] = tgt[perm_t[j]] - mean_t  (j < T)
 *   params[slot]      = params ; adam_m = adam_v = 0 ; geom = (K,S,T) ; state[tick & 1][slot] = fresh
 * perm_* NULL = identity, means NULL = no centring (n_src > 0: the means are computed by this call, see the struct).  A job with
 * params == NULL parks the slot (level = m).
 * `jobs` is a HOST array (copied into the kernel arguments); at most NDP_MAX_LOAD_JOBS per call.       */
typedef struct ndp_load_job {
    const float *src, *tgt;          /* raw clouds [n][3] (device) */
    const int *perm_s, *perm_t;      /* first S / T entries of the sampling permutations (device int32) or NULL */
    const float *ldmk_s, *ldmk_t;    /* [K][3] or NULL */
    const float *params;             /* [m][p_stride] initial parameters (device) or NULL = park */
    float *means;                    /* [8] as ndp_pair_means leaves them, or NULL.  With n_src > 0 they are an OUTPUT first: the call
                                        computes the means of src [n_src] / tgt [n_tgt] into `means` -- ONE launch for all such jobs
                                        of the call, the arithmetic of ndp_pair_means (same bits) -- and then centres with them */
    int slot, K, S, T;
    int n_src, n_tgt;                /* points of the raw clouds when the means are to be computed here, else 0 */
} ndp_load_job;
#define NDP_MAX_LOAD_JOBS 16
int ndp_engine_load(const ndp_engine *e, int tick, const ndp_load_job *jobs, int n_jobs, void *stream);

/* Same launches with HIP events around every kernel, recorded on `stream`; ms_out[NDP_TICK_KERNELS] (HOST memory)
 * receives the summed durations of the forward, NN, loss/gradient, backward-2 (+ heads), backward-1 and update
 * kernels over the n_ticks ticks.  Synchronises `stream`.  Measurement aid for bench.py (roofline), not a product path. */
#define NDP_TICK_KERNELS 6
int ndp_engine_run_timed(const ndp_engine *e, int tick0, int n_ticks, void *stream, float *ms_out);

/* ONE tick, restricted to the launches of stages [stage_lo, stage_hi]: 0 forward, 1 nearest neighbours, 2 loss / decision /
 * dL/dx', 3 bwd2, 4 bwd1, 5 update.  Test and measurement aid: the buffers a kernel leaves behind (act, heads, dO, gpart) can be
 * read between stages; the stages 0..5 of a tick run in order, in any grouping, equal ndp_engine_run(e, tick, 1).               */
int ndp_engine_run_stages(const ndp_engine *e, int tick, int stage_lo, int stage_hi, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NDP_HIP_H */
