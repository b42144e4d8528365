/* flowcon_hip.h -- C ABI of libflowcon_hip.so: MI355X (gfx950) bijector kernels.
 *
 * Every entry point
 *   - takes raw DEVICE pointers (f32 unless stated) and plain sizes; no torch types,
 *   - allocates nothing, is asynchronous on `stream` (a hipStream_t passed as void*),
 *   - returns a hipError_t as int (0 == hipSuccess),
 *   - never mutates its inputs (x may alias y where stated).
 * The reference (FlowConductor, pure Python on ATen) has no FFI; each function below names
 * the reference op sequence it stands in for (file:line under the reference tree).
 * INTEGRATION.md shows the ctypes binding a maintainer would add on the reference side.
 *
 * Common conventions
 *   x, y        [n, d] row-major; transformed columns are `cols[0..d_t)` (int32) or, when
 *               cols == NULL, d_t == d and dim j is column j.  Columns not listed in `cols`
 *               are copied x -> y unchanged (coupling identity half, coupling.py:96-98).
 *   params      per-sample parameters [n, rowlen]; shared_params != 0 means a single
 *               [rowlen] row used for the whole batch (nonlinearities.py:246-247).
 *   logabsdet   [n] or NULL. lad_mode: 0 store, 1 accumulate (+=), 2 store negated,
 *               3 accumulate negated.
 *   err_flag    device uint32 word or NULL; kernels OR in FC_ERR_* bits, the host reads it
 *               to raise the reference's exceptions (transforms/base.py:10-19).
 */
#ifndef FLOWCON_HIP_H_
#define FLOWCON_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FC_ERR_OUTSIDE_DOMAIN 1u /* InputOutsideDomain, splines/rational_quadratic.py:81-82 */
#define FC_ERR_DISCRIMINANT 2u   /* assert (discriminant >= 0).all(), rational_quadratic.py:142 */
#define FC_ERR_NONFINITE 4u

/* ABI version of this header; fc_abi_version() must return it.  Bumped whenever an exported entry, an accepted enum value
 * or a documented behaviour changes (2: round 4 -- fc_rq_fused_linear_backward is one launch, fc_comm_* entries of round 3,
 * FC_AFFINE_MAF_SOFTPLUS / FC_RQ_STREAMED_WEIGHTS). */
#define FC_ABI_VERSION 2

int fc_abi_version(void);

/* ---- rational-quadratic spline ------------------------------------------------------- */
#define FC_RQ_ACCUMULATE_LOGABSDET 1 /* fc_rq_spline_fused_linear: logabsdet[n] += sum (the caller's running
                                       total of CompositeTransform._cascade, transforms/base.py:45-52) */
#define FC_RQ_FORCE_TILE 2 /* fc_rq_spline: always the LDS-tile kernel, never the register / wave kernel (A/B
                              measurements; the results are the same) */
#define FC_RQ_RAW_WEIGHTS 4 /* fc_rq_spline_fused_linear: w_pad / bias_pad are the final nn.Linear's tensors as they
                               are, [d_t * 23, 64] and [d_t * 23] (no padding rows; same results) */
#define FC_RQ_STREAMED_WEIGHTS 8 /* fc_rq_spline_fused_general: always the streamed-weight kernel, never the
                                    resident-weight instances of fc_rq_fused4 (A/B measurements and tests) */

typedef struct fc_rq_config {
  int32_t num_bins;       /* K */
  int32_t tails;          /* 0: none, 1: "linear" (rational_quadratic.py:32-42) */
  int32_t inverse;        /* 0 forward, 1 inverse */
  int32_t flags;          /* FC_RQ_* bits */
  float left, right, bottom, top; /* linear tails: -B, B, -B, B */
  double min_bin_width, min_bin_height, min_derivative; /* python floats of the reference */
  float wh_divisor;       /* sqrt(hidden_features) or 1 (coupling.py:554-559) */
  float softplus_beta;    /* 1, or ln2/(1-min_derivative) (rational_quadratic.py:100-103) */
  float tail_constant;    /* (float)log(exp(1-min_derivative)-1) (rational_quadratic.py:34) */
  float reserved2;
} fc_rq_config;

/* Replaces unconstrained_rational_quadratic_spline / rational_quadratic_spline
 * (splines/rational_quadratic.py:13-181) together with the reshape + in-place scaling +
 * sum_except_batch of PiecewiseRationalQuadraticCouplingTransform (coupling.py:279-293,
 * 549-582), the masked AR variant (autoregressive/autoregressive.py:583-621) and
 * PiecewiseRationalQuadraticCDF (nonlinearities.py:429-487).
 * params row layout per transformed dim: [K widths | K heights | K-1 (linear tails) or K+1
 * derivatives]; rowlen = d_t * (3K -/+ 1). */
int fc_rq_spline(const float* x, float* y, const float* params, const int32_t* cols,
                 float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t,
                 int32_t shared_params, int32_t lad_mode, const fc_rq_config* cfg, void* stream);

/* Final conditioner layer fused with the RQ-spline coupling bijector: the [n, d_t*(3K-1)] parameter
 * tensor  h @ W^T + b  (flowcon/nn/nets/resnet.py:91,99 final_layer) is produced on the matrix cores (three-term
 * scaled f16 splits, f32-GEMM accuracy) straight into the registers of the lanes that evaluate the spline
 * (coupling.py:279-293, 549-582); it never reaches HBM.  cfg->flags: FC_RQ_ACCUMULATE_LOGABSDET, FC_RQ_RAW_WEIGHTS.  Specialised: hidden == 64, 1 <= d_t <= 32, K == 8, linear tails, d <= 128,
 * n % 32 == 0 (callers route other shapes / the leftover rows through fc_rq_spline).
 *   h        [n, 64]  last hidden activation of the conditioner (input of its final Linear)
 *   w_pad    [dp*24, 64]  the weight, zero-padded from 23 to 24 rows per dim (row j*24+i = W row j*23+i, i < 23)
 *                         and from d_t to dp = ceil(d_t / 4) * 4 dims
 *   bias_pad [dp*24]      bias with the same padding */
int fc_rq_spline_fused_linear(const float* x, float* y, const float* h, const float* w_pad,
                              const float* bias_pad, const int32_t* cols, float* logabsdet,
                              uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t, int32_t hidden,
                              const fc_rq_config* cfg, void* stream);

/* The same fusion for general layer shapes -- the reference's DEFAULT coupling layer has num_bins = 10
 * (coupling.py:507), any hidden_features (nn/nets/resnet.py:62) and allows tails = None (coupling.py:543-547):
 * K = 4..16, cfg->tails 0 or 1, hidden in {64, 128, 256} (narrower activations zero-padded), 1 <= d_t <= 32, d <= 128,
 * n % 32 == 0.  The weights are not resident in registers here: the host packs them once per parameter version into
 * matrix-core fragment order (scaled by a power of two per group of 4 dims, split into two f16 pieces) and every wave
 * streams the fragments of its dims from L2.
 *   h         [n, hidden]
 *   w_frag    f16 [groups][hidden/32][T][2][64][8], groups = ceil(d_t/4), T = ceil(P/4), P = 3K -/+ 1: fragment
 *             (group, k-step, tile t, piece hi/lo): lane l holds 2^S W[dim 4 group + ((l&15)>>2)][param 4t + (l&3)]
 *             [k = 32 kstep + 8 (l>>4) + j], j < 8
 *   w_unscale f32 [groups] = 2^-S;  bias_pad f32 [groups][4][4T]
 * cfg->flags: FC_RQ_ACCUMULATE_LOGABSDET, FC_RQ_STREAMED_WEIGHTS.  Without tails, inputs outside [left, right] set
 * FC_ERR_OUTSIDE_DOMAIN.  hidden == 64 with K = 4..7, 9..11 (linear tails) or 4..10 (no tails) -- K = 10 is the
 * reference's default num_bins -- runs with both weight pieces resident in registers and a hand-scheduled evaluation
 * (fc_rq_fused4_body.h), same results contract. */
int fc_rq_spline_fused_general(const float* x, float* y, const float* h, const void* w_frag,
                               const float* w_unscale, const float* bias_pad, const int32_t* cols,
                               float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t,
                               int32_t hidden, const fc_rq_config* cfg, void* stream);

/* Backward of the fused layer above (forward direction, hidden == 64): what torch.autograd yields for the reference's
 * final Linear + spline (nn/nets/resnet.py:99 + rational_quadratic.py:13-181; the reference trains through them,
 * examples/toy_2d.py:57-68) WITHOUT the [n, d_t P] parameter / parameter-gradient tensors: the parameters are recomputed
 * on the matrix cores from the saved h, the closed-form spline backward runs on them in registers.
 * One launch (role must be FC_RQ_BACKWARD_ONE_LAUNCH; ABI 1 ran two launches, roles 0 "dx" and 1 "dw", each recomputing G):
 *   grad_x [n, d] (= grad_y on the other columns), grad_h [n, 64] = W^T G              (deterministic)
 *   grad_bias_pad [groups][4][4T] += sum_n G,  grad_w_pad [groups][4][4T][64] += sum_n G (x) h
 *                                              (both accumulate with float atomics: zero them first)
 * grad_x / grad_h must not alias grad_y (they serve as scratch between the kernel's sweeps over groups of 16 dims).
 * w_frag / w_unscale / bias_pad as for fc_rq_spline_fused_general with hidden == 64; wt_frag f16
 * [groups][4 hidden tiles][KK][2][64][8], KK = ceil(4T / 8): fragment (hidden tile ht, k-step kk, piece): lane l holds
 * 2^S W[dim 4 group + (l>>4)][param 8 kk + j][hidden 16 ht + (l&15)].  n % 32 == 0; cfg->inverse must be 0; 4T <= 32. */
#define FC_RQ_BACKWARD_ONE_LAUNCH 3
int fc_rq_fused_linear_backward(int32_t role, const float* x, const float* h, const float* grad_y,
                                const float* grad_logabsdet, const void* w_frag, const float* w_unscale,
                                const float* bias_pad, const void* wt_frag, const int32_t* cols, float* grad_x,
                                float* grad_h, float* grad_bias_pad, float* grad_w_pad, int64_t n, int32_t d,
                                int32_t d_t, const fc_rq_config* cfg, void* stream);

/* Autoregressive INVERSE of a MADE-conditioned layer with the D passes on the device (the reference's sampling loop,
 * flowcon/transforms/autoregressive/autoregressive.py:44-53: D conditioner passes, pass d fixing column d): a wave keeps 16
 * rows, the pre-masked hidden stack (made.py:205-283: initial layer + residual blocks, mask * weight) in LDS, and per pass
 * runs the stack, the params_per_dim final-layer rows of dim d and the element-wise inverse of that column.
 *   z, y [n, d] (d <= 64, n % 16 == 0); logabsdet [n] = sum over the columns (added onto it with FC_RQ_ACCUMULATE_LOGABSDET
 *   in cfg->flags).
 *   hidden_frag / hidden_unscale / hidden_bias: the image of fc_resnet_hidden_packed made from the MASKED weights
 *   (rows 64-padded, in_features = d, 1 k-step for d <= 32 else 2), num_blocks <= 3 ReLU residual blocks.
 *   final_frag: f16 [d][2 k-steps][PT][2 pieces][64 lanes][8], PT = ceil(params_per_dim / 16): lane l of fragment
 *   (dim, ks, t, piece) holds 2^S_dim Wmasked[dim * P + 16 t + (l & 15)][32 ks + 8 (l >> 4) + j] (zero rows beyond P);
 *   final_unscale [d] = 2^-S_dim; final_bias [d][16 PT].
 *   kind FC_MADE_AFFINE (params_per_dim 2: unconstrained scale, shift -- autoregressive.py:97-129; cfg may be NULL) or
 *   FC_MADE_RQ (cfg: the spline, any K <= 16 and tail mode, autoregressive.py:529-621; cfg->inverse is ignored).
 *   units_needed [d] or NULL: pass c reads only the first units_needed[c] hidden units (in feature order 0..63 of EVERY
 *   hidden layer: the caller renumbers the units so that each pass's units form a prefix -- for the reference's degrees,
 *   made.py:13-24, the units of degree <= c) and the kernel computes just the 16-unit product tiles and 32-unit k-steps
 *   that hold them; the other units may then hold any finite value (they meet zeroed weights only).  NULL: all 64.
 *   err_flag: FC_ERR_* bits of the spline inverse. */
#define FC_MADE_AFFINE 0
#define FC_MADE_RQ 1
int fc_made_inverse(const float* z, float* y, float* logabsdet, const void* hidden_frag, const float* hidden_unscale,
                    const float* hidden_bias, const void* final_frag, const float* final_unscale, const float* final_bias,
                    const int32_t* units_needed, uint32_t* err_flag, int64_t n, int32_t d, int32_t num_blocks, int32_t params_per_dim, int32_t kind,
                    const fc_rq_config* cfg, void* stream);

/* Backward of fc_affine in the forward direction, per-sample parameters (coupling.py:234-252,
 * autoregressive.py:97-129 under torch.autograd): grad_x[n, cols[j]] = gy s; grad_params in the layout of
 * `params` for the same `activation`.  Other columns of grad_x are NOT written. */
int fc_affine_backward(const float* x, const float* params, const int32_t* cols, const float* grad_y,
                       const float* grad_logabsdet, float* grad_x, float* grad_params, int64_t n,
                       int32_t d, int32_t d_t, int32_t activation, void* stream);

/* Backward of fc_rq_spline in the forward direction (what torch.autograd yields for the reference's op
 * sequence, rational_quadratic.py:13-181; the reference trains through it, examples/toy_2d.py:57-68):
 *   grad_x[n, cols[j]]            = gy dy/dx + gl dlogabsdet/dx      (the other columns: grad_x = grad_y, the
 *                                                                      bijector copies them)
 *   grad_params[n, j*P .. j*P+P)  = gy dy/dp + gl dlogabsdet/dp      for the P = 3K-1 (3K+1) raw values of dim j
 * x [n, d], params [n, d_t*P] (per-sample rows only), grad_y [n, d], grad_logabsdet [n] or NULL (zeros).
 * cfg->inverse must be 0; num_bins <= 32. */
int fc_rq_spline_backward(const float* x, const float* params, const int32_t* cols, const float* grad_y,
                          const float* grad_logabsdet, float* grad_x, float* grad_params, int64_t n,
                          int32_t d, int32_t d_t, const fc_rq_config* cfg, void* stream);

/* Hidden layers of the ResidualNet conditioner (flowcon/nn/nets/resnet.py:39-53, 93-99) as one kernel:
 *   h = W0 x[:, id_cols] + b0;  per block: h += W2 relu(W1 relu(h) + b1) + b2          -> h [n, 64]
 * Products on the f16 matrix cores as three-term scaled two-piece splits (f32-GEMM accuracy); a wave carries
 * 16 samples through all layers in registers; the only HBM traffic is x in and h out.
 * `activation` (FC_ACT_*) is the blocks' activation: ReLU kernels carry nothing but a v_max; the others share one
 * kernel family with a uniform switch (tanh, SiLU, ELU, LeakyReLU, sigmoid as ATen computes them in float32).
 * Specialised: hidden == 64, num_blocks <= 4, no batch norm / active dropout (context: next entry),
 * in_features <= 64, n % 16 == 0, h 16-byte aligned.
 * Weights are the nn.Linear tensors as they are, row-major f32: w0 [64, in_features]; wb [blocks][2][64][64]
 * (linear_layers[0], linear_layers[1] of each block); b0 [64]; bb [blocks][2][64]. */
#define FC_ACT_RELU 0
#define FC_ACT_TANH 1
#define FC_ACT_SILU 2
#define FC_ACT_ELU 3        /* activation_param = alpha */
#define FC_ACT_LEAKY_RELU 4 /* activation_param = negative slope */
#define FC_ACT_SIGMOID 5
int fc_resnet_hidden(const float* x, float* h, const int32_t* id_cols, const float* w0,
                     const float* b0, const float* wb, const float* bb, int64_t n, int32_t d,
                     int32_t in_features, int32_t hidden, int32_t num_blocks, int32_t activation,
                     float activation_param, void* stream);

/* fc_resnet_hidden with the weights prepared ahead of time (same results, bit for bit): w_frag = the layers' f16
 * fragment images back to back ([layer][k-step][tile][piece][lane][8], FC_PACK_HIDDEN jobs of fc_pack_fragments: the
 * initial layer with nks = 1 (in_features <= 32) or 2, every block layer with nks = 2, nt = 4), w_unscale [L] and
 * bias_acc [L][64] from the same jobs, L = 1 + 2 num_blocks.  The kernel then starts with a copy of the image into LDS
 * instead of two rounds of weight loads, the scaling / splitting arithmetic and two barriers (~19 us per launch at
 * 2 blocks).  No context. */
int fc_resnet_hidden_packed(const float* x, float* h, const int32_t* id_cols, const void* w_frag,
                            const float* w_unscale, const float* bias_acc, int64_t n, int32_t d,
                            int32_t in_features, int32_t hidden, int32_t num_blocks, int32_t activation,
                            float activation_param, void* stream);

/* One kernel per AFFINE coupling layer (flowcon/transforms/coupling.py:73-100 + :212-269 with a ResidualNet conditioner,
 * nn/nets/resnet.py:55-100): hidden stack, final Linear and the affine / additive bijector; neither h nor the [n, 2 d_t]
 * parameter tensor reaches memory.  y [n, d] (must not alias x) receives the identity columns unchanged and the transformed
 * ones; logabsdet [n] is written, or added to when accumulate != 0 (CompositeTransform's running total).
 *   w_frag / w_unscale / bias_acc: the image of fc_resnet_hidden_packed with ONE MORE 64 x 64 layer, the final Linear
 *   re-ordered as rows 0..31 = its shift rows of dims 0..31, rows 32..63 = its scale rows (zero rows beyond d_t; additive:
 *   no scale rows), i.e. L = 2 + 2 num_blocks layers.
 * scale_activation: FC_AFFINE_SIGMOID_PLUS2, FC_AFFINE_SOFTPLUS_CLAMP3, FC_AFFINE_ADDITIVE, or FC_AFFINE_MAF_SOFTPLUS (the density
 * direction of a masked-autoregressive affine layer, autoregressive.py:97-129: id_cols = tr_cols = all columns, weights pre-masked,
 * final-layer rows in the same [shift 32 | scale 32] order).  ReLU conditioner, hidden == 64
 * (narrower: zero-padded), num_blocks <= 3, in_features <= 64, d_t <= 32, d <= 128, n % 16 == 0; the weight image and the
 * waves' row tiles (8 x 16 x (d | 1) floats) must fit the CU's 160 KB of LDS (else hipErrorInvalidConfiguration). */
int fc_affine_coupling_resnet(const float* x, float* y, const int32_t* id_cols, const int32_t* tr_cols,
                              const void* w_frag, const float* w_unscale, const float* bias_acc, float* logabsdet,
                              int64_t n, int32_t d, int32_t in_features, int32_t d_t, int32_t hidden, int32_t num_blocks,
                              int32_t scale_activation, int32_t inverse, int32_t accumulate, void* stream);

/* Backward of fc_resnet_hidden (what torch.autograd yields for resnet.py:39-53, 93-99): the activations are recomputed
 * from x; grad_h [n, 64] in -> grad_x_id [n, 32 K0S] (gradient wrt x[:, id_cols], K0S = 1 for in_features <= 32 else 2),
 * and, ACCUMULATED with atomics (zero them first): grad_w0 [64][32 K0S], grad_wb [2 num_blocks][64][64], grad_b [L][64],
 * L = 1 + 2 num_blocks.  hidden == 64, num_blocks <= 2, ReLU, no context, n % 128 == 0.  The weight gradients contract
 * over samples on the exact-f32 matrix instruction (v_mfma_f32_16x16x4_f32).
 *   w_frag   forward fragments, f16: layer 0 [K0S][4][2][64][8], then per layer [2][4][2][64][8]; rows of tile t in
 *            accumulator order: row rho <-> feature 32 (t >> 1) + 8 (rho >> 2) + 4 (t & 1) + (rho & 3)
 *   wt_frag  fragments of the transposed weights: per hidden layer [2][4][2][64][8] (rows = in-features in the same order,
 *            k = out-features), last: W0^T [2][2 K0S][2][64][8] (rows = identity features, natural order)
 *   w_unscale [L];  bias_acc [L][4][16]: bias_acc[l][g][4 t + r] = bias_l[32 (t >> 1) + 8 g + 4 (t & 1) + r] */
int fc_resnet_hidden_backward(const float* x, const float* grad_h, const int32_t* id_cols, const void* w_frag,
                              const void* wt_frag, const float* w_unscale, const float* bias_acc, float* grad_x_id,
                              float* grad_w0, float* grad_wb, float* grad_b, int64_t n, int32_t d, int32_t in_features,
                              int32_t hidden, int32_t num_blocks, int32_t activation, void* stream);

/* The same, with the gradient wrt the identity columns ADDED into the full-width gradient of the layer input instead of
 * written to its own [n, 32 K0S] tensor: grad_x_full[:, id_cols] += ... (every (row, identity column) is owned by one
 * lane).  Saves the index_add pass over [n, d] that follows in a coupling layer's backward. */
int fc_resnet_hidden_backward_accum(const float* x, const float* grad_h, const int32_t* id_cols, const void* w_frag,
                                    const void* wt_frag, const float* w_unscale, const float* bias_acc, float* grad_x_full,
                                    float* grad_w0, float* grad_wb, float* grad_b, int64_t n, int32_t d, int32_t in_features,
                                    int32_t hidden, int32_t num_blocks, int32_t activation, void* stream);

/* The same stack for WIDE conditioners: hidden in {128, 256} (hidden_features is a free constructor argument,
 * resnet.py:62; narrower widths zero-padded by the host), any num_blocks <= 16, no context.  The activations of a
 * 64-sample tile live in LDS as matrix-core B operands shared by the workgroup's 8 waves; wave w owns the output
 * features [w hidden/8, (w+1) hidden/8) of every layer and streams their weight fragments from L2.  n % 64 == 0.
 *   w_frag    f16: layer 0 [hidden/16][K0S][2][64][8] (K0S = 1 for in_features <= 32, else 2), then per layer
 *             [hidden/16][hidden/32][2][64][8]: fragment (tile t, k-step, piece hi/lo): lane l holds
 *             2^S_layer W[16 t + (l & 15)][32 kstep + 8 (l >> 4) + j], j < 8
 *   w_unscale f32 [1 + 2 num_blocks] = 2^-S_layer;  bias f32 [1 + 2 num_blocks][hidden] */
int fc_resnet_hidden_wide(const float* x, float* h, const int32_t* id_cols, const void* w_frag,
                          const float* w_unscale, const float* bias, int64_t n, int32_t d, int32_t in_features,
                          int32_t hidden, int32_t num_blocks, int32_t activation, float activation_param,
                          void* stream);

/* The same with a context (resnet.py:48-49, 94-97): the initial layer sees [x[:, id_cols] | context]
 * (w0 [64, in_features + context_features]) and every block gates its output,
 *   h += (W2 relu(W1 relu(h) + b1) + b2) * sigmoid(Wc context + bc)      (F.glu of the concatenation),
 * wc [blocks][64][context_features] / bc [blocks][64] = blocks[i].context_layer.  context [n, context_features]
 * row-major; context_features <= 32, in_features + context_features <= 64, num_blocks <= 3. */
#define FC_CONTEXT_GLU 1      /* ResidualNet: as above */
#define FC_CONTEXT_ADDITIVE 2 /* MADE (transforms/made.py:100-140, 239-246): w0 [64, in_features] only;
                                 h = W0 x + b0 + act(Wc[0] c + bc[0]); per block h += W2 act(W1 act(h) + b1 + Wc[1+i] c
                                 + bc[1+i]) + b2;  wc [blocks + 1][64][context_features], bc [blocks + 1][64] */
int fc_resnet_hidden_context(const float* x, const float* context, float* h, const int32_t* id_cols,
                             const float* w0, const float* b0, const float* wb, const float* bb,
                             const float* wc, const float* bc, int64_t n, int32_t d, int32_t in_features,
                             int32_t context_features, int32_t hidden, int32_t num_blocks, int32_t context_mode,
                             int32_t activation, float activation_param, void* stream);

/* ---- linear / quadratic / cubic splines ------------------------------------------------------ */
#define FC_SPLINE_LINEAR 0    /* row per dim: [K pdf]                          (splines/linear.py:38-105) */
#define FC_SPLINE_QUADRATIC 1 /* row per dim: [K widths | K-1 or K+1 heights]  (splines/quadratic.py:55-159) */
#define FC_SPLINE_CUBIC 2     /* row per dim: [K widths | K heights | dl | dr] (splines/cubic.py:63-267) */

typedef struct fc_spline_config {
  int32_t kind;           /* FC_SPLINE_* */
  int32_t num_bins;
  int32_t tails;          /* 0 none, 1 "linear" */
  int32_t inverse;
  float left, right, bottom, top;      /* linear tails: -B, B, -B, B */
  double min_bin_width, min_bin_height; /* python floats of the reference */
  float width_divisor, height_divisor;  /* sqrt(hidden_features) or 1 (coupling.py:438-440,
                                           autoregressive.py:430-432: AR quadratic scales widths only) */
  float cubic_eps;                      /* 1e-5 */
  float cubic_quadratic_threshold;      /* 1e-3 */
} fc_spline_config;

/* Sibling piecewise bijectors of the RQ spline, same conventions as fc_rq_spline.  Replaces
 * {linear,quadratic,cubic}_spline and their unconstrained_* wrappers behind
 * Piecewise{Linear,Quadratic,Cubic}CouplingTransform (coupling.py:299-499),
 * MaskedPiecewise{Linear,Quadratic,Cubic}AutoregressiveTransform (autoregressive.py:321-526) and
 * Piecewise{Linear,Quadratic,Cubic}CDF (nonlinearities.py:250-427). */
int fc_piecewise_spline(const float* x, float* y, const float* params, const int32_t* cols,
                        float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t,
                        int32_t shared_params, int32_t lad_mode, const fc_spline_config* cfg,
                        void* stream);

/* Backward of fc_piecewise_spline in the forward direction (cfg->inverse == 0), per-sample rows: grad_x [n, d] (the
 * d_t transformed columns are written; the caller owns the identity columns) and grad_params [n, d_t P] from grad_y
 * [n, d] and grad_logabsdet [n] (NULL = zeros).  What torch.autograd yields for splines/linear.py:38-105,
 * quadratic.py:55-159, cubic.py:63-267: the kernel differentiates the forward evaluation in forward mode, one thread per
 * (element, parameter).  P <= 32 parameters per element. */
int fc_piecewise_spline_backward(const float* x, const float* params, const int32_t* cols, const float* grad_y,
                                 const float* grad_logabsdet, float* grad_x, float* grad_params, int64_t n,
                                 int32_t d, int32_t d_t, const fc_spline_config* cfg, void* stream);

/* ---- affine / additive with per-sample parameters ------------------------------------------ */
#define FC_AFFINE_SIGMOID_PLUS2 0   /* row [shift d_t | u d_t], s = sigmoid(u+2)+1e-3 (coupling.py:224) */
#define FC_AFFINE_SOFTPLUS_CLAMP3 1 /* row [shift | u], s = clamp(softplus(u)+1e-3, 0, 3) (coupling.py:225) */
#define FC_AFFINE_SCALE_GIVEN 2     /* row [shift | s]: scale already activated by the caller */
#define FC_AFFINE_ADDITIVE 3        /* row [shift d_t], s = 1, logabsdet = 0 (coupling.py:255-269) */
#define FC_AFFINE_MAF_SOFTPLUS 4    /* row [d_t, 2] interleaved (u, shift), s = softplus(u)+1e-3
                                       (autoregressive/autoregressive.py:97-129) */

#define FC_AFFINE_SHIFT_TANH2 5     /* row [p d_t], shift = 2*tanh(p), s = 1 (autoregressive.py:164-175;
                                       the reference's inverse subtracts raw p: use ADDITIVE) */

#define FC_AFFINE_SCALE_SOFTPLUS 6   /* row [p d_t], shift = 0, s = softplus(p)+1e-5 (conditional.py:212-272) */

/* y = x*s + shift (forward) or (x - shift)/s (inverse) on the `cols` columns, logabsdet =
 * +/- sum_j log s.  Replaces AffineCouplingTransform / AdditiveCouplingTransform
 * (coupling.py:212-269) incl. the split/merge of coupling.py:82-98, and
 * MaskedAffineAutoregressiveTransform._elementwise_{forward,inverse}
 * (autoregressive/autoregressive.py:97-129). */
int fc_affine(const float* x, float* y, const float* params, const int32_t* cols,
              float* logabsdet, int64_t n, int32_t d, int32_t d_t, int32_t activation,
              int32_t inverse, int32_t shared_params, int32_t lad_mode, void* stream);

/* ---- base-distribution epilogue ---------------------------------------------------------- */
/* out[i] = -0.5 * sum_j z[i,j]^2 - log_z (+ add[i] when add != NULL).  Replaces
 * StandardNormal._log_prob (distributions/normal.py:23-33) and the `log_prob + logabsdet` of
 * Flow._log_prob (flows/base.py:48). */
int fc_standard_normal_log_prob(const float* z, const float* add, float* out, int64_t n,
                                int32_t d, float log_z, void* stream);

/* ---- permutation ------------------------------------------------------------------------- */
/* y[o, j, i] = x[o, perm[j], i] for a tensor viewed as [outer, d, inner]; bit-exact.  x != y.
 * Replaces Permutation._permute (transforms/permutations.py:27-46, torch.index_select). */
int fc_permute(const float* x, float* y, const int32_t* perm, int64_t outer, int32_t d,
               int64_t inner, void* stream);

/* ---- batch-shared point-wise affine maps ---------------------------------------------------- */
/* x, y viewed as [n, m] (m = elements of one batch item); scale/shift have 1 or m entries.
 * mode 0: y = x*scale + shift                    (standard.py:54-60, normalization.py:171-187)
 * mode 1: y = (x - shift)/scale                  (standard.py:62-68, normalization.py:189-204)
 * mode 2: y = weight*((x - mean)/scale) + shift  (BatchNorm eval forward, normalization.py:98-118;
 *         scale = sqrt(var + eps), shift = bias; aux_mean/aux_weight have m entries)
 * mode 3: y = scale*((x - shift)/weight) + mean  (BatchNorm eval inverse, normalization.py:120-141)
 * The logabsdet of these maps is a per-call constant the host computes, as in the reference. */
int fc_pointwise_affine(const float* x, float* y, const float* scale, const float* shift,
                        const float* aux_mean, const float* aux_weight, int64_t n, int64_t m,
                        int32_t scale_len, int32_t shift_len, int32_t mode, void* stream);

/* ---- element-wise non-linearities -------------------------------------------------------------- */
#define FC_EW_EXP 0              /* nonlinearities.py:18-32 */
#define FC_EW_TANH 1             /* :35-48 */
#define FC_EW_LOGTANH 2          /* :51-112   p0 cut_point, p1 alpha, p2 beta, p3 tanh(cut_point) */
#define FC_EW_LEAKY_RELU 3       /* :115-136  p0 slope, p1 1/slope, aux[0] = log_negative_slope */
#define FC_EW_SIGMOID 4          /* :139-169  p0 eps, aux[0] = temperature */
#define FC_EW_SOFTPLUS 5         /* :172-189  p0 threshold, p1 eps */
#define FC_EW_CAUCHY_CDF 6       /* :212-231 */
#define FC_EW_EXTENDED_SOFTPLUS 7 /* :519-552 aux[m] = raw shift; logabsdet_elem = log diag-jacobian */
#define FC_EW_GLU 8              /* :197-209  aux[n, m] = context; gate = sigmoid(context) */

/* x, y viewed as [n, m].  logabsdet_row[i] = sum over the m elements of row i (NULL to skip);
 * logabsdet_elem [n, m] receives the un-summed values (NULL to skip).  Domain violations of the
 * inverses OR FC_ERR_OUTSIDE_DOMAIN into err_flag (InputOutsideDomain, nonlinearities.py:26,43,158,224). */
int fc_elementwise(const float* x, float* y, float* logabsdet_row, float* logabsdet_elem,
                   const float* aux, uint32_t* err_flag, int64_t n, int64_t m, int32_t kind,
                   int32_t inverse, float p0, float p1, float p2, float p3, void* stream);

/* ---- sum of sigmoids ------------------------------------------------------------------------ */
/* Monotone bijector y = sum_k w_k sigmoid(a_k (x - s_k)) / sum_k w_k + extended_softplus(x) - offset
 * with per-(sample, dim) raw rows [S shift | S log_scale | S raw_softmax | 1 softplus shift]
 * (rowlen = d_t * (3S + 1)); logabsdet = sum_j logaddexp(log-jac sigmoids, log-jac softplus).
 * inverse != 0: x = f^-1(y + offset): per-element bracket from [-lim, lim], then a safeguarded Newton search inside the
 * bracket (at most `bisection_iterations` steps; it converges in 5-9 where the reference's bisection spends all 50),
 * then the reference's 2 closing Newton steps; logabsdet = -log f'(x).  A NEGATIVE `bisection_iterations` runs the
 * reference's plain bisection with that many steps instead (A/B measurements only).
 * Replaces SumOfSigmoids.forward (adaptive_sigmoids.py:108-142) with ExtendedSoftplus
 * (nonlinearities.py:519-552), MonotonicTransform.inverse (no_analytic_inv/base.py:23-103) and
 * MaskedSumOfSigmoidsTransform._elementwise_{forward,inverse} (autoregressive.py:301-318). */
int fc_sum_of_sigmoids(const float* x, float* y, const float* params, const int32_t* cols,
                       float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t,
                       int32_t n_sigmoids, int32_t inverse, int32_t bisection_iterations,
                       float bisection_lim, float offset, float log_scale_postact,
                       int32_t shared_params, int32_t lad_mode, void* stream);

/* Backward of fc_sum_of_sigmoids in the forward direction, all d columns transformed, per-sample rows (what
 * torch.autograd yields for adaptive_sigmoids.py:108-142 + nonlinearities.py:519-552): grad_x [n, d] and grad_params
 * [n, d * (3S + 1)] from grad_y [n, d] and grad_logabsdet [n] (NULL = zeros); closed-form derivatives, the chain through
 * tanh / sigmoid / the renormalised softmax / softplus included. */
int fc_sum_of_sigmoids_backward(const float* x, const float* params, const float* grad_y,
                                const float* grad_logabsdet, float* grad_x, float* grad_params, int64_t n, int32_t d,
                                int32_t n_sigmoids, float log_scale_postact, void* stream);

/* ---- row-per-wavefront bijectors with dense parameters (d <= 512) ------------------------------ */
/* K Householder reflections out -= (out.q_k)(2/|q_k|^2) q_k, k = 0..K-1 (reverse != 0: K-1..0).
 * q: [K, d] shared, or [n, K, d] when per_sample != 0.  logabsdet is identically 0.
 * Replaces _apply_batchwise_transforms[_nodet] (transforms/orthogonal.py:144-194) behind
 * HouseholderSequence / ParametrizedHouseHolder forward/inverse (orthogonal.py:63-141). */
int fc_householder(const float* x, float* y, const float* q, int64_t n, int32_t d,
                   int32_t num_transforms, int32_t per_sample, int32_t reverse, void* stream);

/* y = x + u_hat * tanh(x.w + b); logabsdet = log(1e-7 + |1 + sum_j u_hat_j (1 - tanh^2) w_j|).
 * u_hat is the constrained u (host computes it from w, u: a [1, d] expression); b is a device scalar.
 * per_sample != 0: w, u_hat are [n, d] and b is [n] (ConditionalPlanarTransform, conditional.py:824-838).
 * Replaces PlanarTransform.forward / forward_logabsdet (no_analytic_inv/planar.py:30-49). */
int fc_planar(const float* x, float* y, float* logabsdet, const float* w, const float* u_hat,
              const float* b, int64_t n, int32_t d, int32_t per_sample, void* stream);

/* Backward of fc_planar with batch-shared parameters (per_sample == 0): grad_x [n, d] is written, grad_w [d],
 * grad_u_hat [d] and grad_b [1] are ACCUMULATED (atomic adds of per-wave partial sums: zero them first).
 * grad_logabsdet may be NULL (zeros).  What torch.autograd yields for no_analytic_inv/planar.py:30-49 from u_hat on;
 * the constraint u -> u_hat stays a host-side torch expression on [1, d]. */
int fc_planar_backward(const float* x, const float* grad_y, const float* grad_logabsdet, const float* w,
                       const float* u_hat, const float* b, float* grad_x, float* grad_w, float* grad_u_hat,
                       float* grad_b, int64_t n, int32_t d, void* stream);

/* Backward of fc_householder with batch-shared q [K, d] (per_sample == 0), from the saved OUTPUT y of the forward
 * call (each reflection is its own inverse, so the intermediates are recovered by walking back): grad_x [n, d] is
 * written, grad_q [K, d] is ACCUMULATED (zero it first).  `reverse` as in the forward call.
 * What torch.autograd yields for orthogonal.py:144-194. */
int fc_householder_backward(const float* y, const float* grad_y, const float* q, float* grad_x, float* grad_q,
                            int64_t n, int32_t d, int32_t num_transforms, int32_t reverse, void* stream);

/* Element-wise middle of the backward of fc_sylvester with batch-shared parameters (no_analytic_inv/planar.py:144-166):
 * pre [n, d] = R1 Q^T z + b (recomputed by the caller), grad_act_inout [n, d] = gradient wrt tanh(pre) coming from the
 * R2 / Q product, grad_logabsdet [n] or NULL, r_diag_prod [d] = diag R1 * diag R2.  In place: grad_act_inout becomes the
 * gradient wrt pre including the log-determinant's share; grad_bias [d] and grad_r_diag_prod [d] are ACCUMULATED (zero
 * them first).  The matrix products around it are plain GEMMs and the two Householder sequences use
 * fc_householder / fc_householder_backward (flowconductor_amd.ops._SylvesterFunction). */
int fc_sylvester_mid_backward(const float* pre, float* grad_act_inout, const float* grad_logabsdet,
                              const float* r_diag_prod, float* grad_bias, float* grad_r_diag_prod, int64_t n, int32_t d,
                              void* stream);

/* Dense linear maps with batch-shared [d, d] matrices given TRANSPOSED (a_t[j*d + i] = A[i][j]).
 * mode 0: y = A x + bias                        (linear.py:45-52 cached weight; bias may be NULL)
 * mode 1: y = B (A x) + bias, A = U, B = L      (lu.py:56-68, two F.linear)
 * mode 2: y = A^-1 B^-1 (x - bias), A = U upper, B = L unit-lower (lu.py:70-91, solve_triangular) */
int fc_linear(const float* x, float* y, const float* a_t, const float* b_t, const float* bias,
              int64_t n, int32_t d, int32_t mode, void* stream);

/* Per-sample dense [d, d] matrices M [n, d, d] (row-major, as a hyper-network emits them), d <= 512.
 * mode 0: y = M x                 mode 1: y = M^T x          (ConditionalRotationTransform, conditional.py:374-401)
 * mode 2: y = L (U x), mode 3: y = U^-1 L^-1 x with, as in ConditionalLUTransform (conditional.py:300-346),
 *   L = sp * tril(M, -1) + I,  U = sp * triu(M, 1) + diag(softplus(diag M) + eps),  sp = softplus(scale)
 *   given as `offdiag_scale`; logabsdet (may be NULL) = +/- sum_i log U_ii.
 * HBM-bound: every matrix element is read exactly once, rows coalesced. */
int fc_linear_per_sample(const float* x, float* y, float* logabsdet, const float* m, int64_t n,
                         int32_t d, int32_t mode, float offdiag_scale, float eps, void* stream);

/* Sylvester flow, fused: y = z + Q R2 tanh(R1 Q^T z + bias), Q = num_householder reflections q,
 * logabsdet = sum_j log(1 + (1 - tanh^2(.)_j) * r_diag_prod_j), r_diag_prod = diag(R1)*diag(R2).
 * Shared parameters: r1_t / r2_t are the upper-triangular matrices TRANSPOSED.  per_sample != 0: q [n, M, d],
 * r1_t / r2_t [n, d, d] ROW-MAJOR AND UNTRANSPOSED, as a hyper-network emits them (only the upper triangle of every
 * row is read: half the bytes), bias / r_diag_prod [n, d] (the D-general conditional form, conditional.py:936-953).
 * Replaces SylvesterTransform.forward (no_analytic_inv/planar.py:144-166). */
int fc_sylvester(const float* x, float* y, float* logabsdet, const float* q, const float* r1_t,
                 const float* r2_t, const float* bias, const float* r_diag_prod, int64_t n, int32_t d,
                 int32_t num_householder, int32_t per_sample, void* stream);

/* Shared-weight Sylvester flow as two dense products on the matrix cores (planar.py:144-166 with the batch-
 * independent chains folded into W1 = R1 Q^T and W2 = Q R2, both [d, d] row-major):
 *   y = x + W2 tanh(W1 x + bias),  logabsdet[n] = sum_i log(1 + (1 - tanh^2(.)_i) r_diag_prod_i).
 * d % 32 == 0, d <= 128, n % 16 == 0, x / y / w1 / w2 16-byte aligned.  Products: three-term scaled f16 splits. */
int fc_sylvester_mm(const float* x, float* y, float* logabsdet, const float* w1, const float* w2,
                    const float* bias, const float* r_diag_prod, int64_t n, int32_t d, void* stream);

/* y = W x + bias for a batch-independent dense [d, d] matrix on the matrix cores (same split-f16 products):
 * LULinear / Linear forward with W = L U (lu.py:56-68, linear.py:45-60), a HouseholderSequence folded into its
 * orthogonal matrix (orthogonal.py:63-85).  bias may be NULL.  d % 32 == 0, d <= 128, n % 16 == 0; x / y / w 16-byte aligned. */
int fc_dense_mm(const float* x, float* y, const float* w, const float* bias, int64_t n, int32_t d, void* stream);

/* ---- weight packing on the device ------------------------------------------------------------------------- */
/* f32 nn.Linear tensors -> the scaled two-piece f16 matrix-core fragments the kernels above take (w_frag / wt_frag),
 * their power-of-two unscale factors and the packed biases: one workgroup per scale group, one launch per weight set
 * (training re-packs every step).  `jobs` is a DEVICE array of fc_pack_job. */
#define FC_PACK_FINAL 0      /* fc_rq_spline_fused_general w_frag + bias_pad of one group of 4 dims */
#define FC_PACK_FINAL_T 1    /* fc_rq_fused_linear_backward wt_frag of one group */
#define FC_PACK_HIDDEN 2     /* fc_resnet_hidden_backward w_frag + bias_acc of one layer */
#define FC_PACK_HIDDEN_T 3   /* its wt_frag of one hidden layer */
#define FC_PACK_HIDDEN_T0 4  /* its wt_frag of the initial layer (W0^T) */
typedef struct fc_pack_job {
  const float* w;       /* source matrix [rows, cols], row-major */
  const float* b;       /* source bias or NULL */
  void* frag;           /* f16 fragments of this group: [nks * nt][2 pieces][64 lanes][8] */
  float* unscale;       /* 2^-S of this group */
  float* bias_out;      /* packed bias or NULL */
  int32_t rows, cols;   /* valid extent of w (everything outside reads as 0) */
  int32_t mode;         /* FC_PACK_* */
  int32_t p, pp;        /* FINAL*: parameters per dim P and its padding 4T */
  int32_t nks, nt;      /* fragment image [nks][nt] (FINAL_T: [nt hidden tiles][nks k-steps]) */
  int32_t group;        /* FINAL*: group of 4 dims */
} fc_pack_job;
int fc_pack_fragments(const void* jobs, int32_t num_jobs, void* stream);
int fc_pack_job_bytes(void);

/* ---- multi-GPU: the one collective of the path ------------------------------------------------------ */
/* Batch-sharded log_prob (one process per GPU, contiguous row shards, replicated weights) exchanges
 * only {sum of log_prob, row count}: 16 bytes per evaluation.  The reference has no distributed code; these entries
 * are what SURVEY.md 8b names for the N > 1 path.  RCCL is resolved at run time (the process' already loaded
 * librccl.so.1 first); without it these return hipErrorNotSupported.  Return value: 0, a hipError_t, or
 * 10000 + ncclResult_t.
 *   fc_comm_unique_id   rank 0 fills a 128-byte id and hands it to the other ranks by any host channel
 *   fc_comm_init_rank   collective over all ranks; binds the calling thread's CURRENT device to `rank`
 *   fc_allreduce_loglik sum_count: DEVICE pointer to two float64 {sum, count}, reduced in place (sum) across the
 *                       ranks of `comm`, asynchronous on `stream` (ncclAllReduce, RCCL over xGMI)
 *   fc_comm_init_rank_on_device  the same after hipSetDevice(device) in the calling thread (the current device is per
 *                       host thread: use this one when the set-up runs off the thread that owns the device)
 *   fc_comm_destroy     releases the communicator (collective: every rank's outstanding work must have finished)
 *   fc_comm_abort       tears a communicator down without waiting for the other ranks (failed set-up on a peer) */
#define FC_COMM_UNIQUE_ID_BYTES 128
int fc_comm_unique_id(void* id_out128);
int fc_comm_init_rank(void** comm_out, int32_t nranks, const void* id128, int32_t rank);
int fc_comm_init_rank_on_device(void** comm_out, int32_t nranks, const void* id128, int32_t rank, int32_t device);
int fc_comm_destroy(void* comm);
int fc_comm_abort(void* comm);
int fc_allreduce_loglik(double* sum_count, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWCON_HIP_H_ */
