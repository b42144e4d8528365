// Fused final-Linear + RQ-spline kernel, third structure: split-f16 matrix cores, parameters born in the
// evaluating lane's registers.  gfx950.
//
//   params[n, :] = W h[n, :] + b        (flowcon/nn/nets/resnet.py:91,99, the conditioner's final Linear)
//   y, logabsdet = rq_spline(x, params) (flowcon/transforms/coupling.py:279-293,549-582)
//
// Why not the f32-input MFMA (the first two versions of this kernel, see fc_rq_fused.hip): v_mfma_f32_*_f32 runs at the f32 VALU
// rate and, measured here, does not overlap with VALU work at all -- MFMA-only 0.85 ms, spline-only
// 0.72 ms, both 1.29 ms per 2^20-row launch, whichever way the two streams were interleaved.  The 16-bit
// matrix pipe is 16x faster.  So the f32 product is computed to f32 accuracy on it: after an exact
// power-of-two scaling (per wave for its weight rows, per sample for the h row, so that the row maximum
// sits in [2^10, 2^11)) every value is split into two f16 pieces, x = xh + xl + O(2^-22 max|x|), and
//   W h = Wl hh + Wh hl + Wh hh                    (+ Wl hl <= 2^-22 |W||h|, dropped)
// is accumulated in f32 by v_mfma_f32_16x16x32_f16: 3 terms x 2 k-steps per 16x16 output tile, 3/16 of
// the f32-MFMA cycles.  Error against float64: max 1.7e-7, rms 2.2e-8 of sum|W||h| -- an f32 GEMM's own
// rounding is 4.1e-7 / 3.4e-8 on the same data (tools/probe/split_accuracy.py).  [A three-piece bf16
// split (6 terms, no scaling needed) measured 0.77 ms; a bf16 MFMA costs its full 16 cycles beside a
// VALU-bound stream (tools/probe/mfma_valu_overlap.hip), so halving the MFMA count pays.]
//
// Layout.  The product is taken transposed, P^T = W h^T: A = weight rows (features), B = h^T (samples on
// the columns).  The C layout then gives lane (s = lane & 15, g = lane >> 4) the features 4g..4g+3 of
// sample s; with the weight rows of a tile ordered as (dim g, param 4t + r), six tiles hand that lane all
// 24 (23 + pad) parameters of element (sample s, dim 4w + g) in its own accumulators: no LDS round trip
// for the parameters (the f32 kernels moved 98 KB of them through LDS per 32-row tile).  The scaling is
// undone for free: logit = fma(acc, 2^-(S_w + T_s) / sqrt(hidden), bias / sqrt(hidden)).
//
// One 512-thread workgroup per CU walks 32-row tiles.  Wave w owns transformed dims 4w..4w+3 for the whole
// kernel: both weight pieces stay in 96 VGPRs.  Per 16-sample block a wave issues 36 MFMAs for the NEXT
// block and evaluates one element per lane of the CURRENT block, hand-interleaved: the evaluation is
// generated straight-line code with 36 hook points (tools/gen_fused_eval.py), hook n issues MFMA n and pins
// it with a sched_barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "fc_tile.h"
#include "fc_math.h"
#include "fc_rq_op.h"
#include "fc_rq_fused.h"
#include "fc_split.h"
#include "fc_lane.h"
#include "../../include/flowcon_hip.h"

// tools/probe/build_fused_variants.sh only: ablation builds (1 no evaluation, 4 no MFMAs, 8 loads from L2,
// 16 clock stamps, 32 no h conversion).
#ifndef FC_ABL
#define FC_ABL 0
#endif

namespace fc {

#ifndef FC_HOOK_MASK
#define FC_HOOK_MASK 0   // sched_barrier mask at each MFMA hook: 0 pins everything
#endif
#define FC_F2 f2
#ifndef FC_PRIO_PERIOD
#define FC_PRIO_PERIOD 36   // hooks per priority sawtooth of 4 levels (measured: 4, 8 slower; 12 +1.2 %; 36 = 72; 144 slower)
#endif

constexpr int kCt3 = 6;                       // 16-feature tiles per wave: 4 dims x 24 padded params
constexpr int kHB = kH + 16;                  // f16 per h row in LDS: 160 B.  ds_read_b128 is served in four NON-contiguous 16-lane
                                              // groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) on 64 banks: rows of a
                                              // group are conflict-free iff the stride is 32 mod 64 bytes (tools/lds_conflicts.py);
                                              // the 144 B of rounds 1-2 were 2-way on every fragment read (24 % of the kernel's LDS cycles)
constexpr int kKnotFloats = (kK + 1) * 64 * 2;     // per wave: [slot][lane] (x, y) knots
constexpr int kDerFloats = (kK + 1) * 64;          // per wave: [slot][lane] derivative logits
constexpr int kTabBytes = 8 * (kKnotFloats + kDerFloats) * 4;

// LDS image for tiles of R rows (R = 64 unless the x tile of a wide input does not fit, then 32)
template <int R>
struct Fused3Lds {
  static constexpr int kHPiece = R * kHB;                  // f16 per piece per buffer
  static constexpr int kHbufBytes = 2 * 2 * kHPiece * 2;   // [buf][piece][row][kHB] f16
  static constexpr int kLpartBytes = 3 * 8 * R * 4;        // [ring of 3][wave][row]
  static constexpr int kHscaleBytes = 3 * R * 4;           // [ring of 3][row] 2^-T of the h row
  static constexpr size_t bytes(int d) {
    return (size_t)kHbufBytes + kLpartBytes + kHscaleBytes + kTabBytes + 3 * R * (d + 4) * 4 + kDt * 4;
  }
};

size_t fused3_lds_bytes(int d, int rows) { return rows == 64 ? Fused3Lds<64>::bytes(d) : Fused3Lds<32>::bytes(d); }

// R: rows per tile (64, or 32), NB = R / 16 sample blocks per tile; XV: float4 of the x tile per thread
// kFull: all 32 dims (every wave has spline work; no per-wave guards in the loop); kPadX: D % 4 == 0
template <bool kInv, int R, int XV, bool kFull, bool kPadX>
__global__ __launch_bounds__(512) void rq_fused_linear_kernel3(RQOp<kK> op, FusedArgs a) {
  using L = Fused3Lds<R>;
  constexpr int kHPiece = L::kHPiece;
  constexpr int NB = R / 16;      // sample blocks per tile
  constexpr int HV = R / 32;      // float4 of the h tile per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  _Float16* hbuf = reinterpret_cast<_Float16*>(smem3);                               // [2][2][R][kHB]
  float* lpart = reinterpret_cast<float*>(smem3 + L::kHbufBytes);                    // [2][8][R]
  float* hscale = lpart + 3 * 8 * R;                                                  // [3][R]
  float* tabs = hscale + 3 * R;                                                       // [8 waves][knots | derivs]
  float* xbuf = tabs + kTabBytes / 4;                                                 // [2][R][D + 4]
  // x rows in LDS: padded by 4 floats when D is a multiple of 4 (bank spread, float4 pieces stay inside a row);
  // for other D the tile is kept as the contiguous [R * D] block it is in memory (an odd stride spreads the banks
  // by itself) -- R * D is always a whole number of float4
  const int D = a.D, XS = kPadX ? D + 4 : D;
  int* cs = reinterpret_cast<int*>(xbuf + 3 * R * (D + 4));                           // [kDt]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s16 = lane & 15, g = lane >> 4;
  const int64_t stride = gridDim.x;
  const int64_t tile0 = blockIdx.x;
  if (tile0 >= a.tiles) return;
  if (tid < kDt) cs[tid] = tid < a.dt ? a.cols[tid] : 0;
  const int WD = kFull ? 8 : (a.dt + 3) >> 2;    // dim groups = waves with spline work
  const bool dim_ok = kFull || 4 * wave + g < a.dt;   // (the last group may be partly padding)
  const bool active = kFull || wave < WD;
  const float inv_div = op.inv_div;
#if FC_ABL & 16
  const uint64_t stamp_r_entry = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- resident weights ---------------------------------------------------------------------------------
  // A operand of tile t, k-step ks: lane holds W[row(t, lane & 15)][k = 32 ks + 8 (lane >> 4) + j], j < 8,
  // where row(t, rho) = padded feature (dim 4w + (rho >> 2)) * 24 + 4t + (rho & 3).
  // Accumulator slot 4t + r of a lane holds parameter slot_param(4t + r) of its element: width and height logit i sit
  // in the adjacent slots 2i, 2i + 1, so the packed (width, height) arithmetic of the evaluation reads register pairs
  // as they are (with the natural order the compiler assembles every pair with two moves).
  constexpr auto slot_param = [](int s) { return s < 16 ? ((s & 1) ? 8 + (s >> 1) : (s >> 1)) : s; };
  f16x8 wh[kCt3][2], wl[kCt3][2];
  float w_unscale;
  {
    float wv[kCt3][2][8];
    float wmax = 0.f;
#pragma unroll
    for (int t = 0; t < kCt3; ++t) {
      // (padding rows -- parameter 23 of a dim, dims beyond dt -- are zeros: read as such whether or not the arrays
      //  carry them, a.wrows = 24 or 23)
      const int wdim = 4 * (active ? wave : 0) + (s16 >> 2), wprm = slot_param(4 * t + (s16 & 3));
      const bool wreal = wprm < kPP - 1 && wdim < a.dt;
      const int row = wreal ? wdim * a.wrows + wprm : 0;
      const float wmask = wreal ? 1.f : 0.f;     // (a multiplication, not a guarded load: the loads stay two
                                                 //  unconditional 16-byte requests per fragment, all in flight together)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float4* src = reinterpret_cast<const float4*>(a.wpad + (int64_t)row * kH + 32 * ks + 8 * g);
        const float4 v0 = src[0], v1 = src[1];
        const float v[8] = {v0.x * wmask, v0.y * wmask, v0.z * wmask, v0.w * wmask,
                            v1.x * wmask, v1.y * wmask, v1.z * wmask, v1.w * wmask};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          wv[t][ks][j] = v[j];
          wmax = fmaxf(wmax, fabsf(v[j]));
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    float w_scale;
    pow2_scale(wmax, w_scale, w_unscale);
#pragma unroll
    for (int t = 0; t < kCt3; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          _Float16 ph, pl;
          split2(wv[t][ks][j] * w_scale, ph, pl);
          wh[t][ks][j] = ph;
          wl[t][ks][j] = pl;
        }
  }
  // bias of lane (s, g), register r of tile t: feature (dim 4w + g, param 4t + r), resident (24 registers).
  // The width / height logits are divided by sqrt(hidden_features) (coupling.py:565-566) and only ever feed a
  // softmax, evaluated as exp2 of differences: division and log2(e) are folded into the fma that also undoes the
  // scaling, so their bias is kept pre-multiplied by log2(e) / sqrt(hidden_features).
  const float wh_mul = (float)((double)inv_div * 1.4426950408889634);
  f32x4 bw[kCt3];
#pragma unroll
  for (int t = 0; t < kCt3; ++t) {
    const int bdim = 4 * (active ? wave : 0) + g;
    const float* bsrc = a.bias + bdim * a.wrows;
    const float m = t < 4 ? wh_mul : op.q.beta;   // params 0..15 are widths and heights; the derivative logits only ever
                                                  // enter softplus(beta u): beta folded in like the factors above
    const bool bdim_ok = bdim < a.dt;
    auto bias_of = [&](int slot) {      // unconditional load from a valid slot, masked by a multiplication
      const int prm = slot_param(slot);
      const bool real = prm < kPP - 1 && bdim_ok;
      return (real ? bsrc[prm] : a.bias[0]) * (real ? m : 0.f);
    };
    bw[t] = f32x4{bias_of(4 * t), bias_of(4 * t + 1), bias_of(4 * t + 2), bias_of(4 * t + 3)};
  }
  // Knot constants of fc_rq_fused3_eval.inc (x: widths axis, y: heights axis), formed in double once per kernel:
  // knot_{i+1} = kc_i + (sum of the first i + 1 softmax numerators) * (sc1 / their total) for the lower half, and
  // kc_i - (sum of the last K - 1 - i numerators) * (sc1 / total) for the upper half.
  const double span_x = (double)op.q.right - (double)op.q.left, span_y = (double)op.q.top - (double)op.q.bottom;
  const f2 sc1 = {(float)(span_x * (double)op.q.cw), (float)(span_y * (double)op.q.ch)};
  auto knot_const = [&](int i) {
    if (i < kK / 2)
      return f2{(float)((double)op.q.left + span_x * (double)op.q.min_w * (double)(i + 1)),
                (float)((double)op.q.bottom + span_y * (double)op.q.min_h * (double)(i + 1))};
    return f2{(float)((double)op.q.right - span_x * (double)op.q.min_w * (double)(kK - 1 - i)),
              (float)((double)op.q.top - span_y * (double)op.q.min_h * (double)(kK - 1 - i))};
  };
  const f2 kc0 = knot_const(0), kc1 = knot_const(1), kc2 = knot_const(2), kc3 = knot_const(3), kc4 = knot_const(4),
           kc5 = knot_const(5), kc6 = knot_const(6);
  static_assert(kK == 8, "kc0 .. kc6: the generated evaluation is for 8 bins");

  // Lane-private bin tables (fc_rq_fused3_eval.inc): slots 0 and K are the interval ends / the linear-tail
  // derivative constant (rational_quadratic.py:33-36) and never change; slots 1..K-1 are rewritten per element.
  // [slot][lane] layout: every access of a wave is conflict-free whatever the lanes' bin indices are.
  float* ktab = tabs + wave * (kKnotFloats + kDerFloats) + lane * 2;
  float* dtab = tabs + wave * (kKnotFloats + kDerFloats) + kKnotFloats + lane;
  *reinterpret_cast<f2*>(ktab) = f2{op.q.left, op.q.bottom};
  *reinterpret_cast<f2*>(ktab + kK * 128) = f2{op.q.right, op.q.top};
  dtab[0] = op.q.tail_const * op.q.beta;
  dtab[kK * 64] = op.q.tail_const * op.q.beta;

  uint32_t err = 0;
  int mycol = 0;     // this lane's column of x (cs[] is read ONCE, behind the prologue's barrier: per step it was an LDS round trip in front of the x read)
  const int xvec = R * D / 4;     // float4 per x tile: thread tid owns slots tid + 512 k, k < XV
  // (named scalars, not arrays: hipcc keeps register arrays that are written under `if (has_next)` in scratch)
  // kCarry: x pieces beyond the first ride in registers across the tile.  Only the headline shape (d_t = 32, D <= 64) has room
  // for them (246-251 registers, no scratch); the other instantiations spilled 12-144 B per lane and read those pieces when
  // they park the tile instead (one exposed L2 / HBM latency per tile for layers the kernel was not tuned for).
  constexpr bool kCarry = kFull && XV <= 2;
  float4 hv0, hv1, xv0, xv1, xv2, xv3;
  hv0 = hv1 = xv0 = xv1 = xv2 = xv3 = float4{0.f, 0.f, 0.f, 0.f};
  int64_t fetched = tile0;
  auto fetch = [&](int64_t t) __attribute__((always_inline)) {
    fetched = t;
    if (FC_ABL & 8) t = tile0;   // ablation: every tile's loads hit in L2
    const float4* hg = reinterpret_cast<const float4*>(a.h + t * R * kH);
    hv0 = hg[tid];
    if constexpr (HV > 1 && kFull) hv1 = hg[tid + 512];
    const float4* xg = reinterpret_cast<const float4*>(a.x + t * R * D);
    xv0 = xg[tid < xvec ? tid : 0];
    if constexpr (kCarry) {
      if constexpr (XV > 1) xv1 = xg[tid + 512 < xvec ? tid + 512 : 0];
      if constexpr (XV > 2) xv2 = xg[tid + 1024 < xvec ? tid + 1024 : 0];
      if constexpr (XV > 3) xv3 = xg[tid + 1536 < xvec ? tid + 1536 : 0];
    }
  };
  auto xslot = [&](int buf, int i) __attribute__((always_inline)) {   // float4 index i of a [R, D] tile -> its LDS position
    if constexpr (!kPadX) return reinterpret_cast<float4*>(xbuf + buf * R * XS + 4 * i);
    const int e = i * 4, r = e / D, c = e - r * D;
    return reinterpret_cast<float4*>(xbuf + (buf * R + r) * XS + c);
  };
  // thread tid holds h[row (tid >> 4) + 32 k][4 (tid & 15) ..]: the 16 threads of a row are 16 adjacent lanes
  auto park_h = [&](int buf, int xbuf3, int k, const float4& hvk) __attribute__((always_inline)) {
    const int c = (tid & 15) * 4, r = (tid >> 4) + 32 * k;
#if FC_ABL & 32   // ablation: the tile arrives already split (no conversion arithmetic): upper bound of what a producer-side split buys
    {
      _Float16* dst = hbuf + (buf * 2 * R + r) * kHB + c;
      *reinterpret_cast<u32x2*>(dst) = u32x2{__float_as_uint(hvk.x) & 0x3bff3bffu, __float_as_uint(hvk.y) & 0x3bff3bffu};
      *reinterpret_cast<u32x2*>(dst + kHPiece) = u32x2{__float_as_uint(hvk.z) & 0x3bff3bffu, __float_as_uint(hvk.w) & 0x3bff3bffu};
      if ((tid & 15) == 0) hscale[xbuf3 * R + r] = 1.f;
      return;
    }
#endif
    const float v[4] = {hvk.x, hvk.y, hvk.z, hvk.w};
    const float m = row16_allmax(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    float sc, un;
    pow2_scale(m, sc, un);
    u32x2 p0, p1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint32_t ph, pl;
      split2_pair(v[2 * j], v[2 * j + 1], sc, ph, pl);
      p0[j] = ph; p1[j] = pl;
    }
    _Float16* dst = hbuf + (buf * 2 * R + r) * kHB + c;
    *reinterpret_cast<u32x2*>(dst) = p0;
    *reinterpret_cast<u32x2*>(dst + kHPiece) = p1;
    if ((tid & 15) == 0) hscale[xbuf3 * R + r] = un;
  };
  // h tile -> hbuf[hb2] (ring of 2), its row scales and the x tile -> ring slot x3 (ring of 3)
  auto park = [&](int hb2, int x3) __attribute__((always_inline)) {
    park_h(hb2, x3, 0, hv0);
    if constexpr (HV > 1 && !kFull)      // (the generic variants carry one piece of h and one of x across the tile)
      hv1 = reinterpret_cast<const float4*>(a.h + ((FC_ABL & 8) ? tile0 : fetched) * R * kH)[tid + 512];
    if constexpr (HV > 1) park_h(hb2, x3, 1, hv1);
    if (tid < xvec) *xslot(x3, tid) = xv0;
    if constexpr (!kCarry && XV > 1) {
      const float4* xg = reinterpret_cast<const float4*>(a.x + ((FC_ABL & 8) ? tile0 : fetched) * R * D);
      xv1 = xg[tid + 512 < xvec ? tid + 512 : 0];
      if constexpr (XV > 2) xv2 = xg[tid + 1024 < xvec ? tid + 1024 : 0];
      if constexpr (XV > 3) xv3 = xg[tid + 1536 < xvec ? tid + 1536 : 0];
    }
    if constexpr (XV > 1) if (tid + 512 < xvec) *xslot(x3, tid + 512) = xv1;
    if constexpr (XV > 2) if (tid + 1024 < xvec) *xslot(x3, tid + 1024) = xv2;
    if constexpr (XV > 3) if (tid + 1536 < xvec) *xslot(x3, tid + 1536) = xv3;
  };
  // B operand (h^T piece `hp`, k-step ks) of block `blk` in buffer `hb`:
  // lane holds h[sample 16 blk + (lane & 15)][k = 32 ks + 8 (lane >> 4) + j]
  auto hfrag = [&](int hb, int blk, int hp, int ks) {
    return *reinterpret_cast<const f16x8*>(hbuf + ((hb * 2 + hp) * R + 16 * blk + s16) * kHB + 32 * ks + 8 * g);
  };

  // MFMA number n of a block: term n / 12 (0: Wl hh, 1: Wh hl, 2: Wh hh -- small products first),
  // k-step (n / 6) % 2, tile n % 6
  auto mfma_n = [&](auto N, f32x4 (&acc)[kCt3], const f16x8& bcur) {
    constexpr int n = decltype(N)::value, term = n / 12, ks = (n / 6) % 2, t = n % 6;
    const f16x8& aop = term == 0 ? wl[t][ks] : wh[t][ks];
    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, bcur, n < kCt3 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[t], 0, 0, 0);
  };
  constexpr auto term_h = [](int term) { return term == 1 ? 1 : 0; };   // h piece of a term

#define FC_ALL36(M)                                                                                          \
  M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18)     \
  M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31) M(32) M(33) M(34) M(35)

  // Prologue only: the parameters of block `blk` of the tile in buffer `hb`, nothing to evaluate yet.
  auto produce_only = [&](f32x4 (&acc)[kCt3], int hb, int blk) {
    f16x8 bcur = hfrag(hb, blk, term_h(0), 0);
    auto one = [&](auto N) {
      constexpr int n = decltype(N)::value;
      if constexpr (n % 6 == 0 && n > 0) bcur = hfrag(hb, blk, term_h(n / 12), (n / 6) % 2);
      mfma_n(N, acc, bcur);
    };
#define FC_S(n) one(std::integral_constant<int, n>{});
    FC_ALL36(FC_S)
#undef FC_S
  };

  // One step: evaluate this lane's element of block `cblk` of the tile in buffer `xb` from the accumulators
  // `pa`, and produce into `acc` the accumulators of block `pblk` of the tile in buffer `hb`.
  // (`lad_out`: this lane's logabsdet term; the sums over a sample's four lanes are taken once per tile, after_steps())
  // (measured and dropped: the NEXT step's x element and row scale read late in this step's evaluation, so that a step does
  //  not open with an LDS round trip -- 40 B of spills, 12 400 -> 13 400 cycles per tile)
  auto step = [&](const f32x4 (&pa)[kCt3], int xb, int cblk, f32x4 (&acc)[kCt3], int hb, int pblk, float& lad_out) {
    float* xr = xbuf + (xb * R + 16 * cblk + s16) * XS + mycol;
    const float x = *xr;
    const float c_d = hscale[xb * R + 16 * cblk + s16] * w_unscale;   // undoes both scalings (a power of two)
    const float c_wh = c_d * wh_mul;        // (c_d is a power of two: the product is exact)
    const float c_ud = c_d * op.q.beta;
    // h^T fragments are read one group of 6 MFMAs ahead of their use
    f16x8 bcur, bnext = hfrag(hb, pblk, term_h(0), 0);
    auto hook = [&](auto N) {
      constexpr int n = decltype(N)::value;
      if constexpr (n % 6 == 0) {
        bcur = bnext;
        if constexpr (n + 6 < 36) bnext = hfrag(hb, pblk, term_h((n + 6) / 12), ((n + 6) / 6) % 2);
      }
      if constexpr (!(FC_ABL & 4)) mfma_n(N, acc, bcur);
      // Fairness between the two waves of a SIMD.  VALU issue goes to the higher s_setprio, then to the OLDER
      // wave: left alone, waves 0-3 run each step unimpeded, wait ~3700 cycles per tile at the barriers, and
      // waves 4-7 finish alone at single-wave issue rate.  A priority that falls as a wave advances (a
      // sawtooth over FC_PRIO_PERIOD hooks) always favours the wave that is behind, so both reach the barrier together.
      if constexpr (n % (FC_PRIO_PERIOD / 4) == FC_PRIO_PERIOD / 4 - 1)
        __builtin_amdgcn_s_setprio(3 - ((n + 1) % FC_PRIO_PERIOD) / (FC_PRIO_PERIOD / 4));
      __builtin_amdgcn_sched_barrier(FC_HOOK_MASK);
    };
    const RQParams& q = op.q;
    const float inv_beta = op.inv_beta;   // softplus(x, beta) = log1p(exp(beta x)) * (1 / beta): exact at beta = 1
    float y, lad;
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_sched_barrier(0);
#define FC_HOOK(n) hook(std::integral_constant<int, n>{});
#define FC_WH_SLOT(i) ((i) < 8 ? 2 * (i) : 2 * ((i) - 8) + 1)
#define FC_WH(i) __builtin_fmaf(pa[FC_WH_SLOT(i) >> 2][FC_WH_SLOT(i) & 3], c_wh, bw[FC_WH_SLOT(i) >> 2][FC_WH_SLOT(i) & 3])
#define FC_UD(j) __builtin_fmaf(pa[((j) + 16) >> 2][((j) + 16) & 3], c_ud, bw[((j) + 16) >> 2][((j) + 16) & 3])
#define FC_KNOT_ST(slot, v) *reinterpret_cast<f2*>(ktab + (slot) * 128) = (v)
#define FC_KNOT_LD(i, off) *reinterpret_cast<const f2*>(ktab + ((i) + (off)) * 128)
#define FC_DER_ST(slot, v) dtab[(slot) * 64] = (v)
#define FC_DER_LD(i, off) dtab[((i) + (off)) * 64]
#define FC_COUNT_GE(count, a, b)                                                              \
  do {                                                                                        \
    const float fc_b_ = (b);                                                                  \
    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(count) : "v"(a), "v"(fc_b_) : "vcc"); \
  } while (0)
#if FC_ABL & 1
    FC_ALL36(FC_HOOK)
    y = x + (FC_WH(0) + FC_WH(5) + FC_WH(10) + FC_WH(15) + FC_UD(0) + FC_UD(5)) * 0.f + q.left * 0.f;
    lad = 0.f;
#else
#include "fc_rq_fused3_eval.inc"
#endif
#undef FC_COUNT_GE
#undef FC_DER_LD
#undef FC_DER_ST
#undef FC_KNOT_LD
#undef FC_KNOT_ST
#undef FC_UD
#undef FC_WH
#undef FC_WH_SLOT
#undef FC_HOOK
    if (dim_ok) *xr = y;
    lad_out = dim_ok ? lad : 0.f;
  };
  // logabsdet partials of this wave's 4 dims for the blocks of a tile: lanes s, s+16, s+32, s+48 hold the same sample.  The
  // four blocks' terms are merged in three swap + add steps (rows4_sum4) and every lane stores one partial, instead of a
  // two-step all-reduce with selects and a quarter-wave store per block.
  auto after_steps = [&](int xb, const float (&lb)[NB]) __attribute__((always_inline)) {
    if constexpr (NB == 4) {     // lb[0] already holds the merge of blocks 0 and 1
      const float l = lane_merge16(lb[0], lane_merge32(lb[2], lb[3]));       // rows: blocks 0, 2, 1, 3 (rows4_sum4)
      lpart[(xb * 8 + wave) * R + 16 * rows4_sum4_index(g) + s16] = l;
    } else {
      const float m = lane_merge32(lb[0], lb[1]);       // lanes 0-31: block 0 (rows 0 + 2, 1 + 3), lanes 32-63: block 1
      const float l = m + lane_xor16(m, lane);
      if ((g & 1) == 0) lpart[(xb * 8 + wave) * R + 16 * (g >> 1) + s16] = l;
    }
  };

#if FC_ABL & 16   // ablation: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped around the loop
  const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#if FC_ABL & 16   // cycles each wave spends in the phases of the loop / waiting at its two barriers
  uint64_t barrier_wait = 0, phase_cyc[6] = {0, 0, 0, 0, 0, 0}, phase_t = __builtin_amdgcn_s_memtime();
  // phase k ends at FC_PHASE(k): 0 loop overhead + fetch, 1 steps 0..NB-2, 2 park, 3 barrier, 5 write-out, 4 last step
#define FC_PHASE(k)                                              \
  do {                                                           \
    const uint64_t now = __builtin_amdgcn_s_memtime();           \
    phase_cyc[k] += now - phase_t;                               \
    phase_t = now;                                               \
  } while (0)
#define FC_TIMED_BARRIER()                                        \
  do {                                                            \
    const uint64_t b0 = __builtin_amdgcn_s_memtime();             \
    __syncthreads();                                              \
    barrier_wait += __builtin_amdgcn_s_memtime() - b0;            \
  } while (0)
#else
#define FC_TIMED_BARRIER() __syncthreads()
#define FC_PHASE(k)
#endif
  f32x4 acc0[kCt3], acc1[kCt3];
#pragma unroll
  for (int t = 0; t < kCt3; ++t) acc0[t] = acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Rings: the h tile is double-buffered; the x tile, its row scales and the logabsdet partials live in a ring
  // of three, because the results of a tile leave only after the NEXT tile's barrier (one barrier per tile):
  //   iteration i:  steps 0..NB-2 of tile i | park tile i+1 | BARRIER | write out tile i-1 | last step of tile i
  // Every thread writes out exactly the float4 slots it parks, so the ring needs no further synchronisation.
  auto write_out = [&](int64_t t, int x3) __attribute__((always_inline)) {
    float4* yg = reinterpret_cast<float4*>(a.y + t * R * D);
#pragma unroll
    for (int k = 0; k < XV; ++k)
      if (tid + 512 * k < xvec) yg[tid + 512 * k] = *xslot(x3, tid + 512 * k);
    if (tid < R) {
      const float* lp = lpart + x3 * 8 * R + tid;
      float l = lp[0];
#pragma unroll
      for (int w = 1; w < 8; ++w)
        if (w < WD) l += lp[w * R];
      // running total of the composite (base.py:51 `total_logabsdet += logabsdet`) or a fresh value
      a.logabsdet[t * R + tid] = a.accumulate ? a.logabsdet[t * R + tid] + l : l;
    }
  };
  fetch(tile0);
  park(0, 0);
  __syncthreads();
  mycol = cs[(4 * wave + g) & (kDt - 1)];
  if (active) produce_only(acc0, 0, 0);   // block 0 of the first tile
  int hb = 0, x3 = 0;          // ring slots of the current tile
  int64_t prev_tile = -1;
  for (int64_t tile = tile0; tile < a.tiles; tile += stride) {
    const bool has_next = tile + stride < a.tiles;
    const int x3n = x3 == 2 ? 0 : x3 + 1, x3p = x3 == 0 ? 2 : x3 - 1;
    FC_PHASE(0);
    if (has_next) fetch(tile + stride);
    // Steps 0 .. NB-2: evaluate block j of `tile`, produce its block j + 1.  (acc0 / acc1 alternate; NB is
    // even, so every tile starts with its block 0 in acc0.)
    float lb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) lb[b] = 0.f;
    if (active) {
      step(acc0, x3, 0, acc1, hb, 1, lb[0]);
      if constexpr (NB == 4) {
        step(acc1, x3, 1, acc0, hb, 2, lb[1]);
        lb[0] = lane_merge32(lb[0], lb[1]);      // (merged as soon as both exist: one value less across the next steps)
        step(acc0, x3, 2, acc1, hb, 3, lb[2]);
      }
    }
    FC_PHASE(1);
    if (has_next) park(hb ^ 1, x3n);
    FC_PHASE(2);
    FC_TIMED_BARRIER();
    FC_PHASE(3);
    if (prev_tile >= 0) write_out(prev_tile, x3p);   // complete since every wave passed this barrier
    FC_PHASE(5);
    // Last step: evaluate block NB-1, produce block 0 of the next tile (unconditional: on the last tile the
    // MFMAs work on stale h rows into accumulators nobody reads -- a branch would split the interleaved block).
    if (active) {
      step(acc1, x3, NB - 1, acc0, hb ^ 1, 0, lb[NB - 1]);
      after_steps(x3, lb);
    }
    FC_PHASE(4);
    prev_tile = tile;
    hb ^= 1;
    x3 = x3n;
  }
  __syncthreads();
  if (prev_tile >= 0) write_out(prev_tile, x3 == 0 ? 2 : x3 - 1);
#if FC_ABL & 16   // the stamps overwrite two outputs of the workgroup's first tile: probe builds only
  if (tid == 0) {
    a.y[tile0 * R * D] = (float)(__builtin_amdgcn_s_memtime() - stamp_c0);
    a.y[tile0 * R * D + 1] = (float)(__builtin_amdgcn_s_memrealtime() - stamp_r0);
    a.y[tile0 * R * D + 2] = (float)(stamp_r0 - stamp_r_entry);                 // prologue, 10 ns ticks
    a.y[tile0 * R * D + 3] = (float)(stamp_r_entry & 0xffffff);                  // entry time (for launch skew)
  }
  if (lane == 0) {
    a.y[tile0 * R * D + 4 + wave] = (float)barrier_wait;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.y[tile0 * R * D + 12 + wave * 6 + k] = (float)phase_cyc[k];
  }
#endif
  if (err && a.err) atomicOr(a.err, err);
}
#undef FC_ALL36
#undef FC_TIMED_BARRIER
#undef FC_PHASE

template <bool kInv, int R, int XV, bool kFull, bool kPadX>
static hipError_t launch_cfg(const RQOp<kK>& op, const FusedArgs& a, unsigned grid, hipStream_t stream) {
  const size_t lds = Fused3Lds<R>::bytes(a.D);
  if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(
      attr, reinterpret_cast<const void*>(&rq_fused_linear_kernel3<kInv, R, XV, kFull, kPadX>), 160 * 1024);
  if (ea != hipSuccess) return ea;
  hipLaunchKernelGGL((rq_fused_linear_kernel3<kInv, R, XV, kFull, kPadX>), dim3(grid), dim3(512), lds, stream, op, a);
  return hipGetLastError();
}

template <bool kInv, int R, int XV>
static hipError_t launch_one(const RQOp<kK>& op, const FusedArgs& a, unsigned grid, hipStream_t stream) {
  if (a.D & 3) return launch_cfg<kInv, R, XV, false, false>(op, a, grid, stream);   // unpadded rows: generic variant
  return a.dt == kDt ? launch_cfg<kInv, R, XV, true, true>(op, a, grid, stream)
                     : launch_cfg<kInv, R, XV, false, true>(op, a, grid, stream);
}

// `a.tiles` counts tiles of `rows` rows (64 or 32)
hipError_t launch_fused3(const RQOp<kK>& op, const FusedArgs& a, int rows, unsigned grid, hipStream_t stream) {
  const bool inv = op.q.inverse != 0;
  if (rows == 64) {
    if (a.D <= 32) return inv ? launch_one<true, 64, 1>(op, a, grid, stream) : launch_one<false, 64, 1>(op, a, grid, stream);
    if (a.D <= 64) return inv ? launch_one<true, 64, 2>(op, a, grid, stream) : launch_one<false, 64, 2>(op, a, grid, stream);
    return inv ? launch_one<true, 64, 4>(op, a, grid, stream) : launch_one<false, 64, 4>(op, a, grid, stream);
  }
  if (a.D <= 64) return inv ? launch_one<true, 32, 1>(op, a, grid, stream) : launch_one<false, 32, 1>(op, a, grid, stream);
  return inv ? launch_one<true, 32, 2>(op, a, grid, stream) : launch_one<false, 32, 2>(op, a, grid, stream);
}

}  // namespace fc
