// Fused final-Linear + RQ-spline kernel, fourth structure: kernel 3 (fc_rq_fused3.hip) for bin counts other than 8,
// written once and compiled per shape (one translation unit per bin count and tail mode, fc_rq_fused4_k<K>[_box].hip, which
// defines FC_F4_K, FC_F4_TAILS, FC_F4_NAME and FC_F4_EVAL_INC before including this file).  hidden_features = 64, linear
// tails or none (coupling.py:543-547: 3K - 1 or 3K + 1 parameters per dim), up to 32 transformed dims.  gfx950.
//
//   params[n, :] = W h[n, :] + b        (flowcon/nn/nets/resnet.py:91,99; num_bins defaults to 10, coupling.py:507)
//   y, logabsdet = rq_spline(x, params) (flowcon/transforms/coupling.py:279-293,549-582; rational_quadratic.py:13-181)
//
// What kernel 3 keeps in registers does not fit at K = 10: 29 parameters per dim are 8 output tiles per wave (4 dims x
// 32 padded parameters), both weight pieces 128 registers, two accumulator sets 64, the bias 32.  Differences:
//   * ONE accumulator set.  The evaluation reads every raw parameter in its first tenth (the 2K width / height logits
//     into the softmax registers, the K - 1 derivative logits into the lane's LDS table); from there on the
//     accumulators are dead, and the 6 CT MFMAs of the NEXT block are issued into them, hooked into the rest of the
//     evaluation (tools/gen_fused_eval.py --bins K places the hooks after the last read).
//   * the bias lives in LDS ([wave][g][slot], pre-multiplied like kernel 3's), read back as CT 16-byte loads per element;
//   * the weights are not split in the kernel: the resident fragments are loaded from the packed image the general
//     kernel streams (fc_pack_fragments FC_PACK_FINAL: scaled by a power of two per group of 4 dims, two f16 pieces),
//     each lane picking the fragment rows of ITS accumulator slots -- width and height logit i in adjacent slots 2i,
//     2i + 1, so the packed (width, height) arithmetic reads register pairs as they are;
//   * 32-row tiles (the lane-private bin tables take 8 (K + 1) 768 bytes: 66 KB at K = 10, which rules out kernel 3's
//     64 rows).  48-row tiles -- one accumulator set does not need an even number of blocks -- were measured and are no
//     faster: the tile hand-over (park, barrier, write-out) is work per ROW, 40 of the 244 cycles a row takes at either
//     size (tools/probe/fused4_clock.py).
// Everything else -- transposed product, split-f16 terms, rings, one barrier per tile, priority sawtooth -- is kernel 3's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
#include "fc_device.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "fc_rq_fused_general.h"
#include "fc_rq_op.h"
#include "fc_split.h"
#include "fc_tile.h"

#define FC_F4_CAT2(a, b) a##b
#define FC_F4_CAT(a, b) FC_F4_CAT2(a, b)

namespace fc {
namespace FC_F4_CAT(f4, FC_F4_NAME) {

#define FC_F2 f2
constexpr int K = FC_F4_K;
constexpr bool kTails = FC_F4_TAILS != 0;
constexpr int P = kTails ? 3 * K - 1 : 3 * K + 1;   // parameters per dim
constexpr int CT = (P + 3) / 4;         // 16-feature tiles per wave: 4 dims x 4 CT padded parameters
constexpr int PP = 4 * CT;
constexpr int NM = 6 * CT;              // MFMAs per 16-sample block: 3 split terms x 2 k-steps x CT tiles
constexpr int R = 32;                   // rows per tile
constexpr int kHB = 64 + 16;            // f16 per h row in LDS: 160 B (conflict-free ds_read_b128, see fc_rq_fused3.hip)
constexpr int kBiasRow = 36;            // floats per (wave, g) bias row: 144 B apart, the four rows a wave reads share no bank
constexpr int kKnotFloats = (K + 1) * 64 * 2;
constexpr int kDerFloats = (K + 1) * 64;
constexpr int kTabBytes = 8 * (kKnotFloats + kDerFloats) * 4;
constexpr int kHPiece = R * kHB;
constexpr int kHbufBytes = 2 * 2 * kHPiece * 2;   // [buf][piece][row][kHB] f16
constexpr int kLpartBytes = 3 * 8 * R * 4;        // [ring of 3][wave][row]
constexpr int kHscaleBytes = 3 * R * 4;
constexpr int kBiasBytes = 8 * 4 * kBiasRow * 4;
static_assert(PP <= 32 && K <= 11 && K >= 4, "accumulator slots / knot constants");

constexpr size_t lds_bytes(int d) {
  return (size_t)kHbufBytes + kLpartBytes + kHscaleBytes + kTabBytes + kBiasBytes + (size_t)3 * R * (d + 4) * 4 + 32 * 4;
}

template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

// XV: float4 of the x tile per thread; kFull: 32 transformed dims (every wave has spline work); kPadX: D % 4 == 0
template <bool kInv, int XV, bool kFull, bool kPadX>
__global__ __launch_bounds__(512) void rq_fused_linear_kernel4(RQOp<K> op, GenArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
  _Float16* hbuf = reinterpret_cast<_Float16*>(smem4);                 // [2][2][R][kHB]
  float* lpart = reinterpret_cast<float*>(smem4 + kHbufBytes);         // [3][8][R]
  float* hscale = lpart + 3 * 8 * R;                                   // [3][R]
  float* tabs = hscale + 3 * R;                                        // [8 waves][knots | derivs]
  float* bias_lds = tabs + kTabBytes / 4;                              // [8 waves][4][kBiasRow]
  float* xbuf = bias_lds + 8 * 4 * kBiasRow;                           // [3][R][D + 4]
  const int D = a.D, XS = kPadX ? D + 4 : D;
  int* cs = reinterpret_cast<int*>(xbuf + 3 * R * (D + 4));            // [32]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s16 = lane & 15, g = lane >> 4;
  const int64_t stride = gridDim.x;
  const int64_t tile0 = blockIdx.x;
  if (tile0 >= a.tiles) return;
  // (lanes of dims beyond dt evaluate the first transformed column again, results dropped: an error they flag is that
  //  column's own -- column 0 may be an identity feature outside the box, which the reference never checks)
  if (tid < 32) cs[tid] = a.cols[tid < a.dt ? tid : 0];
  const int WD = kFull ? 8 : (a.dt + 3) >> 2;          // dim groups = waves with spline work
  const bool dim_ok = kFull || 4 * wave + g < a.dt;
  const bool active = kFull || wave < WD;
  const int grp = active ? wave : 0;

  // Accumulator slot 4t + r of a lane holds parameter slot_param(4t + r) of its element
  auto slot_param = [](int s) { return s < 2 * K ? ((s & 1) ? K + (s >> 1) : (s >> 1)) : s; };
  // ---- resident weights: fragment (k-step ks, tile t, piece) of the packed image, rows permuted to slot order -------
  // image lane l' of tile t' holds W[dim 4 grp + ((l' & 15) >> 2)][param 4t' + (l' & 3)][k = 32 ks + 8 (l' >> 4) + j]
  f16x8 wh[CT][2], wl[CT][2];
  {
    const f16x8* wbase = a.wfrag + (size_t)grp * 2 * CT * 2 * 64;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int p = slot_param(4 * t + (lane & 3));
      const int st = p >> 2, sl = (lane & ~3) | (p & 3);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        wh[t][ks] = wbase[((ks * CT + st) * 2 + 0) * 64 + sl];
        wl[t][ks] = wbase[((ks * CT + st) * 2 + 1) * 64 + sl];
      }
    }
  }
  const float w_unscale = a.wun[grp];
  // The width / height logits are divided by sqrt(hidden_features) (coupling.py:565-566) and only ever feed a softmax,
  // evaluated as exp2 of differences; the derivative logits only ever enter softplus(beta u): the factors are folded
  // into the fma that undoes the operand scaling, so the bias is kept pre-multiplied by them.
  const float inv_div = op.inv_div;
  const float wh_mul = (float)((double)inv_div * 1.4426950408889634);
  for (int i = tid; i < WD * 4 * PP; i += 512) {
    const int row = i / PP, slot = i - row * PP;       // row = group * 4 + g
    const int prm = slot_param(slot);
    bias_lds[row * kBiasRow + slot] = a.bias[row * PP + prm] * (slot < 2 * K ? wh_mul : op.q.beta);
  }
  const f32x4* bwp = reinterpret_cast<const f32x4*>(bias_lds + (grp * 4 + g) * kBiasRow);

  // knot constants of the generated evaluation (formed in double on the host, fc_rq_op.h rq_finish_params)
  const RQParams& q = op.q;
  const f2 sc1 = {q.sc1x, q.sc1y};
  const f2 kc0 = {q.kcx[0], q.kcy[0]}, kc1 = {q.kcx[1], q.kcy[1]}, kc2 = {q.kcx[2], q.kcy[2]}, kc3 = {q.kcx[3], q.kcy[3]},
           kc4 = {q.kcx[4], q.kcy[4]}, kc5 = {q.kcx[5], q.kcy[5]}, kc6 = {q.kcx[6], q.kcy[6]}, kc7 = {q.kcx[7], q.kcy[7]},
           kc8 = {q.kcx[8], q.kcy[8]}, kc9 = {q.kcx[9], q.kcy[9]};
  (void)kc0; (void)kc1; (void)kc2; (void)kc3; (void)kc4; (void)kc5; (void)kc6; (void)kc7; (void)kc8; (void)kc9;

  // Lane-private bin tables: knot slots 0 and K are the interval ends and never change, slots 1..K-1 are rewritten per
  // element; derivative slots 0 and K hold the linear-tail constant (rational_quadratic.py:33-36) -- without tails all
  // K + 1 derivative logits come from the conditioner.  [slot][lane] layout.
  float* ktab = tabs + wave * (kKnotFloats + kDerFloats) + lane * 2;
  float* dtab = tabs + wave * (kKnotFloats + kDerFloats) + kKnotFloats + lane;
  *reinterpret_cast<f2*>(ktab) = f2{q.left, q.bottom};
  *reinterpret_cast<f2*>(ktab + K * 128) = f2{q.right, q.top};
  if constexpr (kTails) {
    dtab[0] = q.tail_const * q.beta;
    dtab[K * 64] = q.tail_const * q.beta;
  }

  uint32_t err = 0;
  int mycol = 0;     // this lane's column of x (cs[] is read ONCE, behind the prologue's barrier: per step it was an LDS round trip in front of the x read)
  const int xvec = R * D / 4;     // float4 per x tile: thread tid owns slots tid + 512 k, k < XV
  float4 hv0, xv0, xv1;
  hv0 = xv0 = xv1 = float4{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int64_t t) __attribute__((always_inline)) {
    hv0 = reinterpret_cast<const float4*>(a.h + t * R * 64)[tid];
    const float4* xg = reinterpret_cast<const float4*>(a.x + t * R * D);
    xv0 = xg[tid < xvec ? tid : 0];
    if constexpr (XV > 1) xv1 = xg[tid + 512 < xvec ? tid + 512 : 0];
  };
  auto xslot = [&](int buf, int i) __attribute__((always_inline)) {   // float4 index i of a [R, D] tile -> its LDS position
    if constexpr (!kPadX) return reinterpret_cast<float4*>(xbuf + buf * R * XS + 4 * i);
    const int e = i * 4, r = e / D, c = e - r * D;
    return reinterpret_cast<float4*>(xbuf + (buf * R + r) * XS + c);
  };
  // thread tid holds h[row tid >> 4][4 (tid & 15) ..]: the 16 threads of a row are 16 adjacent lanes.
  // h tile -> hbuf[hb2] (ring of 2), its row scales and the x tile -> ring slot x3 (ring of 3)
  auto park = [&](int hb2, int x3) __attribute__((always_inline)) {
    const int c = (tid & 15) * 4, r = tid >> 4;
    const float v[4] = {hv0.x, hv0.y, hv0.z, hv0.w};
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
    _Float16* dst = hbuf + (hb2 * 2 * R + r) * kHB + c;
    *reinterpret_cast<u32x2*>(dst) = p0;
    *reinterpret_cast<u32x2*>(dst + kHPiece) = p1;
    if ((tid & 15) == 0) hscale[x3 * R + r] = un;
    if (tid < xvec) *xslot(x3, tid) = xv0;
    if constexpr (XV > 1) if (tid + 512 < xvec) *xslot(x3, tid + 512) = xv1;
  };
  // B operand (h^T piece `hp`, k-step ks) of block `blk` in buffer `hb`:
  // lane holds h[sample 16 blk + (lane & 15)][k = 32 ks + 8 (lane >> 4) + j]
  auto hfrag = [&](int hb, int blk, int hp, int ks) {
    return *reinterpret_cast<const f16x8*>(hbuf + ((hb * 2 + hp) * R + 16 * blk + s16) * kHB + 32 * ks + 8 * g);
  };
  // MFMA number n of a block: term n / (2 CT) (0: Wl hh, 1: Wh hl, 2: Wh hh -- small products first),
  // k-step (n / CT) % 2, tile n % CT
  auto mfma_n = [&](auto N, f32x4 (&acc)[CT], const f16x8& bcur) {
    constexpr int n = decltype(N)::value, term = n / (2 * CT), ks = (n / CT) % 2, t = n % CT;
    const f16x8& aop = term == 0 ? wl[t][ks] : wh[t][ks];
    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop, bcur, n < CT ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[t], 0, 0, 0);
  };
  constexpr auto term_h = [](int term) { return term == 1 ? 1 : 0; };   // h piece of a term

  // Prologue only: the parameters of block `blk` of the tile in buffer `hb`, nothing to evaluate yet.
  auto produce_only = [&](f32x4 (&acc)[CT], int hb, int blk) {
    f16x8 bcur = hfrag(hb, blk, term_h(0), 0);
    static_for([&](auto N) {
      constexpr int n = decltype(N)::value;
      if constexpr (n % CT == 0 && n > 0) bcur = hfrag(hb, blk, term_h(n / (2 * CT)), (n / CT) % 2);
      mfma_n(N, acc, bcur);
    }, std::make_integer_sequence<int, NM>{});
  };

  // One step: evaluate this lane's element of block `cblk` of the tile in ring slot `xb` from the accumulators, then
  // (from the point where the evaluation has read them all) produce into the SAME accumulators the parameters of block
  // `pblk` of the tile in buffer `hb`.
  auto step = [&](f32x4 (&acc)[CT], int xb, int cblk, int hb, int pblk, float& lad_out) {
    float* xr = xbuf + (xb * R + 16 * cblk + s16) * XS + mycol;
    const float x = *xr;
    const float c_d = hscale[xb * R + 16 * cblk + s16] * w_unscale;   // undoes both scalings (a power of two)
    const float c_wh = c_d * wh_mul;        // (c_d is a power of two: the product is exact)
    const float c_ud = c_d * q.beta;
    f32x4 bw[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) bw[t] = bwp[t];
    // h^T fragments are read one group of CT MFMAs ahead of their use
    f16x8 bcur, bnext = hfrag(hb, pblk, term_h(0), 0);
    constexpr int PQ = (NM + 3) / 4;       // hooks per priority level
    auto hook = [&](auto N) {
      constexpr int n = decltype(N)::value;
      if constexpr (n % CT == 0) {
        bcur = bnext;
        if constexpr (n + CT < NM) bnext = hfrag(hb, pblk, term_h((n + CT) / (2 * CT)), ((n + CT) / CT) % 2);
      }
      mfma_n(N, acc, bcur);
      // priority that falls as the wave advances through the step: the wave of a SIMD that is behind gets the issue
      // slots (see fc_rq_fused3.hip)
      if constexpr (n % PQ == PQ - 1 || n == NM - 1) __builtin_amdgcn_s_setprio(3 - ((n + 1) % NM) / PQ);
      __builtin_amdgcn_sched_barrier(0);
    };
    const float inv_beta = op.inv_beta;   // softplus(x, beta) = log1p(exp(beta x)) * (1 / beta): exact at beta = 1
    float y, lad;
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_sched_barrier(0);
#define FC_HOOK(n) hook(std::integral_constant<int, n>{});
#define FC_WH_SLOT(i) ((i) < K ? 2 * (i) : 2 * ((i) - K) + 1)
#define FC_WH(i) __builtin_fmaf(acc[FC_WH_SLOT(i) >> 2][FC_WH_SLOT(i) & 3], c_wh, bw[FC_WH_SLOT(i) >> 2][FC_WH_SLOT(i) & 3])
#define FC_UD(j) __builtin_fmaf(acc[((j) + 2 * K) >> 2][((j) + 2 * K) & 3], c_ud, bw[((j) + 2 * K) >> 2][((j) + 2 * K) & 3])
#define FC_KNOT_ST(slot, v) *reinterpret_cast<f2*>(ktab + (slot) * 128) = (v)
#define FC_KNOT_LD(i, off) *reinterpret_cast<const f2*>(ktab + ((i) + (off)) * 128)
#define FC_DER_ST(slot, v) dtab[(slot) * 64] = (v)
#define FC_DER_LD(i, off) dtab[((i) + (off)) * 64]
#define FC_COUNT_GE(count, a, b)                                                              \
  do {                                                                                        \
    const float fc_b_ = (b);                                                                  \
    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(count) : "v"(a), "v"(fc_b_) : "vcc"); \
  } while (0)
#include FC_F4_EVAL_INC
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
  // logabsdet partials of this wave's 4 dims, both blocks of the tile at once: lanes s, s+16, s+32, s+48 hold the same
  // sample; one merge step puts block 0's pair sums into lanes 0-31 and block 1's into lanes 32-63 (fc_lane.h)
  auto after_steps = [&](int xb, float l0, float l1) __attribute__((always_inline)) {
    const float m = lane_merge32(l0, l1);
    const float l = m + lane_xor16(m, lane);
    if ((g & 1) == 0) lpart[(xb * 8 + wave) * R + 16 * (g >> 1) + s16] = l;
  };

  f32x4 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Rings as in kernel 3: h tile double-buffered; x tile, row scales and logabsdet partials in a ring of three, because
  // the results of a tile leave only after the NEXT tile's barrier (one barrier per tile):
  //   iteration i:  step 0 of tile i | park tile i+1 | BARRIER | write out tile i-1 | step 1 of tile i
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
#ifdef FC_F4_STAMP   // probe builds (tools/probe/fused4_clock.py): cycles each wave spends in the phases of the loop
  const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
  uint64_t phase_cyc[6] = {0, 0, 0, 0, 0, 0}, phase_t = stamp_c0;
#define FC_PHASE(k)                                              \
  do {                                                           \
    const uint64_t now = __builtin_amdgcn_s_memtime();           \
    phase_cyc[k] += now - phase_t;                               \
    phase_t = now;                                               \
  } while (0)
#else
#define FC_PHASE(k)
#endif
  fetch(tile0);
  park(0, 0);
  __syncthreads();
  mycol = cs[(4 * wave + g) & 31];
  if (active) produce_only(acc, 0, 0);   // block 0 of the first tile
  int hb = 0, x3 = 0;          // ring slots of the current tile
  int64_t prev_tile = -1;
  for (int64_t tile = tile0; tile < a.tiles; tile += stride) {
    const bool has_next = tile + stride < a.tiles;
    const int x3n = x3 == 2 ? 0 : x3 + 1, x3p = x3 == 0 ? 2 : x3 - 1;
    FC_PHASE(0);
    if (has_next) fetch(tile + stride);
    float lb0 = 0.f, lb1 = 0.f;
    if (active) step(acc, x3, 0, hb, 1, lb0);
    FC_PHASE(1);
    if (has_next) park(hb ^ 1, x3n);
    FC_PHASE(2);
    __syncthreads();
    FC_PHASE(3);
    if (prev_tile >= 0) write_out(prev_tile, x3p);   // complete since every wave passed this barrier
    FC_PHASE(4);
    // Last step: evaluate block 1, produce block 0 of the next tile (unconditional: on the last tile the MFMAs work on
    // stale h rows into accumulators nobody reads -- a branch would split the interleaved block).
    if (active) {
      step(acc, x3, 1, hb ^ 1, 0, lb1);
      after_steps(x3, lb0, lb1);
    }
    FC_PHASE(5);
    prev_tile = tile;
    hb ^= 1;
    x3 = x3n;
  }
  __syncthreads();
  if (prev_tile >= 0) write_out(prev_tile, x3 == 0 ? 2 : x3 - 1);
#ifdef FC_F4_STAMP   // the stamps overwrite outputs of the workgroup's first tile
  if (tid == 0) {
    a.y[tile0 * R * D] = (float)(__builtin_amdgcn_s_memtime() - stamp_c0);
    a.y[tile0 * R * D + 1] = (float)(__builtin_amdgcn_s_memrealtime() - stamp_r0);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a.y[tile0 * R * D + 4 + wave * 6 + k] = (float)phase_cyc[k];
  }
#endif
#undef FC_PHASE
  if (err && a.err) atomicOr(a.err, err);
}

template <bool kInv, int XV, bool kFull, bool kPadX>
static hipError_t launch_cfg(const RQOp<K>& op, const GenArgs& a, hipStream_t stream) {
  const size_t lds = lds_bytes(a.D);
  if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(
      attr, reinterpret_cast<const void*>(&rq_fused_linear_kernel4<kInv, XV, kFull, kPadX>), 160 * 1024);
  if (ea != hipSuccess) return ea;
  const int64_t cus = device_cu_count();
  const unsigned grid = (unsigned)(cus < a.tiles ? cus : a.tiles);
  hipLaunchKernelGGL((rq_fused_linear_kernel4<kInv, XV, kFull, kPadX>), dim3(grid), dim3(512), lds, stream, op, a);
  return hipGetLastError();
}

template <bool kInv, int XV>
static hipError_t launch_one(const RQOp<K>& op, const GenArgs& a, hipStream_t stream) {
  if (a.D & 3) return launch_cfg<kInv, XV, false, false>(op, a, stream);   // unpadded rows: generic variant
  return a.dt == 32 ? launch_cfg<kInv, XV, true, true>(op, a, stream) : launch_cfg<kInv, XV, false, true>(op, a, stream);
}

}  // namespace f4<name>

// a.tiles counts 32-row tiles; a.H == 64, q.K == FC_F4_K, q.tails == FC_F4_TAILS
hipError_t FC_F4_CAT(launch_fused4_, FC_F4_NAME)(const RQParams& q, const GenArgs& a, hipStream_t stream) {
  using namespace FC_F4_CAT(f4, FC_F4_NAME);
  RQOp<K> op;
  op.q = q;
  op.inv_div = 1.f / q.wh_div;
  op.inv_beta = 1.f / q.beta;
  const bool inv = q.inverse != 0;
  if (R * a.D / 4 <= 512) return inv ? launch_one<true, 1>(op, a, stream) : launch_one<false, 1>(op, a, stream);
  return inv ? launch_one<true, 2>(op, a, stream) : launch_one<false, 2>(op, a, stream);
}
size_t FC_F4_CAT(fused4_lds_bytes_, FC_F4_NAME)(int d) { return FC_F4_CAT(f4, FC_F4_NAME)::lds_bytes(d); }

}  // namespace fc
