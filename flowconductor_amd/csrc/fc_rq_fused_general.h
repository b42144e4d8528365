// Fused final-Linear + RQ-spline kernel for GENERAL layer shapes (the reference's defaults and beyond):
//   K = 4 .. 16 bins, linear tails or none, hidden width 64 / 128 / 256 (narrower widths run
//   zero-padded), up to 32 transformed dims per launch, D <= 128.
//
//   params[n, :] = W h[n, :] + b        (flowcon/nn/nets/resnet.py:91,99: the conditioner's final Linear;
//                                        num_bins defaults to 10, coupling.py:507; hidden_features is free, resnet.py:62)
//   y, logabsdet = rq_spline(x, params) (coupling.py:279-293, 549-582; rational_quadratic.py:13-181)
//
// fc_rq_fused3.hip is the hand-scheduled special case (K = 8, hidden 64, linear tails: both weight pieces resident
// in 96 registers of every wave).  At K = 10 the resident weights alone would need 128 registers per wave next to
// two accumulator sets, at hidden 256 four times that: here the weights are NOT resident.  The host packs them once
// per parameter version into matrix-core fragment order -- scaled by a power of two per group of 4 dims and split
// into two f16 pieces (fc_split.h) -- and every wave streams the fragments of ITS dims from L2 (the whole image is
// <= 1 MB, shared by the 32 CUs of an XCD) straight into A operands.  The rest is the structure of kernel 3:
//   * product transposed (A = weight rows, B = h^T), so the C layout hands lane (sample s, dim 4w+g) all its 3K-/+1
//     parameters in its own accumulators -- the [N, d_t (3K-/+1)] tensor never exists in memory;
//   * one 512-thread workgroup per CU walks 32-row tiles; wave w owns dims 4w..4w+3; h tiles (scaled + split once per
//     row) and x tiles double-buffered in LDS, next tile's global loads in flight during the current tile's work,
//     ONE barrier per tile;
//   * per tile a wave accumulates both 16-sample blocks against each weight fragment it fetches (3 split terms x 2
//     blocks = 6 MFMAs per 2 KB of weights), then evaluates its two elements per lane with the stand-alone kernel's
//     RQOp<K>::eval_core on register parameters (the same arithmetic, operation for operation, as fc_rq_spline).
// The two waves of a SIMD are not phase-locked (nothing but the tile barrier couples them), so one wave's weight
// fetch / MFMA phase overlaps the other's spline arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_device.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "fc_rq_op.h"
#include "fc_split.h"
#include "fc_tile.h"
#include "../../include/flowcon_hip.h"

namespace fc {

constexpr int kGenRows = 32;      // rows per tile = two 16-sample blocks
constexpr int kGenThreads = 512;

struct GenArgs {
  const float* x;          // [N, D]
  float* y;                // [N, D]
  const float* h;          // [N, H]  last hidden activation (H = 64, 128 or 256; zero-padded by the producer)
  const f16x8* wfrag;      // [groups][H/32][T][2 pieces][64 lanes]  packed weight fragments (ops.pack_final_layer_general)
  const float* wun;        // [groups]  2^-S of the group's weight scale
  const float* bias;       // [groups][4][PP]
  const int32_t* cols;     // [dt]
  float* logabsdet;        // [N]
  uint32_t* err;
  int64_t tiles;           // 32-row tiles
  int D, H, dt, accumulate;
};

template <int K, bool kTails>
struct GenShape {
  static constexpr int P = kTails ? 3 * K - 1 : 3 * K + 1;
  static constexpr int PP = (P + 3) / 4 * 4;
  static constexpr int T = PP / 4;               // 16-row MFMA tiles per wave (4 dims x PP parameters)
  static constexpr int TC = T > 8 ? (T + 1) / 2 : T;   // fragments fetched per batch (register budget)
};

inline size_t gen_lds_bytes(int d, int h, int rows = kGenRows) {
  const size_t hb = (size_t)2 * 2 * rows * (h + 16) * 2;      // [buf][piece][row][H + 16] f16
  const size_t xb = (size_t)2 * rows * (d + 4) * 4;           // [buf][row][D + 4]
  return hb + xb + 2 * rows * 4 /* row scales */ + 2 * 8 * rows * 4 /* logabsdet partials */ + 32 * 4 +
         (size_t)32 * 52 * 4 /* bias image, PP <= 52 */;
}

// max over the aligned group of G lanes (16, 32 or 64) a row of h is spread over
template <int G>
__device__ __forceinline__ float group_allmax(float m, int lane) {
  m = row16_allmax(m);
  if constexpr (G >= 32) m = fmaxf(m, lane_xor16(m, lane));
  if constexpr (G >= 64) m = fmaxf(m, lane_xor32(m, lane));
  return m;
}

// HQ = H / 64 (1, 2, 4): float4 pieces of the h tile per thread
// (K <= 8 at hidden <= 128: 128 registers, so that TWO workgroups share a CU -- the kernel spends half of its wave cycles
//  waiting for weight fragments from L2: 0.79 -> 0.73 ms per 2^20 rows at K = 8 / hidden 64; K = 10 needs 171 registers)
constexpr int gen_waves_per_simd(int k, int hq) { return (k <= 8 && hq <= 2) ? 4 : 2; }

// 16-sample blocks per tile.  A wave streams its weight fragments from L2 once per TILE, so at hidden 128 / 256 (64 / 128 KB
// per wave and tile; K = 10 / hidden 256: 59 % of the wave cycles waiting on memory, profiles/r03_general_h256_sq_counters.txt)
// three blocks per tile cut the stream by a third -- where the accumulators (T <= 8 tiles x 3 blocks) and the LDS image
// (48-row h and x tiles, D <= 85) fit.  The last 48-row tile may be partial (the entry takes multiples of 32 rows).
// (the 128-register kernels that share a CU in pairs -- K <= 8 at hidden 128 -- have no room for a third accumulator set)
constexpr int gen_blocks_static(int k, int t, int hq) { return (hq >= 2 && t <= 8 && gen_waves_per_simd(k, hq) == 2) ? 3 : 2; }
inline int gen_blocks(int k, int t, int hq, int d) {
#ifdef FC_GEN_TWO_BLOCKS   // probe builds: A/B against the two-block tiles
  return 2;
#endif
  return (gen_blocks_static(k, t, hq) == 3 && 48 * d / 4 <= 2 * kGenThreads && gen_lds_bytes(d, 64 * hq, 48) <= 160 * 1024) ? 3 : 2;
}

// Round 3: PARTLY RESIDENT weights at hidden 64 for the shapes that live at two waves per SIMD anyway (K >= 9: the
// reference's default K = 10).  Those kernels used 157 - 219 of their 256 registers and spent most of their time waiting
// for the same 16 - 24 KB of weight fragments from L2 on every 32-row tile (K = 10: 0.97 ms per 2^20 rows against 0.43 ms
// of the fully resident K = 8 kernel).  The first `gen_resident_pairs` (hi, lo) fragment pairs of a wave now stay in
// registers for the whole kernel; only the rest is streamed, all of it requested at the top of the product phase.
// T tiles x 2 k-steps pairs in all; 8 registers per pair; chosen so that no instantiation spills (tools/kernel_stats.py).
constexpr int gen_resident_pairs(int t, int hq, int k) {
  if (hq != 1 || k <= 8) return 0;
  // what fits without spills next to two accumulator sets, the bias rows and the evaluation's temporaries (tools/kernel_stats.py):
  // K = 9: 10 of 14 pairs, K = 10: 7 of 16, K = 11: 3 of 16 / 18; from K = 12 on nothing is left
  (void)t;
  return k == 9 ? 10 : k == 10 ? 7 : k == 11 ? 3 : 0;
}

template <int K, bool kTails, int HQ, int NBK = 2>
__global__ __launch_bounds__(kGenThreads, gen_waves_per_simd(K, HQ)) void rq_fused_general_kernel(RQOp<K> op, GenArgs a) {
  using S = GenShape<K, kTails>;
  constexpr int PP = S::PP, T = S::T;
  [[maybe_unused]] constexpr int TC = S::TC;
  constexpr int RES = NBK == 2 ? gen_resident_pairs(T, HQ, K) : 0;
  constexpr int R = 16 * NBK;
  constexpr bool kPartial = (R % kGenRows) != 0;         // the last tile may hold fewer rows
  constexpr int HV = R * 16 * HQ / kGenThreads;          // float4 of the h tile per thread
  static_assert(R * 16 * HQ % kGenThreads == 0, "h tile: whole float4 rounds");
  constexpr int H = 64 * HQ, KS = H / 32, HB = H + 16;   // row stride 32 mod 64 bytes: conflict-free b128 fragment reads (tools/lds_conflicts.py)
  constexpr int kRowLanes = H / 4;                       // threads that share a row of the h tile
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  const int D = a.D;
  const bool pad_x = (D & 3) == 0;
  const int XS = pad_x ? D + 4 : D;
  _Float16* hbuf = reinterpret_cast<_Float16*>(gsm);                              // [2][2][R][HB]
  float* xbuf = reinterpret_cast<float*>(gsm + (size_t)2 * 2 * R * HB * 2);       // [2][R][D + 4]
  float* hscale = xbuf + 2 * R * (D + 4);                                          // [2][R]
  float* lpart = hscale + 2 * R;                                                   // [2][8][R]
  int* cs = reinterpret_cast<int*>(lpart + 2 * 8 * R);                             // [32]
  float* bias_lds = reinterpret_cast<float*>(cs + 32);                             // [8 groups][4][PP]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s16 = lane & 15, g = lane >> 4;
  const int64_t stride = gridDim.x, tile0 = blockIdx.x;
  const int64_t n_rows = a.tiles * kGenRows, tiles = (n_rows + R - 1) / R;
  if (tile0 >= tiles) return;
  // (lanes of dims beyond dt evaluate the first transformed column again, results dropped: whatever error they flag is that
  //  column's own -- column 0 may be an identity feature outside the box, which the reference never checks, coupling.py:79-88)
  if (tid < 32) cs[tid] = a.cols[tid < a.dt ? tid : 0];
  const int WD = (a.dt + 3) >> 2;                 // waves with spline work
  const bool active = wave < WD;
  const bool dim_ok = 4 * wave + g < a.dt;
  const int grp = active ? wave : 0;

  // bias image in LDS (read back per element: resident it would cost PP registers per lane); lane (s, g) uses the
  // parameters of dim 4 grp + g, in accumulator order.  Weight unscale of the group.
  for (int i = tid; i < WD * 4 * PP; i += kGenThreads) bias_lds[i] = a.bias[i];
  const f32x4* bw = reinterpret_cast<const f32x4*>(bias_lds + (grp * 4 + g) * PP);
  const float w_un = a.wun[grp];
  const f16x8* wsrc = a.wfrag + (size_t)grp * KS * T * 2 * 64 + lane;

  f16x8 rh[RES > 0 ? RES : 1], rl[RES > 0 ? RES : 1];       // resident pairs: index i = ks * T + t, the image's own order
  if constexpr (RES > 0) {
#pragma unroll
    for (int i = 0; i < RES; ++i) {
      rh[i] = wsrc[(i * 2 + 0) * 64];
      rl[i] = wsrc[(i * 2 + 1) * 64];
    }
  }

  uint32_t err = 0;
  const int xvec = R * D / 4;                     // float4 per x tile (R * D is a multiple of 4)
  float4 hv[HV], xv0, xv1;
  xv0 = xv1 = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < HV; ++k) hv[k] = float4{0.f, 0.f, 0.f, 0.f};
  // rows of tile t that exist (a partial last tile: the missing rows are read as the tile's first float4 -- any finite
  // values do --, evaluated like the others and never written out)
  auto rows_of = [&](int64_t t) __attribute__((always_inline)) {
    if constexpr (!kPartial) return R;
    const int64_t left = n_rows - t * R;
    return left < R ? (int)left : R;
  };
  auto fetch = [&](int64_t t) __attribute__((always_inline)) {
    const int rows = rows_of(t);
    const int hvalid = rows * kRowLanes, xvalid = kPartial ? rows * D / 4 : xvec;
    const float4* hg = reinterpret_cast<const float4*>(a.h + t * R * H);
#pragma unroll
    for (int k = 0; k < HV; ++k) hv[k] = hg[(!kPartial || tid + kGenThreads * k < hvalid) ? tid + kGenThreads * k : 0];
    const float4* xg = reinterpret_cast<const float4*>(a.x + t * R * D);
    xv0 = xg[tid < xvalid ? tid : 0];
    xv1 = xg[tid + kGenThreads < xvalid ? tid + kGenThreads : 0];
  };
  auto xslot = [&](int buf, int i) __attribute__((always_inline)) {
    if (!pad_x) return reinterpret_cast<float4*>(xbuf + buf * R * (D + 4) + 4 * i);
    const int e = i * 4, r = e / D, c = e - r * D;
    return reinterpret_cast<float4*>(xbuf + buf * R * (D + 4) + r * XS + c);
  };
  // float4 index f = tid + 512 k of the [R, H] tile: row f / (H/4), columns 4 (f % (H/4)) ..; the H/4 threads of a
  // row are an aligned group of 16 / 32 / 64 lanes of one wave
  auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < HV; ++k) {
      const int f = tid + kGenThreads * k, r = f / kRowLanes, c = (f % kRowLanes) * 4;
      const float v[4] = {hv[k].x, hv[k].y, hv[k].z, hv[k].w};
      const float m = group_allmax<kRowLanes>(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), lane);
      float sc, un;
      pow2_scale(m, sc, un);
      f16x4 p0, p1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        _Float16 ph, pl;
        split2(v[j] * sc, ph, pl);
        p0[j] = ph;
        p1[j] = pl;
      }
      _Float16* dst = hbuf + ((size_t)(buf * 2) * R + r) * HB + c;
      *reinterpret_cast<f16x4*>(dst) = p0;
      *reinterpret_cast<f16x4*>(dst + (size_t)R * HB) = p1;
      if ((f % kRowLanes) == 0) hscale[buf * R + r] = un;
    }
    if (tid < xvec) *xslot(buf, tid) = xv0;
    if (tid + kGenThreads < xvec) *xslot(buf, tid + kGenThreads) = xv1;
  };
  auto hfrag = [&](int buf, int blk, int piece, int ks) __attribute__((always_inline)) {
    return *reinterpret_cast<const f16x8*>(hbuf + ((size_t)(buf * 2 + piece) * R + 16 * blk + s16) * HB + 32 * ks + 8 * g);
  };
  auto write_out = [&](int64_t t, int buf) __attribute__((always_inline)) {
    const int rows = rows_of(t);
    const int xvalid = kPartial ? rows * D / 4 : xvec;
    float4* yg = reinterpret_cast<float4*>(a.y + t * R * D);
    if (tid < xvalid) yg[tid] = *xslot(buf, tid);
    if (tid + kGenThreads < xvalid) yg[tid + kGenThreads] = *xslot(buf, tid + kGenThreads);
    if (tid < rows) {
      const float* lp = lpart + buf * 8 * R + tid;
      float l = lp[0];
#pragma unroll
      for (int w = 1; w < 8; ++w)
        if (w < WD) l += lp[w * R];
      a.logabsdet[t * R + tid] = a.accumulate ? a.logabsdet[t * R + tid] + l : l;
    }
  };

  fetch(tile0);
  park(0);
  __syncthreads();
  int buf = 0;
  for (int64_t tile = tile0; tile < tiles; tile += stride) {
    const bool has_next = tile + stride < tiles;
    if (has_next) fetch(tile + stride);
    if (active) {
      // ---- parameters of all blocks: acc[b][t] = sum over k of (scaled W)(scaled h)^T, three split terms -------------
      f32x4 acc[NBK][T];
#pragma unroll
      for (int b = 0; b < NBK; ++b)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (RES > 0) {
        // The streamed pairs come in chunks of <= CH requests, a chunk's requests all out before its first product; the
        // resident pairs are multiplied between the first chunk's requests and its products (they cover that L2 round trip).
        constexpr int NP = KS * T, NS = NP - RES, CH = NS <= 8 ? NS : (NS + 1) / 2 <= 8 ? (NS + 1) / 2 : 8;
        f16x8 bh0, bl0, bh1, bl1;
        auto products = [&](int i, const f16x8& ah, const f16x8& al) __attribute__((always_inline)) {
          const int ks = i / T, t = i - ks * T;
          if (t == 0 || i == RES) {       // (re-)read the h^T fragments of this k-step
            bh0 = hfrag(buf, 0, 0, ks); bl0 = hfrag(buf, 0, 1, ks);
            bh1 = hfrag(buf, 1, 0, ks); bl1 = hfrag(buf, 1, 1, ks);
          }
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh1, acc[1][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl1, acc[1][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, acc[1][t], 0, 0, 0);
        };
#pragma unroll
        for (int c0 = 0; c0 < NS; c0 += CH) {
          f16x8 sh[CH], sl[CH];
#pragma unroll
          for (int i = 0; i < CH; ++i)
            if (c0 + i < NS) {
              sh[i] = wsrc[((RES + c0 + i) * 2 + 0) * 64];
              sl[i] = wsrc[((RES + c0 + i) * 2 + 1) * 64];
            }
          if (c0 == 0) {
#pragma unroll
            for (int i = 0; i < RES; ++i) products(i, rh[i], rl[i]);
          }
#pragma unroll
          for (int i = 0; i < CH; ++i)
            if (c0 + i < NS) products(RES + c0 + i, sh[i], sl[i]);
        }
      } else if constexpr (gen_waves_per_simd(K, HQ) == 4 || T > 10) {
        // (the 128-register kernels: four waves per SIMD cover each other's fragment loads; T > 10: two fragment sets in
        //  flight next to 2 T accumulators spill.  A k-step's fragments are requested together, in batches of TC)
        static_assert(NBK == 2, "two blocks per tile");
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
          const f16x8 bh0 = hfrag(buf, 0, 0, ks), bl0 = hfrag(buf, 0, 1, ks);
          const f16x8 bh1 = hfrag(buf, 1, 0, ks), bl1 = hfrag(buf, 1, 1, ks);
          const f16x8* wk = wsrc + (size_t)ks * T * 2 * 64;
#pragma unroll
          for (int t0 = 0; t0 < T; t0 += TC) {
            f16x8 ah[TC], al[TC];
#pragma unroll
            for (int t = 0; t < TC; ++t)
              if (t0 + t < T) {
                ah[t] = wk[((t0 + t) * 2 + 0) * 64];
                al[t] = wk[((t0 + t) * 2 + 1) * 64];
              }
#pragma unroll
            for (int t = 0; t < TC; ++t)
              if (t0 + t < T) {
                // small products first; consecutive MFMAs alternate between the two blocks' accumulators
                acc[0][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh0, acc[0][t0 + t], 0, 0, 0);
                acc[1][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh1, acc[1][t0 + t], 0, 0, 0);
                acc[0][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl0, acc[0][t0 + t], 0, 0, 0);
                acc[1][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl1, acc[1][t0 + t], 0, 0, 0);
                acc[0][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh0, acc[0][t0 + t], 0, 0, 0);
                acc[1][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh1, acc[1][t0 + t], 0, 0, 0);
              }
          }
        }
      } else {
        // Each k-step's fragments come in two halves (tiles [0, C0) and [C0, T)); a half is requested while the other
        // half's products issue -- the second half of k-step ks during the first half's MFMAs, the first half of ks + 1
        // during the second's -- so a wave always has an L2 request in flight behind its matrix work (with the loads of
        // a whole k-step requested and waited for together the kernel spent 59 % of its wave cycles waiting on memory,
        // profiles/r03_general_h256_sq_counters.txt).
        constexpr int C0 = (T + 1) / 2, C1 = T - C0;
        f16x8 a0h[C0], a0l[C0], a1h[C1 > 0 ? C1 : 1], a1l[C1 > 0 ? C1 : 1];
        auto request0 = [&](int ks) __attribute__((always_inline)) {
          const f16x8* wk = wsrc + (size_t)ks * T * 2 * 64;
#pragma unroll
          for (int t = 0; t < C0; ++t) {
            a0h[t] = wk[(t * 2 + 0) * 64];
            a0l[t] = wk[(t * 2 + 1) * 64];
          }
        };
        auto request1 = [&](int ks) __attribute__((always_inline)) {
          const f16x8* wk = wsrc + (size_t)ks * T * 2 * 64;
#pragma unroll
          for (int t = 0; t < C1; ++t) {
            a1h[t] = wk[((C0 + t) * 2 + 0) * 64];
            a1l[t] = wk[((C0 + t) * 2 + 1) * 64];
          }
        };
        request0(0);
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
          f16x8 bh[NBK], bl[NBK];
#pragma unroll
          for (int b = 0; b < NBK; ++b) {
            bh[b] = hfrag(buf, b, 0, ks);
            bl[b] = hfrag(buf, b, 1, ks);
          }
          // small products first; consecutive MFMAs go round the blocks' accumulators
          auto products = [&](int t, const f16x8& ah, const f16x8& al) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < NBK; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[b], acc[b][t], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NBK; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[b], acc[b][t], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NBK; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[b], acc[b][t], 0, 0, 0);
          };
          // (sched_barrier: left alone the scheduler sinks every request to just before its first use -- one fragment pair,
          //  nine MFMAs, ahead -- which covers a fifth of an L2 round trip)
          request1(ks);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < C0; ++t) products(t, a0h[t], a0l[t]);
          __builtin_amdgcn_sched_barrier(0);
          request0(ks + 1 < KS ? ks + 1 : ks);      // (the last k-step asks for its own first half again: no branch in the stream)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < C1; ++t) products(C0 + t, a1h[t], a1l[t]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- the elements of this lane: (sample 16 b + s16, dim 4 wave + g) -------------------------------------------
#pragma unroll
      for (int b = 0; b < NBK; ++b) {
        const int row = 16 * b + s16;
        float* xr = xbuf + buf * R * (D + 4) + row * XS + cs[(4 * wave + g) & 31];
        const float xin = *xr;
        const float c = hscale[buf * R + row] * w_un;      // undoes both scalings (a power of two)
        float p[PP];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const f32x4 bt = bw[t];
#pragma unroll
          for (int r = 0; r < 4; ++r) p[4 * t + r] = __builtin_fmaf(acc[b][t][r], c, bt[r]);
        }
        float yv, lad;
        if constexpr (kPartial) {
          // rows beyond the batch in the last (partial) tile are phantom copies of the tile's first piece: they must not
          // raise the domain error (without tails an identity feature among those values may lie outside the box)
          uint32_t e = 0;
          op.template eval_core<true>(p, xin, yv, lad, e);
          err |= row < rows_of(tile) ? e : 0u;
        } else {
          op.template eval_core<true>(p, xin, yv, lad, err);
        }
        if (dim_ok) *xr = yv;
        const float l = rows4_allsum(dim_ok ? lad : 0.f, lane);
        if (g == 0) lpart[(buf * 8 + wave) * R + row] = l;
      }
    }
    if (has_next) park(buf ^ 1);
    __syncthreads();
    write_out(tile, buf);
    buf ^= 1;
  }
  if (err && a.err) atomicOr(a.err, err);
}

template <int K, bool kTails, int HQ, int NBK>
hipError_t launch_general_nbk(const RQOp<K>& op, const GenArgs& a, hipStream_t stream) {
  const size_t lds = gen_lds_bytes(a.D, a.H, 16 * NBK);
  if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(
      attr, reinterpret_cast<const void*>(&rq_fused_general_kernel<K, kTails, HQ, NBK>), 160 * 1024);
  if (ea != hipSuccess) return ea;
  const int64_t cus = device_cu_count(), tiles = (a.tiles * kGenRows + 16 * NBK - 1) / (16 * NBK);
  const int64_t wgs = cus * ((gen_waves_per_simd(K, HQ) == 4 && 2 * lds <= 160 * 1024) ? 2 : 1);
  const unsigned grid = (unsigned)(wgs < tiles ? wgs : tiles);
  hipLaunchKernelGGL((rq_fused_general_kernel<K, kTails, HQ, NBK>), dim3(grid), dim3(kGenThreads), lds, stream, op, a);
  return hipGetLastError();
}

template <int K, bool kTails, int HQ>
hipError_t launch_general_hq(const RQOp<K>& op, const GenArgs& a, hipStream_t stream) {
  if constexpr (gen_blocks_static(K, GenShape<K, kTails>::T, HQ) == 3) {
    if (gen_blocks(K, GenShape<K, kTails>::T, HQ, a.D) == 3) return launch_general_nbk<K, kTails, HQ, 3>(op, a, stream);
  }
  return launch_general_nbk<K, kTails, HQ, 2>(op, a, stream);
}

template <int K, bool kTails>
hipError_t launch_general(const RQParams& q, const GenArgs& a, hipStream_t stream) {
  RQOp<K> op;
  op.q = q;
  op.inv_div = 1.f / q.wh_div;
  op.inv_beta = 1.f / q.beta;
  switch (a.H) {
    case 64: return launch_general_hq<K, kTails, 1>(op, a, stream);
    case 128: return launch_general_hq<K, kTails, 2>(op, a, stream);
    case 256: return launch_general_hq<K, kTails, 4>(op, a, stream);
    default: return hipErrorInvalidValue;
  }
}

// fc_rq_fused4.hip: the resident-weight kernel for the bin counts it is instantiated for (hidden 64, linear tails)
bool fused4_takes(const RQParams& q, const GenArgs& a);
hipError_t launch_fused4(const RQParams& q, const GenArgs& a, hipStream_t stream);

// defined in fc_rq_fused_general_tails.hip / _box.hip (one translation unit per tail mode: they build in parallel)
hipError_t launch_general_tails(int K, const RQParams& q, const GenArgs& a, hipStream_t stream);
hipError_t launch_general_box(int K, const RQParams& q, const GenArgs& a, hipStream_t stream);

}  // namespace fc
