// Fused final-Linear + RQ-spline kernel, third structure: split-bf16 matrix cores, parameters born in the
// evaluating lane's registers.  gfx950.
//
//   params[n, :] = W h[n, :] + b        (flowcon/nn/nets/resnet.py:91,99, the conditioner's final Linear)
//   y, logabsdet = rq_spline(x, params) (flowcon/transforms/coupling.py:279-293,549-582)
//
// Why not the f32-input MFMA (fc_rq_fused.hip, fc_rq_fused2.hip): v_mfma_f32_*_f32 runs at the f32 VALU
// rate and, measured here, does not overlap with VALU work at all -- MFMA-only 0.85 ms, spline-only
// 0.72 ms, both 1.29 ms per 2^20-row launch, whichever way the two streams were interleaved.  The bf16
// matrix pipe is 16x faster and does run beside the VALU.  So the f32 product is computed exactly enough
// on it: every f32 value is split into three bf16 pieces (x = xh + xm + xl, exact), and
//   W h = Wl hh + Wh hl + Wm hm + Wm hh + Wh hm + Wh hh      (+ three terms <= 2^-24 |W||h|, dropped)
// is accumulated in f32 by v_mfma_f32_16x16x32_bf16: 6 MFMA terms x 2 k-steps per 16x16 output tile, 3/8
// of the f32-MFMA cycles, truncation error 1.4e-8 sum|W||h| (an f32 GEMM's own rounding is ~4e-7).
//
// Layout.  The product is taken transposed, P^T = W h^T: A = weight rows (features), B = h^T (samples on
// the columns).  The C layout then gives lane (s = lane & 15, g = lane >> 4) the features 4g..4g+3 of
// sample s; with the weight rows of a tile ordered as (dim g, param 4t + r), six tiles hand that lane all
// 24 (23 + pad) parameters of element (sample s, dim 4w + g) in its own accumulators: no LDS round trip
// for the parameters (the f32 kernels moved 98 KB of them through LDS per 32-row tile).
//
// One 512-thread workgroup per CU walks 32-row tiles.  Wave w owns transformed dims 4w..4w+3 for the whole
// kernel: their weight pieces Wh, Wm stay in 96 VGPRs, Wl in LDS (96 KB for the 8 waves).  Per 16-sample
// block a wave issues 72 MFMAs for the NEXT block and evaluates one element per lane of the CURRENT
// block, hand-interleaved: the evaluation is generated straight-line code with 72 hook points
// (tools/gen_fused_eval.py), hook n issues MFMA n and pins it with a sched_barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "fc_tile.h"
#include "fc_math.h"
#include "fc_rq_op.h"
#include "fc_rq_fused.h"
#include "../../include/flowcon_hip.h"

// tools/probe/build_fused_variants.sh only: ablation builds (1 no evaluation, 4 no MFMAs, 8 loads from L2,
// 16 clock stamps, 32 half of the MFMAs).
#ifndef FC_ABL
#define FC_ABL 0
#endif

namespace fc {

// (x, y) pair of the dual-axis walk.  FC_SCALAR_WALK: plain struct instead of the packed-math vector type.
struct s2 {
  float x, y;
};
__device__ __forceinline__ s2 operator+(s2 a, s2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ s2 operator-(s2 a, s2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ s2 operator*(s2 a, s2 b) { return {a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ s2 operator*(s2 a, float b) { return {a.x * b, a.y * b}; }
__device__ __forceinline__ s2& operator+=(s2& a, s2 b) { a.x += b.x; a.y += b.y; return a; }
#ifndef FC_CUM_T
#define FC_CUM_T double   // at::cumsum on the CPU accumulates f32 in double
#endif
#ifdef FC_SCALAR_WALK
#define FC_F2 s2
#else
#define FC_F2 f2
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kCt3 = 6;                       // 16-feature tiles per wave: 4 dims x 24 padded params
constexpr int kHB = kH + 8;                   // bf16 per h row in LDS (144 B: conflict-free b128 reads)
constexpr int kHPiece = kR * kHB;             // bf16 per piece per buffer
constexpr int kWlBytes = 8 * kCt3 * 2 * 64 * 16;   // Wl fragments [wave][tile][k-step][lane] x 16 B
constexpr int kHbufBytes = 2 * 3 * kHPiece * 2;
constexpr int kBiasBytes = 8 * kCt3 * 4 * 16;      // [wave][tile][g] float4
constexpr int kLpartBytes = 2 * 8 * kR * 4;        // [buf][wave][row]

size_t fused3_lds_bytes(int d) {
  return (size_t)kWlBytes + kHbufBytes + kBiasBytes + kLpartBytes + 2 * kR * (d + 4) * 4 + kDt * 4;
}

// x = h + m + l with bf16 pieces (round-to-nearest-even; both differences are exact in f32)
__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// term order: small products first.  (weight piece, h piece), 0 = high, 1 = middle, 2 = low
__host__ __device__ constexpr int term_w(int term) { return term == 0 ? 2 : (term == 2 || term == 3 ? 1 : 0); }
__host__ __device__ constexpr int term_h(int term) { return term == 1 ? 2 : (term == 2 || term == 4 ? 1 : 0); }

template <bool kInv>
__global__ __launch_bounds__(512) void rq_fused_linear_kernel3(RQOp<kK> op, FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  bf16x8* wl = reinterpret_cast<bf16x8*>(smem3);                                   // [8][6][2][64]
  __bf16* hbuf = reinterpret_cast<__bf16*>(smem3 + kWlBytes);                       // [2][3][kR][kHB]
  f32x4* bbuf = reinterpret_cast<f32x4*>(smem3 + kWlBytes + kHbufBytes);            // [8][6][4]
  float* lpart = reinterpret_cast<float*>(smem3 + kWlBytes + kHbufBytes + kBiasBytes);   // [2][8][kR]
  float* xbuf = lpart + 2 * 8 * kR;                                                 // [2][kR][D + 4]
  const int D = a.D, XS = D + 4;
  int* cs = reinterpret_cast<int*>(xbuf + 2 * kR * XS);                             // [kDt]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s16 = lane & 15, g = lane >> 4;
  const int64_t stride = gridDim.x;
  const int64_t tile0 = blockIdx.x;
  if (tile0 >= a.tiles) return;
  if (tid < kDt) cs[tid] = a.cols[tid];

  // ---- resident weights ---------------------------------------------------------------------------------
  // A operand of tile t, k-step ks: lane holds W[row(t, lane & 15)][k = 32 ks + 8 (lane >> 4) + j], j < 8,
  // where row(t, rho) = padded feature (dim 4w + (rho >> 2)) * 24 + 4t + (rho & 3).
  bf16x8 wh[kCt3][2], wm[kCt3][2];
#pragma unroll
  for (int t = 0; t < kCt3; ++t) {
    const int row = (4 * wave + (s16 >> 2)) * kPP + 4 * t + (s16 & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float4* src = reinterpret_cast<const float4*>(a.wpad + (int64_t)row * kH + 32 * ks + 8 * g);
      const float4 v0 = src[0], v1 = src[1];
      const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      bf16x8 lo8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 ph, pm, pl;
        split3(v[j], ph, pm, pl);
        wh[t][ks][j] = ph;
        wm[t][ks][j] = pm;
        lo8[j] = pl;
      }
      wl[((wave * kCt3 + t) * 2 + ks) * 64 + lane] = lo8;
    }
    // accumulator start values: lane (s, g) register r of tile t is feature (dim 4w + g, param 4t + r)
    if (s16 == 0) {
      const float* bsrc = a.bias + (4 * wave + g) * kPP + 4 * t;
      bbuf[(wave * kCt3 + t) * 4 + g] = f32x4{bsrc[0], bsrc[1], bsrc[2], bsrc[3]};
    }
  }
  const bf16x8* wl_w = wl + wave * kCt3 * 2 * 64 + lane;   // + (t * 2 + ks) * 64
  const f32x4* bb_w = bbuf + wave * kCt3 * 4 + g;           // + t * 4

  uint32_t err = 0;
  const int xvec = kR * D / 4;
  float4 hv, xv0, xv1;
  auto fetch = [&](int64_t t) {
    if (FC_ABL & 8) t = tile0;   // ablation: every tile's loads hit in L2
    hv = reinterpret_cast<const float4*>(a.h + t * kR * kH)[tid];
    const float4* xg = reinterpret_cast<const float4*>(a.x + t * kR * D);
    xv0 = xg[tid < xvec ? tid : 0];
    xv1 = xg[tid + 512 < xvec ? tid + 512 : 0];
  };
  auto xslot = [&](int buf, int i) {   // float4 index i of a [kR, D] tile -> its padded LDS position
    const int e = i * 4, r = e / D, c = e - r * D;
    return reinterpret_cast<float4*>(xbuf + (buf * kR + r) * XS + c);
  };
  auto park = [&](int buf) {
    const int r = tid >> 4, c = (tid & 15) * 4;
    const float v[4] = {hv.x, hv.y, hv.z, hv.w};
    bf16x4 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __bf16 ph, pm, pl;
      split3(v[j], ph, pm, pl);
      p0[j] = ph; p1[j] = pm; p2[j] = pl;
    }
    __bf16* dst = hbuf + (buf * 3 * kR + r) * kHB + c;
    *reinterpret_cast<bf16x4*>(dst) = p0;
    *reinterpret_cast<bf16x4*>(dst + kHPiece) = p1;
    *reinterpret_cast<bf16x4*>(dst + 2 * kHPiece) = p2;
    if (tid < xvec) *xslot(buf, tid) = xv0;
    if (tid + 512 < xvec) *xslot(buf, tid + 512) = xv1;
  };
  // B operand (h^T piece `hp`, k-step ks) of block `blk` in buffer `hb`:
  // lane holds h[sample 16 blk + (lane & 15)][k = 32 ks + 8 (lane >> 4) + j]
  auto hfrag = [&](int hb, int blk, int hp, int ks) {
    return *reinterpret_cast<const bf16x8*>(hbuf + ((hb * 3 + hp) * kR + 16 * blk + s16) * kHB + 32 * ks + 8 * g);
  };

  // MFMA number n of a block: term n / 12, k-step (n / 6) % 2, tile n % 6
  auto mfma_n = [&](auto N, f32x4 (&acc)[kCt3], const bf16x8& bcur, const bf16x8& wlcur, const f32x4& bias) {
    constexpr int n = decltype(N)::value, term = n / 12, ks = (n / 6) % 2, t = n % 6;
    constexpr int wp = term_w(term);
    const bf16x8& aop = wp == 0 ? wh[t][ks] : (wp == 1 ? wm[t][ks] : wlcur);
    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aop, bcur, n < kCt3 ? bias : acc[t], 0, 0, 0);
  };

  // Prologue only: the parameters of block `blk` of the tile in buffer `hb`, nothing to evaluate yet.
  auto produce_only = [&](f32x4 (&acc)[kCt3], int hb, int blk) {
    bf16x8 bcur = hfrag(hb, blk, term_h(0), 0);
    auto step = [&](auto N) {
      constexpr int n = decltype(N)::value;
      if constexpr (n % 6 == 0 && n > 0) bcur = hfrag(hb, blk, term_h(n / 12), (n / 6) % 2);
      // n < 12: term 0, k-step n / 6, tile n % 6 -> Wl fragment index (n % 6) * 2 + n / 6
      const bf16x8 wlcur = n < 12 ? wl_w[((n % 6) * 2 + n / 6) * 64] : bcur;
      mfma_n(N, acc, bcur, wlcur, n < kCt3 ? bb_w[(n % 6) * 4] : f32x4{0.f, 0.f, 0.f, 0.f});
    };
#define FC_S(n) step(std::integral_constant<int, n>{});
    FC_S(0) FC_S(1) FC_S(2) FC_S(3) FC_S(4) FC_S(5) FC_S(6) FC_S(7) FC_S(8) FC_S(9) FC_S(10) FC_S(11)
    FC_S(12) FC_S(13) FC_S(14) FC_S(15) FC_S(16) FC_S(17) FC_S(18) FC_S(19) FC_S(20) FC_S(21) FC_S(22) FC_S(23)
    FC_S(24) FC_S(25) FC_S(26) FC_S(27) FC_S(28) FC_S(29) FC_S(30) FC_S(31) FC_S(32) FC_S(33) FC_S(34) FC_S(35)
    FC_S(36) FC_S(37) FC_S(38) FC_S(39) FC_S(40) FC_S(41) FC_S(42) FC_S(43) FC_S(44) FC_S(45) FC_S(46) FC_S(47)
    FC_S(48) FC_S(49) FC_S(50) FC_S(51) FC_S(52) FC_S(53) FC_S(54) FC_S(55) FC_S(56) FC_S(57) FC_S(58) FC_S(59)
    FC_S(60) FC_S(61) FC_S(62) FC_S(63) FC_S(64) FC_S(65) FC_S(66) FC_S(67) FC_S(68) FC_S(69) FC_S(70) FC_S(71)
#undef FC_S
  };

  // One step: evaluate this lane's element of block `cblk` of the tile in buffer `xb` from the parameters in
  // `pa`, and produce into `acc` the parameters of block `pblk` of the tile in buffer `hb`.
  auto step = [&](const f32x4 (&pa)[kCt3], int xb, int cblk, f32x4 (&acc)[kCt3], int hb, int pblk) {
    float* xr = xbuf + (xb * kR + 16 * cblk + s16) * XS + cs[4 * wave + g];
    const float x = *xr;
    // LDS operands are read well ahead of the MFMA that takes them: h^T fragments one group of 6 MFMAs,
    // Wl fragments 4 MFMAs, accumulator start values 3 MFMAs
    bf16x8 bcur, bnext = hfrag(hb, pblk, term_h(0), 0);
    bf16x8 wlq[4] = {wl_w[0], wl_w[2 * 64], wl_w[4 * 64], wl_w[6 * 64]};
    f32x4 bq[3] = {bb_w[0], bb_w[4], bb_w[8]};
    auto hook = [&](auto N) {
      constexpr int n = decltype(N)::value;
      if constexpr (n % 6 == 0) {
        bcur = bnext;
        if constexpr (n + 6 < 72) bnext = hfrag(hb, pblk, term_h((n + 6) / 12), ((n + 6) / 6) % 2);
      }
      const bf16x8 wlcur = wlq[n < 12 ? n % 4 : 0];
      const f32x4 bcurv = bq[n < kCt3 ? n % 3 : 0];
      if constexpr (n + 4 < 12) wlq[n % 4] = wl_w[(((n + 4) % 6) * 2 + (n + 4) / 6) * 64];
      if constexpr (n + 3 < kCt3) bq[n % 3] = bb_w[(n + 3) * 4];
      if constexpr (!(FC_ABL & 4) && !((FC_ABL & 32) && n >= 36)) mfma_n(N, acc, bcur, wlcur, bcurv);
      __builtin_amdgcn_sched_barrier(0);
    };
    const RQParams& q = op.q;
    const float inv_div = op.inv_div;
    float y, lad;
    __builtin_amdgcn_sched_barrier(0);
#define FC_HOOK(n) hook(std::integral_constant<int, n>{});
#define FC_P(i) pa[(i) >> 2][(i) & 3]
#if FC_ABL & 1
    FC_HOOK(0) FC_HOOK(1) FC_HOOK(2) FC_HOOK(3) FC_HOOK(4) FC_HOOK(5) FC_HOOK(6) FC_HOOK(7) FC_HOOK(8) FC_HOOK(9)
    FC_HOOK(10) FC_HOOK(11) FC_HOOK(12) FC_HOOK(13) FC_HOOK(14) FC_HOOK(15) FC_HOOK(16) FC_HOOK(17) FC_HOOK(18) FC_HOOK(19)
    FC_HOOK(20) FC_HOOK(21) FC_HOOK(22) FC_HOOK(23) FC_HOOK(24) FC_HOOK(25) FC_HOOK(26) FC_HOOK(27) FC_HOOK(28) FC_HOOK(29)
    FC_HOOK(30) FC_HOOK(31) FC_HOOK(32) FC_HOOK(33) FC_HOOK(34) FC_HOOK(35) FC_HOOK(36) FC_HOOK(37) FC_HOOK(38) FC_HOOK(39)
    FC_HOOK(40) FC_HOOK(41) FC_HOOK(42) FC_HOOK(43) FC_HOOK(44) FC_HOOK(45) FC_HOOK(46) FC_HOOK(47) FC_HOOK(48) FC_HOOK(49)
    FC_HOOK(50) FC_HOOK(51) FC_HOOK(52) FC_HOOK(53) FC_HOOK(54) FC_HOOK(55) FC_HOOK(56) FC_HOOK(57) FC_HOOK(58) FC_HOOK(59)
    FC_HOOK(60) FC_HOOK(61) FC_HOOK(62) FC_HOOK(63) FC_HOOK(64) FC_HOOK(65) FC_HOOK(66) FC_HOOK(67) FC_HOOK(68) FC_HOOK(69)
    FC_HOOK(70) FC_HOOK(71)
    y = x + (FC_P(0) + FC_P(5) + FC_P(10) + FC_P(15) + FC_P(16) + FC_P(21)) * 0.f + q.left * 0.f + inv_div * 0.f;
    lad = 0.f;
#else
#include "fc_rq_fused3_eval.inc"
#endif
#undef FC_P
#undef FC_HOOK
    *xr = y;
    // logabsdet partial of this wave's 4 dims: lanes s, s+16, s+32, s+48 hold the same sample
    float l = lad;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (g == 0) lpart[(xb * 8 + wave) * kR + 16 * cblk + s16] = l;
  };

#if FC_ABL & 16   // ablation: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped around the loop
  const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  f32x4 acc0[kCt3], acc1[kCt3];
#pragma unroll
  for (int t = 0; t < kCt3; ++t) acc0[t] = acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(tile0);
  park(0);
  __syncthreads();
  produce_only(acc0, 0, 0);   // block 0 of the first tile
  int tb = 0;
  for (int64_t tile = tile0; tile < a.tiles; tile += stride) {
    const bool has_next = tile + stride < a.tiles;
    if (has_next) fetch(tile + stride);
    step(acc0, tb, 0, acc1, tb, 1);        // A: evaluate block 0 of `tile`, produce its block 1
    if (has_next) park(tb ^ 1);
    __syncthreads();
    // B: evaluate block 1, produce block 0 of the next tile (unconditional: on the last tile the MFMAs work
    // on stale h rows into accumulators nobody reads -- a branch would split the interleaved block).
    step(acc1, tb, 1, acc0, tb ^ 1, 0);
    __syncthreads();
    // Every thread writes out exactly the float4 slots it parks, and lpart is double-buffered, so no third
    // barrier is needed before the next iteration.
    {
      float4* yg = reinterpret_cast<float4*>(a.y + tile * kR * D);
      if (tid < xvec) yg[tid] = *xslot(tb, tid);
      if (tid + 512 < xvec) yg[tid + 512] = *xslot(tb, tid + 512);
      if (tid < kR) {
        const float* lp = lpart + tb * 8 * kR + tid;
        float l = lp[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) l += lp[w * kR];
        a.logabsdet[tile * kR + tid] = l;
      }
    }
    tb ^= 1;
  }
#if FC_ABL & 16   // the stamps overwrite two outputs of the workgroup's first tile: probe builds only
  if (tid == 0) {
    a.y[tile0 * kR * D] = (float)(__builtin_amdgcn_s_memtime() - stamp_c0);
    a.y[tile0 * kR * D + 1] = (float)(__builtin_amdgcn_s_memrealtime() - stamp_r0);
  }
#endif
  if (err && a.err) atomicOr(a.err, err);
}

hipError_t launch_fused3(const RQOp<kK>& op, const FusedArgs& a, unsigned grid, hipStream_t stream) {
  const size_t lds = fused3_lds_bytes(a.D);
  if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rq_fused_linear_kernel3<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rq_fused_linear_kernel3<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (op.q.inverse)
    hipLaunchKernelGGL(rq_fused_linear_kernel3<true>, dim3(grid), dim3(512), lds, stream, op, a);
  else
    hipLaunchKernelGGL(rq_fused_linear_kernel3<false>, dim3(grid), dim3(512), lds, stream, op, a);
  return hipGetLastError();
}

}  // namespace fc
