// Shared-weight Sylvester flow on the matrix cores, gfx950.
//
//   z -> y = z + Q R2 tanh(R1 Q^T z + b),   logabsdet = sum_i log(1 + (1 - tanh^2(.)_i) diag(R1)_i diag(R2)_i)
//
// (flowcon/transforms/no_analytic_inv/planar.py:144-166; Q = product of M Householder reflections,
// orthogonal.py:63-85.)  With parameters shared across the batch the two chains Q^T -> R1 and R2 -> Q are
// fixed D x D matrices: W1 = R1 Q^T, W2 = Q R2 (formed once per call by the host, in float64), and the batch
// sees two dense [N, D] x [D, D] products -- the one place of the path that is a true dense contraction
// (BASELINE.json north_star).  They run as three-term scaled two-piece f16 splits (fc_split.h) with the
// structure of fc_resnet_hidden.hip: a wave carries 16 samples through both products in registers, weight
// rows ordered so that the C layout of the first product is the B operand layout of the second
// (tile t, row 4g + r <-> feature 32 (t >> 1) + 8 g + 4 (t & 1) + r), weights of both products as ready-made
// A fragments in LDS (128 KB at D = 128).  The row-per-wave VALU kernel (fc_rowwave.hip) stays for
// per-sample parameters and other widths: 3.4 ms per 2^18 x 128 rows there, ~0.1 ms here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_split.h"
#include "fc_device.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "../../include/flowcon_hip.h"

namespace fc {

constexpr int kSylThreads = 512;

struct SylArgs {
  const float* x;      // [N, F]
  float* y;            // [N, F]
  float* lad;          // [N]
  const float* w1;     // [F, F] row-major: pre = W1 z + b
  const float* w2;     // [F, F]: y = z + W2 tanh(pre)
  const float* bias;   // [F]
  const float* rdiag;  // [F] diag(R1) * diag(R2)
  int64_t blocks16;
};

__host__ __device__ constexpr int syl_feat(int t, int g, int r) { return 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r; }

// F = 32 KS features: NT = 2 KS accumulator tiles per lane group, KS k-steps
// kDense: only the first product, y = W1 z + b (a dense linear layer with batch-independent weights: LU / Linear
// forward, a Householder sequence folded into its orthogonal matrix)
// BPW: 16-row blocks a wave carries together (each weight fragment read from LDS serves all of them)
template <int KS, bool kDense, int BPW>
__global__ __launch_bounds__(kSylThreads) void sylvester_mm_kernel(SylArgs a) {
  constexpr int F = 32 * KS, NT = 2 * KS;
  constexpr int kFragL = KS * NT * 2;   // fragments of one product: [ks][t][piece]
  extern __shared__ __attribute__((aligned(16))) unsigned char ssmem[];
  f16x8* wfrag = reinterpret_cast<f16x8*>(ssmem);                       // [2][KS][NT][2][64]
  float* bias = reinterpret_cast<float*>(ssmem + (size_t)2 * kFragL * 64 * 16);   // [g][NT * 4]
  float* rdg = bias + 4 * NT * 4;                                        // [g][NT * 4]
  float* wun = rdg + 4 * NT * 4;                                         // [2]
  float* red = wun + 8;                                                  // [8]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s16 = lane & 15, g = lane >> 4;

  // ---- once per workgroup: scale, split and lay out both matrices ------------------------------------
  // fragment entry e = (ks * NT + t) * 64 + lane': W[feat(t, lane' & 15)][32 ks + 8 (lane' >> 4) + j], j < 8 -- the entries
  // cover the matrix exactly once, so ONE pass serves both the maximum and the split: a thread keeps its kPer entries (two
  // 16-byte loads each) in registers across the reduction (round 4; before: one pass for the maximum, a second one with
  // eight scalar loads per entry -- ~10 us of a 100 us launch)
  constexpr int kPer = (KS * NT * 64 + kSylThreads - 1) / kSylThreads;
#pragma unroll
  for (int l = 0; l < (kDense ? 1 : 2); ++l) {
    const float* w = l == 0 ? a.w1 : a.w2;
    float4 keep[kPer][2];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = tid + i * kSylThreads;
      keep[i][0] = keep[i][1] = float4{0.f, 0.f, 0.f, 0.f};
      if (e < KS * NT * 64) {
        const int ln = e & 63, t = (e >> 6) % NT, ks = (e >> 6) / NT;
        const int rho = ln & 15, f = syl_feat(t, rho >> 2, rho & 3);
        const float4* src = reinterpret_cast<const float4*>(w + (size_t)f * F + 32 * ks + 8 * (ln >> 4));
        keep[i][0] = src[0];
        keep[i][1] = src[1];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
        m = fmaxf(fmaxf(m, fmaxf(fabsf(keep[i][h].x), fabsf(keep[i][h].y))), fmaxf(fabsf(keep[i][h].z), fabsf(keep[i][h].w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < kSylThreads / 64; ++i) m = fmaxf(m, red[i]);
    float sc, un;
    pow2_scale(m, sc, un);
    if (tid == 0) wun[l] = un;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = tid + i * kSylThreads;
      if (e >= KS * NT * 64) continue;
      const float v[8] = {keep[i][0].x, keep[i][0].y, keep[i][0].z, keep[i][0].w,
                          keep[i][1].x, keep[i][1].y, keep[i][1].z, keep[i][1].w};
      f16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        _Float16 ph, pl;
        split2(v[j] * sc, ph, pl);
        hi[j] = ph;
        lo[j] = pl;
      }
      // (entry e of the [ks][t] grid -> fragment ((ks * NT + t) * 2 + piece), lane e & 63)
      wfrag[(l * kFragL + (e >> 6) * 2 + 0) * 64 + (e & 63)] = hi;
      wfrag[(l * kFragL + (e >> 6) * 2 + 1) * 64 + (e & 63)] = lo;
    }
  }
  for (int i = tid; i < 4 * NT * 4; i += kSylThreads) {   // accumulator order: [g][t * 4 + r]
    const int gg = i / (NT * 4), t = (i / 4) % NT, r = i & 3;
    bias[i] = a.bias ? a.bias[syl_feat(t, gg, r)] : 0.f;
    rdg[i] = kDense ? 0.f : a.rdiag[syl_feat(t, gg, r)];
  }
  __syncthreads();

  // B operand from this lane's NT * 4 values: scale by the row maximum, split
  auto make_operand = [&](const f32x4 (&v)[NT], f16x8 (&bh)[KS], f16x8 (&bl)[KS]) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[t][r]));
    m = rows4_allmax(m, lane);
    float sc, un;
    pow2_scale(m, sc, un);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 hh, ll;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t ph, pl;
        split2_pair(v[2 * ks + (q >> 1)][2 * (q & 1)], v[2 * ks + (q >> 1)][2 * (q & 1) + 1], sc, ph, pl);
        hh[q] = ph;
        ll[q] = pl;
      }
      bh[ks] = __builtin_bit_cast(f16x8, hh);
      bl[ks] = __builtin_bit_cast(f16x8, ll);
    }
    return un;
  };
  // acc = (scaled W_l) (scaled v)^T: three split terms, small ones first; consecutive MFMAs on different tiles / blocks
  constexpr int kChunk = NT % 4 == 0 ? 4 : 2;      // tiles whose high pieces are held at a time (NT = 2 KS)
  auto product = [&](int l, const f16x8 (&bh)[BPW][KS], const f16x8 (&bl)[BPW][KS], f32x4 (&acc)[BPW][NT]) {
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f16x8* wf = wfrag + (size_t)l * kFragL * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 wl = wf[((ks * NT + t) * 2 + 1) * 64];
#pragma unroll
        for (int b = 0; b < BPW; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[b][ks], acc[b][t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t0 = 0; t0 < NT; t0 += kChunk) {
        f16x8 wh[kChunk];
#pragma unroll
        for (int t = 0; t < kChunk; ++t) wh[t] = wf[((ks * NT + t0 + t) * 2 + 0) * 64];
#pragma unroll
        for (int t = 0; t < kChunk; ++t)
#pragma unroll
          for (int b = 0; b < BPW; ++b)
            acc[b][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], bl[b][ks], acc[b][t0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < kChunk; ++t)
#pragma unroll
          for (int b = 0; b < BPW; ++b)
            acc[b][t0 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], bh[b][ks], acc[b][t0 + t], 0, 0, 0);
      }
    }
  };

  const int64_t nwaves = (int64_t)gridDim.x * (kSylThreads / 64);
  // lane (s, g) holds features 32 ks + 8 g + j of sample s: tile 2 ks + (j >> 2), register j & 3
  auto load_rows = [&](int64_t blk, f32x4 (&dst)[NT]) {
    const float4* xrow = reinterpret_cast<const float4*>(a.x + (blk * 16 + s16) * F);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 v = xrow[8 * (t >> 1) + 2 * g + (t & 1)];
      dst[t] = f32x4{v.x, v.y, v.z, v.w};
    }
  };
  // group grp = blocks BPW grp .. BPW grp + BPW - 1; the last group of an odd count repeats its first block (computed twice,
  // stored once).  One block per wave: the next block's rows are requested before this block's products and land behind them
  // (round 4: the wave used to wait for every block's HBM latency with nothing else in flight); two blocks per wave keep two
  // requests in flight by themselves and have no registers left for a third.
  constexpr bool kPrefetch = BPW == 1;
  const int64_t groups = (a.blocks16 + BPW - 1) / BPW;
  auto blk_of = [&](int64_t grp, int b) {
    const int64_t blk = grp * BPW + b;
    return blk < a.blocks16 ? blk : a.blocks16 - 1;
  };
  const int64_t grp0 = (int64_t)blockIdx.x * (kSylThreads / 64) + wave;
  f32x4 znext[kPrefetch ? NT : 1];
  if constexpr (kPrefetch) {
    if (grp0 < groups) load_rows(grp0, znext);
  }
  for (int64_t grp = grp0; grp < groups; grp += nwaves) {
    asm volatile("" ::: "memory");   // keeps the loop-invariant LDS fragment loads inside the loop
    f32x4 z[BPW][NT];
    if constexpr (kPrefetch) {
#pragma unroll
      for (int t = 0; t < NT; ++t) z[0][t] = znext[t];
      if (grp + nwaves < groups) load_rows(grp + nwaves, znext);
    } else {
#pragma unroll
      for (int b = 0; b < BPW; ++b) load_rows(blk_of(grp, b), z[b]);
    }
    f16x8 bh[BPW][KS], bl[BPW][KS];
    f32x4 acc[BPW][NT];
    float un[BPW];
#pragma unroll
    for (int b = 0; b < BPW; ++b) un[b] = make_operand(z[b], bh[b], bl[b]);
    product(0, bh, bl, acc);
    const f32x4* bsrc = reinterpret_cast<const f32x4*>(bias + g * NT * 4);
    const f32x4* rsrc = reinterpret_cast<const f32x4*>(rdg + g * NT * 4);
    if constexpr (kDense) {
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        if (grp * BPW + b >= a.blocks16) continue;
        const float c = un[b] * wun[0];
        float4* yrow = reinterpret_cast<float4*>(a.y + ((grp * BPW + b) * 16 + s16) * F);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 bb = bsrc[t];
          yrow[8 * (t >> 1) + 2 * g + (t & 1)] =
              float4{__builtin_fmaf(acc[b][t][0], c, bb[0]), __builtin_fmaf(acc[b][t][1], c, bb[1]),
                     __builtin_fmaf(acc[b][t][2], c, bb[2]), __builtin_fmaf(acc[b][t][3], c, bb[3])};
        }
      }
      continue;
    }
    float lsum[BPW];
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      const float c = un[b] * wun[0];
      lsum[b] = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 bb = bsrc[t], rd = rsrc[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // (tanh / log on the lean primitives, one exponential / one v_log each: libm's were ~10 000 vector operations per
          //  row of 128 features, more than half of the kernel)
          const float act = tanh_lean(__builtin_fmaf(acc[b][t][r], c, bb[r]));     // tanh(R1 Q^T z + b)
          acc[b][t][r] = act;
          const float arg = 1.f + (1.f - act * act) * rd[r];
          lsum[b] += (arg > 0.f && arg < INFINITY) ? log_lean(arg) : logf(arg);   // planar.py:160-163
        }
      }
      un[b] = make_operand(acc[b], bh[b], bl[b]);
    }
    product(1, bh, bl, acc);
    if constexpr (BPW > 1) {      // (the rows are read again for the residual instead of living through both products: L2)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int b = 0; b < BPW; ++b) load_rows(blk_of(grp, b), z[b]);
    }
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      if (grp * BPW + b >= a.blocks16) continue;
      const float c = un[b] * wun[1];
      float4* yrow = reinterpret_cast<float4*>(a.y + ((grp * BPW + b) * 16 + s16) * F);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        yrow[8 * (t >> 1) + 2 * g + (t & 1)] = float4{z[b][t][0] + acc[b][t][0] * c, z[b][t][1] + acc[b][t][1] * c,
                                                      z[b][t][2] + acc[b][t][2] * c, z[b][t][3] + acc[b][t][3] * c};
      const float l = rows4_allsum(lsum[b], lane);
      if (g == 0) a.lad[(grp * BPW + b) * 16 + s16] = l;
    }
  }
}

template <int KS, bool kDense, int BPW>
static hipError_t launch_syl_bpw(const SylArgs& a, int cus, hipStream_t s) {
  constexpr int NT = 2 * KS;
  const size_t lds = (size_t)2 * KS * NT * 2 * 64 * 16 + (2 * 4 * NT * 4 + 16) * 4;
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(attr, reinterpret_cast<const void*>(&sylvester_mm_kernel<KS, kDense, BPW>),
                                               160 * 1024);
  if (ea != hipSuccess) return ea;
  int64_t grid = cus;
  const int64_t need = ((a.blocks16 + BPW - 1) / BPW + 7) / 8;
  if (grid > need) grid = need;
  hipLaunchKernelGGL((sylvester_mm_kernel<KS, kDense, BPW>), dim3((unsigned)grid), dim3(kSylThreads), lds, s, a);
  return hipGetLastError();
}

#ifndef FC_SYL_BPW
#define FC_SYL_BPW 1
#endif
template <int KS, bool kDense>
static hipError_t launch_syl(const SylArgs& a, int cus, hipStream_t s) {
  // two blocks per wave once every wave of the launch has a pair to carry
  if constexpr (FC_SYL_BPW == 2 && KS == 4 && !kDense) {
    if (a.blocks16 >= 2 * 8 * (int64_t)cus) return launch_syl_bpw<KS, kDense, 2>(a, cus, s);
  }
  return launch_syl_bpw<KS, kDense, 1>(a, cus, s);
}

}  // namespace fc

extern "C" int fc_sylvester_mm(const float* x, float* y, float* logabsdet, const float* w1, const float* w2,
                               const float* bias, const float* r_diag_prod, int64_t n, int32_t d, void* stream) {
  if (n < 0 || d <= 0 || d % 32 != 0 || d > 128 || n % 16 != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !logabsdet || !w1 || !w2 || !bias || !r_diag_prod) return hipErrorInvalidValue;
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w1 | (uintptr_t)w2) & 15u) != 0) return hipErrorInvalidValue;
  fc::SylArgs a{x, y, logabsdet, w1, w2, bias, r_diag_prod, n / 16};
  const int cus = fc::device_cu_count();
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (d / 32) {
    case 1: return fc::launch_syl<1, false>(a, cus * 2, s);
    case 2: return fc::launch_syl<2, false>(a, cus * 2, s);
    case 3: return fc::launch_syl<3, false>(a, cus, s);
    default: return fc::launch_syl<4, false>(a, cus, s);
  }
}

extern "C" int fc_dense_mm(const float* x, float* y, const float* w, const float* bias, int64_t n, int32_t d,
                           void* stream) {
  if (n < 0 || d <= 0 || d % 32 != 0 || d > 128 || n % 16 != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !w) return hipErrorInvalidValue;
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) & 15u) != 0) return hipErrorInvalidValue;
  fc::SylArgs a{x, y, nullptr, w, nullptr, bias, nullptr, n / 16};
  const int cus = fc::device_cu_count();
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (d / 32) {
    case 1: return fc::launch_syl<1, true>(a, cus * 2, s);
    case 2: return fc::launch_syl<2, true>(a, cus * 2, s);
    case 3: return fc::launch_syl<3, true>(a, cus, s);
    default: return fc::launch_syl<4, true>(a, cus, s);
  }
}
