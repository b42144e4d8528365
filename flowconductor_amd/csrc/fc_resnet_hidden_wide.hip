// Hidden layers of a WIDE ResidualNet conditioner (hidden_features 128 or 256; narrower widths zero-padded) as one
// kernel on the f16 matrix cores (split-f32 products, fc_split.h), gfx950.
//
//   h = W0 x_id + b0;   for each block:  h += W2 act(W1 act(h) + b1) + b2          -> h [N, H]
//
// (flowcon/nn/nets/resnet.py:39-53, 93-99; hidden_features is a free constructor argument there, :62, and NSF
//  conditioners are commonly 128-256 wide.)  fc_resnet_hidden.hip keeps the weight fragments of ALL layers in LDS and
//  lets one wave carry 16 samples through the stack; at H = 256 one layer alone is 256 KB of fragments, so here the
//  roles are turned round:
//    * a 512-thread workgroup owns a tile of 64 samples; the ACTIVATIONS of the tile live in LDS as ready-made B
//      operands (both f16 pieces, one power-of-two scale per sample row), shared by all 8 waves;
//    * wave w owns the output features [w H/8, (w+1) H/8) of every layer -- its accumulators, its slice of the
//      residual stream (registers) -- and streams the weight fragments of exactly those rows from L2 (each fragment is
//      used by one wave per tile; the packed image of all layers is <= 1.2 MB);
//    * between layers: per-row maximum across the waves by an LDS atomic max, then every wave scales, splits and
//      writes its slice of the next B operand.  Two barriers per layer, none inside a product.
// HBM traffic: the identity columns of x in, h out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_device.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "fc_split.h"
#include "../../include/flowcon_hip.h"

namespace fc {

constexpr int kWR = 64;          // rows per tile: four 16-sample blocks
constexpr int kWThreads = 512;
// Waves per SIMD the register budget is set for: 4 = 128 registers, TWO workgroups per CU (their LDS, 2 x 68 KB, always
// fitted; the H = 256 kernel's 152 registers did not).  The kernel waits on its weight stream from L2 for ~60 % of its
// wave cycles, so the second workgroup pays: 0.62 -> 0.50 ms per 2^18 rows at H = 256 despite 76 B of spills.
#ifndef FC_WIDE_WAVES
#define FC_WIDE_WAVES 4
#endif

struct WideArgs {
  const float* x;          // [N, D]
  float* h;                // [N, H]
  const int32_t* id_cols;  // [k0]
  const f16x8* wfrag;      // layer 0: [H/16][K0S][2][64]; then per layer [H/16][H/32][2][64]   (ops.pack_resnet_hidden_wide)
  const float* wun;        // [1 + 2 blocks]  2^-S of each layer's weight scale
  const float* bias;       // [1 + 2 blocks][H]
  int64_t tiles;           // 64-row tiles
  int D, k0, k0s, num_blocks, act;
  float act_param;
};

__device__ __forceinline__ float wide_act(float v, int act, float p) {
  switch (act) {
    case FC_ACT_TANH: return tanhf(v);
    case FC_ACT_SILU: return div_lean(v, 1.f + exp_lean(fminf(-v, 87.f)));
    case FC_ACT_ELU: return v > 0.f ? v : p * (exp_lean(v) - 1.f);
    case FC_ACT_LEAKY_RELU: return v > 0.f ? v : v * p;
    case FC_ACT_SIGMOID: return div_lean(1.f, 1.f + exp_lean(fminf(-v, 87.f)));
    default: return fmaxf(v, 0.f);
  }
}

// HQ = H / 64 (2 or 4); kRelu: the blocks' activation is ReLU (a v_max) / the one named by a.act
template <int HQ, bool kRelu>
__global__ __launch_bounds__(kWThreads, FC_WIDE_WAVES) void resnet_hidden_wide_kernel(WideArgs a) {
  constexpr int H = 64 * HQ, KS = H / 32, HB = H + 8, R = kWR;
  constexpr int TPW = H / 16 / 8;            // output tiles (16 features) per wave: 1 or 2
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  _Float16* abuf = reinterpret_cast<_Float16*>(wsm);                       // [2 pieces][R][HB]
  float* ascale = reinterpret_cast<float*>(wsm + (size_t)2 * R * HB * 2);   // [R] 2^-T of the initial layer's input rows
  unsigned* rowmax = reinterpret_cast<unsigned*>(ascale + R);               // [2][R] bit patterns of non-negative floats
  int* ids = reinterpret_cast<int*>(rowmax + 2 * R);                        // [64]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int s16 = lane & 15, g = lane >> 4;
  const int D = a.D, k0 = a.k0, K0S = a.k0s, L = 1 + 2 * a.num_blocks;
  if ((int64_t)blockIdx.x >= a.tiles) return;
  if (tid < 64) ids[tid] = tid < k0 ? a.id_cols[tid] : -1;
  if (tid < 2 * R) rowmax[tid] = 0u;
  __syncthreads();

  // fragment streams of this wave: layer 0, then the 64 x 64-style layers
  const size_t frag0 = (size_t)(H / 16) * K0S * 2 * 64;      // fragments of layer 0 (all waves)
  const size_t fragL = (size_t)(H / 16) * KS * 2 * 64;       // fragments of a hidden layer
  auto wbase = [&](int l) __attribute__((always_inline)) {
    const f16x8* base = l == 0 ? a.wfrag : a.wfrag + frag0 + (size_t)(l - 1) * fragL;
    const int nks = l == 0 ? K0S : KS;
    return base + (size_t)(wave * TPW) * nks * 2 * 64 + lane;
  };

  // acc[b][i] = sum over k of (scaled W_l rows of this wave)(scaled activation)^T, three split terms, small ones first
  auto product = [&](int l, f32x4 (&acc)[4][TPW]) __attribute__((always_inline)) {
    const int nks = l == 0 ? K0S : KS;
    const f16x8* w = wbase(l);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < TPW; ++i) acc[b][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 ah[TPW], al[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      ah[i] = w[((size_t)(i * nks + 0) * 2 + 0) * 64];
      al[i] = w[((size_t)(i * nks + 0) * 2 + 1) * 64];
    }
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
      f16x8 nh[TPW], nl[TPW];
      const int kn = ks + 1 < nks ? ks + 1 : ks;      // next k-step's fragments in flight during this one's MFMAs
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        nh[i] = w[((size_t)(i * nks + kn) * 2 + 0) * 64];
        nl[i] = w[((size_t)(i * nks + kn) * 2 + 1) * 64];
      }
      f16x8 bh[4], bl[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const _Float16* src = abuf + (size_t)(16 * b + s16) * HB + 32 * ks + 8 * g;
        bh[b] = *reinterpret_cast<const f16x8*>(src);
        bl[b] = *reinterpret_cast<const f16x8*>(src + (size_t)R * HB);
      }
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[b], acc[b][i], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[b], acc[b][i], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[b], acc[b][i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        ah[i] = nh[i];
        al[i] = nl[i];
      }
    }
  };
  // Linear output of layer l on this wave's features: undo both scalings, add the bias (one fma = one rounding)
  auto finish = [&](int l, const float (&un)[4], const f32x4 (&acc)[4][TPW], f32x4 (&out)[4][TPW]) __attribute__((always_inline)) {
    const float wu = a.wun[l];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const float4 bq = *reinterpret_cast<const float4*>(a.bias + (size_t)l * H + 16 * (wave * TPW + i) + 4 * g);
      const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float c = un[b] * wu;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[b][i][r] = __builtin_fmaf(acc[b][i][r], c, bv[r]);
      }
    }
  };
  // v (this wave's slice of an activation, all four blocks) -> B operand in LDS.  `par` alternates per call: the
  // row maxima of call n live in rowmax[par], those of call n - 1 are cleared meanwhile.  Returns 2^-T per block.
  auto publish = [&](const f32x4 (&v)[4][TPW], int par, float (&un)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[b][i][r]));
      m = rows4_allmax(m, lane);
      if (g == 0) atomicMax(&rowmax[par * R + 16 * b + s16], __float_as_uint(m));   // non-negative floats order as uints
    }
    __syncthreads();     // every wave has finished reading the previous B operand and contributed its maxima
    if (tid < R) rowmax[(par ^ 1) * R + tid] = 0u;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float sc;
      pow2_scale(__uint_as_float(rowmax[par * R + 16 * b + s16]), sc, un[b]);
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        f16x4 p0, p1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          _Float16 ph, pl;
          split2(v[b][i][r] * sc, ph, pl);
          p0[r] = ph;
          p1[r] = pl;
        }
        _Float16* dst = abuf + (size_t)(16 * b + s16) * HB + 16 * (wave * TPW + i) + 4 * g;
        *reinterpret_cast<f16x4*>(dst) = p0;
        *reinterpret_cast<f16x4*>(dst + (size_t)R * HB) = p1;
      }
    }
    __syncthreads();
  };
  auto activate = [&](const f32x4 (&in)[4][TPW], f32x4 (&out)[4][TPW]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[b][i][r] = kRelu ? fmaxf(in[b][i][r], 0.f) : wide_act(in[b][i][r], a.act, a.act_param);
  };

  int par = 0;
  for (int64_t tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    // ---- identity columns of the tile -> B operand of the initial layer: 16 lanes per row, 4 columns each, two
    // passes of 32 rows; zero padding up to 32 K0S columns ----------------------------------------------------------
    __syncthreads();     // the previous tile's last product has read abuf
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = 32 * pass + (tid >> 4), c0 = 4 * (tid & 15);
      const float* xr = a.x + (tile * R + row) * (int64_t)D;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = ids[c0 + j];
        v[j] = col >= 0 ? xr[col] : 0.f;
      }
      const float m = row16_allmax(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      float sc, un0;
      pow2_scale(m, sc, un0);
      if (c0 < 32 * K0S) {
        f16x4 p0, p1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          _Float16 ph, pl;
          split2(v[j] * sc, ph, pl);
          p0[j] = ph;
          p1[j] = pl;
        }
        _Float16* dst = abuf + (size_t)row * HB + c0;
        *reinterpret_cast<f16x4*>(dst) = p0;
        *reinterpret_cast<f16x4*>(dst + (size_t)R * HB) = p1;
      }
      if ((tid & 15) == 0) ascale[row] = un0;
    }
    __syncthreads();

    f32x4 acc[4][TPW], hres[4][TPW], tmp[4][TPW], act[4][TPW];
    float un[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) un[b] = ascale[16 * b + s16];
    product(0, acc);
    finish(0, un, acc, hres);
#pragma unroll 1
    for (int blk = 0; blk < a.num_blocks; ++blk) {
      activate(hres, act);
      publish(act, par, un);
      par ^= 1;
      product(1 + 2 * blk, acc);
      finish(1 + 2 * blk, un, acc, tmp);
      activate(tmp, act);
      publish(act, par, un);
      par ^= 1;
      product(2 + 2 * blk, acc);
      finish(2 + 2 * blk, un, acc, tmp);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) hres[b][i][r] += tmp[b][i][r];      // resnet.py:52 `inputs + temps`
    }
    (void)L;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        float4* dst = reinterpret_cast<float4*>(a.h + (tile * R + 16 * b + s16) * (int64_t)H + 16 * (wave * TPW + i) + 4 * g);
        *dst = float4{hres[b][i][0], hres[b][i][1], hres[b][i][2], hres[b][i][3]};
      }
  }
}

template <int HQ, bool kRelu>
static hipError_t launch_wide(const WideArgs& a, hipStream_t s) {
  constexpr int H = 64 * HQ;
  const size_t lds = (size_t)2 * kWR * (H + 8) * 2 + kWR * 4 + 2 * kWR * 4 + 64 * 4;
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(attr, reinterpret_cast<const void*>(&resnet_hidden_wide_kernel<HQ, kRelu>),
                                               160 * 1024);
  if (ea != hipSuccess) return ea;
  // two workgroups per CU where two activation images fit (H = 128: 35 KB each; H = 256: 68 KB each)
  int64_t grid = (int64_t)device_cu_count() * 2;
  if (grid > a.tiles) grid = a.tiles;
  hipLaunchKernelGGL((resnet_hidden_wide_kernel<HQ, kRelu>), dim3((unsigned)grid), dim3(kWThreads), lds, s, a);
  return hipGetLastError();
}

}  // namespace fc

extern "C" int fc_resnet_hidden_wide(const float* x, float* h, const int32_t* id_cols, const void* w_frag,
                                     const float* w_unscale, const float* bias, int64_t n, int32_t d,
                                     int32_t in_features, int32_t hidden, int32_t num_blocks, int32_t activation,
                                     float activation_param, void* stream) {
  if (n < 0 || d <= 0 || (hidden != 128 && hidden != 256) || num_blocks < 0 || num_blocks > 16) return hipErrorInvalidValue;
  if (activation < FC_ACT_RELU || activation > FC_ACT_SIGMOID) return hipErrorInvalidValue;
  if (in_features <= 0 || in_features > 64 || in_features > d) return hipErrorInvalidValue;
  if (n % fc::kWR != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !h || !id_cols || !w_frag || !w_unscale || !bias) return hipErrorInvalidValue;
  if ((((uintptr_t)h | (uintptr_t)w_frag | (uintptr_t)bias) & 15u) != 0) return hipErrorInvalidValue;
  fc::WideArgs a{x, h, id_cols, static_cast<const fc::f16x8*>(w_frag), w_unscale, bias, n / fc::kWR, d, in_features,
                 in_features > 32 ? 2 : 1, num_blocks, activation, activation_param};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool relu = activation == FC_ACT_RELU;
  if (hidden == 128) return relu ? fc::launch_wide<2, true>(a, s) : fc::launch_wide<2, false>(a, s);
  return relu ? fc::launch_wide<4, true>(a, s) : fc::launch_wide<4, false>(a, s);
}
