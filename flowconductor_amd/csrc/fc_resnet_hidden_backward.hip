// Backward of the ResidualNet conditioner's hidden stack (fc_resnet_hidden.hip), one kernel, gfx950.
//
//   forward:   h0 = W0 x_id + b0;   per block:  a1 = relu(h);  t1 = W1 a1 + b1;  a2 = relu(t1);  h += W2 a2 + b2
//   backward:  given gh = dL/dh [N, 64]:  per block (last first)
//                  gW2 += gh (x) a2,  gb2 += gh;      g1 = (W2^T gh) . [t1 > 0]
//                  gW1 += g1 (x) a1,  gb1 += g1;      gh += (W1^T g1) . [h_in > 0]
//              gW0 += gh (x) x_id,  gb0 += gh;        gx_id = W0^T gh
//
// (what torch.autograd does for flowcon/nn/nets/resnet.py:39-53, 93-99 -- there 5 forward GEMMs, 10 backward GEMMs and
//  ~20 element-wise / reduction kernels per layer, each a pass over [N, 64] in HBM: 6 ms per coupling layer at
//  N = 2^19.)  Here the only HBM traffic is x and gh in, gx_id out: nothing was saved by the forward but x itself --
//  the activations are recomputed.
//
// A wave carries 16 samples through the recomputed forward and through the backward in registers, exactly like the
// forward kernel (products transposed, C layout of one layer = B operand of the next, split-f16 matrix-core products);
// W^T products use a second set of fragments (the same packing applied to the transposed weights), streamed from L2.
// The weight gradients contract over SAMPLES, which live on lanes: per layer the 8 waves of a workgroup write
// (g^T, a^T) for their 8 x 16 samples into a shared LDS image [feature][128 samples]; after a barrier every wave owns
// two of the layer's sixteen 16 x 16 tiles of gW and accumulates them over the 128 samples on the f32-input matrix
// instruction (v_mfma_f32_16x16x4_f32: exact f32 fma chains, no scaling needed for sums over the whole batch), in
// registers across the whole launch; one atomic add per element and workgroup at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "fc_device.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "fc_split.h"
#include "../../include/flowcon_hip.h"

namespace fc {

constexpr int kHbThreads = 512, kHbS = 128 /* samples per round */, kHbLd = kHbS + 4 /* row stride of the staging images */;

struct HidBwdArgs {
  const float* x;          // [N, D]
  const float* gh;         // [N, 64]
  const int32_t* id_cols;  // [k0]
  const f16x8* wf;         // forward fragments: layer 0 [K0S][4][2][64], then per layer [2][4][2][64]  (hid_feat row order)
  const f16x8* wt;         // transposed fragments: per hidden layer [2][4][2][64]; last: W0^T [2][2 K0S][2][64]
  const float* wun;        // [L] 2^-S per layer (shared by both fragment sets)
  const float* bias;       // [L][4][16] accumulator order
  float* gxid;             // [N, 32 K0S], or null when the gradient is added into gx_full
  float* gx_full;          // [N, D]: gx_full[:, id_cols] += the gradient wrt the identity columns (every (row, id column) is
                           // visited by exactly one lane: plain read-modify-write), or null
  float* gw0;              // [64][32 K0S]
  float* gwb;              // [2 blocks][64][64]... flat [(L - 1)][64][64]
  float* gb;               // [L][64]
  int64_t rounds;          // 128-sample rounds
  int D, k0;
};

__host__ __device__ constexpr int hb_feat(int t, int g, int r) { return 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r; }

template <int NB, int K0S>
__global__ __launch_bounds__(kHbThreads) void resnet_hidden_backward_kernel(HidBwdArgs a) {
  constexpr int L = 1 + 2 * NB;
  constexpr int kFrag0 = K0S * 4 * 2, kFragL = 2 * 4 * 2;             // fragments (1 KB each) of layer 0 / a hidden layer
  constexpr int kFragsF = kFrag0 + 2 * NB * kFragL;
  extern __shared__ __attribute__((aligned(16))) unsigned char hbsm[];
  f16x8* wfl = reinterpret_cast<f16x8*>(hbsm);                                    // forward fragments of all layers
  float* gT = reinterpret_cast<float*>(hbsm + (size_t)kFragsF * 64 * 16);         // [64][kHbLd]  g^T of the current layer
  float* aT = gT + 64 * kHbLd;                                                     // [64][kHbLd]  a^T
  float* biasl = aT + 64 * kHbLd;                                                  // [L][64]
  int* ids = reinterpret_cast<int*>(biasl + L * 64);                               // [32 K0S]
  // One layer's W^T fragments (16 KB), brought in by LDS-DMA while the layer's (g^T, a^T) images are being written:
  // one L2 read per workgroup instead of one per wave, its latency behind the staging.  (With 64 identity features the
  // forward fragments need the space: the W^T fragments then stream from L2 per wave.)
  constexpr bool kWtLds = K0S == 1;
  f16x8* wts = reinterpret_cast<f16x8*>(ids + 32 * K0S);                           // [16 fragments][64 lanes]

  // (the wave index as a scalar: tile and strip addresses derived from it stay SGPR bases)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s16_ = lane & 15, g_ = lane >> 4;
  const int s16 = s16_, g = g_;
  const int D = a.D, k0 = a.k0;
  if ((int64_t)blockIdx.x >= a.rounds) return;
  for (int i = tid; i < kFragsF * 64; i += kHbThreads) wfl[i] = a.wf[i];
  for (int i = tid; i < L * 64; i += kHbThreads) biasl[i] = a.bias[i];
  for (int i = tid; i < 32 * K0S; i += kHbThreads) ids[i] = i < k0 ? a.id_cols[i] : -1;
  __syncthreads();

  float wun[L];
#pragma unroll
  for (int l = 0; l < L; ++l) wun[l] = a.wun[l];
  // gx_full mode: are the identity columns every other column of a 16-byte aligned [N, 2 k0] gradient?  (0 / 1 = their
  // parity, -1 = no: scalar read-modify-writes)
  int alternating = -1;
  if (a.gx_full && D == 2 * k0 && k0 == 32 * K0S && (D & 3) == 0 && ((uintptr_t)a.gx_full & 15u) == 0 && (ids[0] == 0 || ids[0] == 1)) {
    alternating = ids[0];
    for (int j = 1; j < k0; ++j)
      if (ids[j] != 2 * j + ids[0]) alternating = -1;
  }

  auto make_operand = [&](const f32x4 (&v)[4], f16x8 (&bh)[2], f16x8 (&bl)[2]) __attribute__((always_inline)) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[t][r]));
    m = rows4_allmax(m, lane);
    float sc, un;
    pow2_scale(m, sc, un);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        _Float16 ph, pl;
        split2(v[t][r] * sc, ph, pl);
        bh[t >> 1][4 * (t & 1) + r] = ph;
        bl[t >> 1][4 * (t & 1) + r] = pl;
      }
    return un;
  };
  // acc[t] = (scaled fragments)(scaled operand)^T over nks k-steps; fragments at wf[((ks * NT + t) * 2 + piece) * 64]
  auto product = [&](const f16x8* wfr, int nks, auto NTc, const f16x8 (&bh)[2], const f16x8 (&bl)[2],
                     f32x4 (&acc)[4]) __attribute__((always_inline)) {
    constexpr int NT = decltype(NTc)::value;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      if (ks < nks) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f16x8 ah = wfr[((ks * NT + t) * 2 + 0) * 64], al = wfr[((ks * NT + t) * 2 + 1) * 64];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks], acc[t], 0, 0, 0);
        }
      }
  };
  using I4 = std::integral_constant<int, 4>;
  using I2K = std::integral_constant<int, 2 * K0S>;

  // weight-gradient accumulators of this wave: two 16 x 16 tiles per layer (tile = (out tile ot, in tile it))
  f32x4 dw[L][2];
#pragma unroll
  for (int l = 0; l < L; ++l) dw[l][0] = dw[l][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc[L];          // threads tid < 64: bias gradient of feature tid
#pragma unroll
  for (int l = 0; l < L; ++l) dbacc[l] = 0.f;

  // (g, a) of this wave's 16 samples -> the shared [feature][sample] images; barrier; this wave's tiles of gW_l
  auto weight_grad = [&](auto Lc, const f32x4 (&gv)[4], const f32x4 (&av)[4], int wt_off, int wt_frags) __attribute__((always_inline)) {
    constexpr int l = decltype(Lc)::value;
    // fresh copies of the lane coordinates behind an opaque asm: the staging / tile addresses of the five weight_grad phases are
    // then formed per phase instead of being hoisted out of the round loop as loop-invariant registers (round 4: the headline
    // instantiation <2, 1> 132 -> 0 B of scratch, <1, 2> 196 -> 68 B, <2, 2> 864 -> 656 B)
    int s16 = s16_, g = g_;
    asm volatile("" : "+v"(s16), "+v"(g));
    __syncthreads();        // the previous layer's tiles have been read, the previous W^T product is done
    if constexpr (kWtLds) {
      // fragments wt_off .. wt_off + wt_frags of the transposed image -> wts; wave w moves fragments w, w + 8
      for (int f = wave; f < wt_frags; f += kHbThreads / 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wt + (size_t)(wt_off + f) * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(wts + f * 64), 16, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = hb_feat(t, g, r);
        gT[f * kHbLd + 16 * wave + s16] = gv[t][r];
        aT[f * kHbLd + 16 * wave + s16] = av[t][r];
      }
    if constexpr (kWtLds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the W^T fragments landed
    __syncthreads();
    // contraction index k = sample: chunk c (4 samples per instruction), slot g' = lane >> 4  <->  sample 32 g' + c
    constexpr int kInTiles = l == 0 ? 2 * K0S : 4;
    const int tile0 = kInTiles == 2 ? wave : 2 * wave;                 // 8 or 16 tiles over 8 waves
    const int ot = tile0 / kInTiles, it0 = tile0 % kInTiles;
    const float* ga = gT + (16 * ot + s16) * kHbLd + 32 * g;
#ifndef FC_HB_ABL
#define FC_HB_ABL 0      // probe builds: 1 = no weight-gradient products (staging and barriers stay)
#endif
#pragma unroll
    for (int c4 = 0; c4 < ((FC_HB_ABL & 1) ? 0 : 8); ++c4) {
      const float4 av4 = *reinterpret_cast<const float4*>(ga + 4 * c4);
      const float avs[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
      for (int i = 0; i < (kInTiles == 2 ? 1 : 2); ++i) {
        const float4 bv4 = *reinterpret_cast<const float4*>(aT + (16 * (it0 + i) + s16) * kHbLd + 32 * g + 4 * c4);
        const float bvs[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) dw[l][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[j], bvs[j], dw[l][i], 0, 0, 0);
      }
    }
    if (tid < 64) {
      const float4* row = reinterpret_cast<const float4*>(gT + tid * kHbLd);
      float s = 0.f;
#pragma unroll 8
      for (int i = 0; i < kHbS / 4; ++i) {
        const float4 v = row[i];
        s += (v.x + v.y) + (v.z + v.w);
      }
      dbacc[l] += s;
    }
  };

  const f16x8* wtl = a.wt + lane;
  for (int64_t round = blockIdx.x; round < a.rounds; round += gridDim.x) {
    // the weight fragments are loop-invariant loads: without this fence the compiler hoists them out of the loop and spills
    asm volatile("" ::: "memory");
    const int64_t row = round * kHbS + 16 * wave + s16;
    // ---- recomputed forward ------------------------------------------------------------------------------------------
    f32x4 xin[4];
    {
      const float* xrow = a.x + row * D;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = hb_feat(t, g, r);
          float v = 0.f;
          if (k < 32 * K0S) {
            const int c = ids[k];
            v = c >= 0 ? xrow[c] : 0.f;
          }
          xin[t][r] = v;
        }
    }
    f16x8 bh[2], bl[2];
    f32x4 acc[4], h[4], hin[NB > 0 ? NB : 1][4], t1[NB > 0 ? NB : 1][4];
    auto finish = [&](int l, float un, const f32x4 (&ac)[4], f32x4 (&out)[4]) __attribute__((always_inline)) {
      const float c = un * wun[l];
      const f32x4* bsrc = reinterpret_cast<const f32x4*>(biasl + l * 64 + g * 16);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 b = bsrc[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] = __builtin_fmaf(ac[t][r], c, b[r]);
      }
    };
    float un = make_operand(xin, bh, bl);
    product(wfl + lane, K0S, I4{}, bh, bl, acc);
    finish(0, un, acc, h);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      f32x4 act[4], tmid[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        hin[b][t] = h[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) act[t][r] = fmaxf(h[t][r], 0.f);
      }
      un = make_operand(act, bh, bl);
      product(wfl + (kFrag0 + (2 * b) * kFragL) * 64 + lane, 2, I4{}, bh, bl, acc);
      finish(1 + 2 * b, un, acc, t1[b]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) act[t][r] = fmaxf(t1[b][t][r], 0.f);
      un = make_operand(act, bh, bl);
      product(wfl + (kFrag0 + (2 * b + 1) * kFragL) * 64 + lane, 2, I4{}, bh, bl, acc);
      finish(2 + 2 * b, un, acc, tmid);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] += tmid[t][r];
    }
    // ---- backward ----------------------------------------------------------------------------------------------------
    f32x4 gh[4];
    {
      const float4* grow = reinterpret_cast<const float4*>(a.gh + row * 64);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 v = grow[(hb_feat(t, g, 0)) >> 2];
        gh[t] = f32x4{v.x, v.y, v.z, v.w};
      }
    }
    auto step_back = [&](auto Bc) __attribute__((always_inline)) {
      constexpr int b = decltype(Bc)::value;
      using L2c = std::integral_constant<int, 2 + 2 * b>;
      using L1c = std::integral_constant<int, 1 + 2 * b>;
      f32x4 act[4], g1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) act[t][r] = fmaxf(t1[b][t][r], 0.f);
      weight_grad(L2c{}, gh, act, (2 * b + 1) * kFragL, kFragL);
      float ug = make_operand(gh, bh, bl);
      product(kWtLds ? wts + lane : wtl + (size_t)((2 * b + 1) * kFragL) * 64, 2, I4{}, bh, bl, acc);      // W2^T gh
      {
        const float c = ug * wun[2 + 2 * b];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) g1[t][r] = t1[b][t][r] > 0.f ? acc[t][r] * c : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) act[t][r] = fmaxf(hin[b][t][r], 0.f);
      weight_grad(L1c{}, g1, act, (2 * b) * kFragL, kFragL);
      ug = make_operand(g1, bh, bl);
      product(kWtLds ? wts + lane : wtl + (size_t)((2 * b) * kFragL) * 64, 2, I4{}, bh, bl, acc);          // W1^T g1
      {
        const float c = ug * wun[1 + 2 * b];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) gh[t][r] += hin[b][t][r] > 0.f ? acc[t][r] * c : 0.f;
      }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    if constexpr (NB >= 2) step_back(B1{});
    if constexpr (NB >= 1) step_back(B0{});
    using L0c = std::integral_constant<int, 0>;
    weight_grad(L0c{}, gh, xin, 2 * NB * kFragL, 2 * (2 * K0S) * 2);
    {
      const float ug = make_operand(gh, bh, bl);
      product(kWtLds ? wts + lane : wtl + (size_t)(2 * NB * kFragL) * 64, 2, I2K{}, bh, bl, acc);          // W0^T gh: rows = identity features
      const float c = ug * wun[0];
      if (a.gx_full) {
        float* grow = a.gx_full + row * a.D;
        if (alternating >= 0) {
          // identity columns 2 j + alternating (the alternating masks of a coupling stack): this lane's four values of
          // tile t belong to eight consecutive columns -- two 16-byte read-modify-writes instead of four 4-byte ones
#pragma unroll
          for (int t = 0; t < 2 * K0S; ++t) {
            float4* p4 = reinterpret_cast<float4*>(grow + 2 * (16 * t + 4 * g));
            float4 u = p4[0], v = p4[1];
            if (alternating == 0) {
              u.x += acc[t][0] * c; u.z += acc[t][1] * c; v.x += acc[t][2] * c; v.z += acc[t][3] * c;
            } else {
              u.y += acc[t][0] * c; u.w += acc[t][1] * c; v.y += acc[t][2] * c; v.w += acc[t][3] * c;
            }
            p4[0] = u;
            p4[1] = v;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 2 * K0S; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = 16 * t + 4 * g + r;
              if (j < k0) grow[ids[j]] += acc[t][r] * c;
            }
        }
      } else {
        float4* out = reinterpret_cast<float4*>(a.gxid + row * (32 * K0S));
#pragma unroll
        for (int t = 0; t < 2 * K0S; ++t) out[4 * t + g] = float4{acc[t][0] * c, acc[t][1] * c, acc[t][2] * c, acc[t][3] * c};
      }
    }
  }
  __syncthreads();
  // ---- weight / bias gradients of this workgroup ------------------------------------------------------------------------
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int in_tiles = l == 0 ? 2 * K0S : 4, in_w = l == 0 ? 32 * K0S : 64;
    const int tile0 = in_tiles == 2 ? wave : 2 * wave;
    const int ot = tile0 / in_tiles, it0 = tile0 % in_tiles;
    float* dst = l == 0 ? a.gw0 : a.gwb + (size_t)(l - 1) * 64 * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < (in_tiles == 2 ? 1 : 2))
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(dst + (size_t)(16 * ot + 4 * g + r) * in_w + 16 * (it0 + i) + s16, dw[l][i][r]);
    if (tid < 64) atomicAdd(a.gb + l * 64 + tid, dbacc[l]);
  }
}

template <int NB, int K0S>
static hipError_t launch_hid_bwd(const HidBwdArgs& a, hipStream_t s) {
  constexpr int L = 1 + 2 * NB;
  const size_t lds = (size_t)(K0S * 8 + 2 * NB * 16) * 64 * 16 + (size_t)2 * 64 * kHbLd * 4 + L * 64 * 4 + 32 * K0S * 4 +
                     (K0S == 1 ? 16 * 1024 : 0);
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(attr, reinterpret_cast<const void*>(&resnet_hidden_backward_kernel<NB, K0S>),
                                               160 * 1024);
  if (ea != hipSuccess) return ea;
  int64_t grid = device_cu_count();
  if (grid > a.rounds) grid = a.rounds;
  hipLaunchKernelGGL((resnet_hidden_backward_kernel<NB, K0S>), dim3((unsigned)grid), dim3(kHbThreads), lds, s, a);
  return hipGetLastError();
}

}  // namespace fc

static int hidden_backward_entry(const float* x, const float* grad_h, const int32_t* id_cols, const void* w_frag,
                                 const void* wt_frag, const float* w_unscale, const float* bias_acc, float* grad_x_id,
                                 float* grad_x_full, float* grad_w0, float* grad_wb, float* grad_b, int64_t n, int32_t d,
                                 int32_t in_features, int32_t hidden, int32_t num_blocks, int32_t activation, void* stream) {
  if (n < 0 || d <= 0 || hidden != 64 || num_blocks < 0 || num_blocks > 2 || activation != FC_ACT_RELU) return hipErrorInvalidValue;
  if (in_features <= 0 || in_features > 64 || in_features > d) return hipErrorInvalidValue;
  if (n % fc::kHbS != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !grad_h || !id_cols || !w_frag || !wt_frag || !w_unscale || !bias_acc || (!grad_x_id == !grad_x_full) || !grad_w0 ||
      !grad_b || (num_blocks > 0 && !grad_wb))
    return hipErrorInvalidValue;
  if ((((uintptr_t)grad_h | (uintptr_t)grad_x_id | (uintptr_t)w_frag | (uintptr_t)wt_frag) & 15u) != 0) return hipErrorInvalidValue;
  fc::HidBwdArgs a{x, grad_h, id_cols, static_cast<const fc::f16x8*>(w_frag), static_cast<const fc::f16x8*>(wt_frag),
                   w_unscale, bias_acc, grad_x_id, grad_x_full, grad_w0, grad_wb, grad_b, n / fc::kHbS, d, in_features};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool wide = in_features > 32;
  switch (num_blocks * 2 + (wide ? 1 : 0)) {
    case 0: return fc::launch_hid_bwd<0, 1>(a, s);
    case 1: return fc::launch_hid_bwd<0, 2>(a, s);
    case 2: return fc::launch_hid_bwd<1, 1>(a, s);
    case 3: return fc::launch_hid_bwd<1, 2>(a, s);
    case 4: return fc::launch_hid_bwd<2, 1>(a, s);
    default: return fc::launch_hid_bwd<2, 2>(a, s);
  }
}

extern "C" int fc_resnet_hidden_backward(const float* x, const float* grad_h, const int32_t* id_cols, const void* w_frag,
                                         const void* wt_frag, const float* w_unscale, const float* bias_acc,
                                         float* grad_x_id, float* grad_w0, float* grad_wb, float* grad_b, int64_t n,
                                         int32_t d, int32_t in_features, int32_t hidden, int32_t num_blocks,
                                         int32_t activation, void* stream) {
  return hidden_backward_entry(x, grad_h, id_cols, w_frag, wt_frag, w_unscale, bias_acc, grad_x_id, nullptr, grad_w0, grad_wb,
                               grad_b, n, d, in_features, hidden, num_blocks, activation, stream);
}

extern "C" int fc_resnet_hidden_backward_accum(const float* x, const float* grad_h, const int32_t* id_cols, const void* w_frag,
                                               const void* wt_frag, const float* w_unscale, const float* bias_acc,
                                               float* grad_x_full, float* grad_w0, float* grad_wb, float* grad_b, int64_t n,
                                               int32_t d, int32_t in_features, int32_t hidden, int32_t num_blocks,
                                               int32_t activation, void* stream) {
  return hidden_backward_entry(x, grad_h, id_cols, w_frag, wt_frag, w_unscale, bias_acc, nullptr, grad_x_full, grad_w0, grad_wb,
                               grad_b, n, d, in_features, hidden, num_blocks, activation, stream);
}
