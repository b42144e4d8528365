// Hidden layers of the ResidualNet conditioner as ONE kernel on the f16 matrix cores (split-f32), gfx950.
//
//   h = W0 x_id + b0;   for each block:  h += W2 relu(W1 relu(h) + b1) + b2          -> h [N, 64]
//
// (flowcon/nn/nets/resnet.py:39-53, 93-99: initial_layer, then pre-activation residual blocks; no
//  context, no batch norm, dropout inactive.)  In PyTorch this is 5 small GEMMs + 9 element-wise
//  kernels + a gather per layer, each a full pass over [N, 64] in HBM; here the only HBM traffic is
//  the x rows in and the h rows out.  The conditioner stays a PyTorch nn.Module (parameters,
//  state_dict, CPU execution); this kernel is the device fast path for its inference forward when
//  the shapes match: hidden = 64, <= 4 blocks, ReLU, <= 64 input features.
//
// With a context (resnet.py:48-49, 94-97; kCtx = 1): the initial layer sees [x_id | context] and every block gates
// its output,  h += (W2 relu(W1 relu(h) + b1) + b2) * sigmoid(Wc context + bc)   (F.glu of the concatenation).
// The context row is one more B operand (<= 32 features: one k-step), split once per 16-sample block and used by
// the gate product of every residual block (12 MFMAs each).
// kCtx = 2 is the MADE form (made.py:100-140, 239-246): the context enters additively,
//   h = W0 x + b0 + act(Wc0 c + bc0);   per block:  h += W2 act(W1 act(h) + b1 + Wc c + bc) + b2
// (the same context operand, one more context product for the initial layer).
//
// Every product runs as three v_mfma_f32_16x16x32_f16 terms on scaled two-piece f16 splits of both
// operands (fc_split.h): f32-GEMM accuracy at 3/16 of the f32-MFMA cycles.
//
// A wave owns 16 samples and pushes them through ALL layers by itself -- no LDS hand-off of
// activations, no barrier in the loop.  The products are taken transposed (A = weight rows, B = act^T),
// so the C layout gives lane (s = lane & 15, g = lane >> 4) 16 features of sample s; the weight rows of
// each layer are ordered so that these are exactly the 16 k-values the lane must supply as B operand
// of the next layer:   tile t, row 4g + r  <->  feature 32 (t >> 1) + 8 g + 4 (t & 1) + r
//                      B fragment of k-step ks, element j  =  accumulator tile 2 ks + (j >> 2), register j & 3.
// Between layers a lane therefore only does bias + ReLU + split on its own 16 registers; the row maximum
// for the scaling takes two cross-lane steps.  The residual stream h lives in 16 registers per lane.
// Weights: both f16 pieces of all layers as ready-made A fragments in LDS (72 KB at <= 32 inputs),
// built once per workgroup from the row-major f32 weights.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_split.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "fc_device.h"
#include "../../include/flowcon_hip.h"

#ifndef FC_HIDDEN_PREFETCH_X
#define FC_HIDDEN_PREFETCH_X 1
#endif

namespace fc {

constexpr int kHid = 64;
constexpr int kHidThreads = 512;

struct HiddenArgs {
  const float* x;         // [N, D]
  float* h;               // [N, 64]
  const int32_t* id_cols; // [k0] identity columns of x feeding the conditioner
  const float* w0;        // [64, k0]      initial_layer.weight, row-major
  const float* b0;        // [64]
  const float* wb;        // [blocks][2][64][64]  linear_layers[0/1].weight of each block, row-major
  const float* bb;        // [blocks][2][64]
  int64_t blocks16;       // number of 16-row blocks
  int D;
  int k0;                 // identity features read from x (k0 + C <= 64)
  const float* ctx;       // [N, C] context rows, or null
  const float* wc;        // [blocks][64][C]  context_layer.weight of each block
  const float* bc;        // [blocks][64]
  int C;                  // context features (<= 32), 0 without context
  int act;                // FC_ACT_* (kAct == 1 kernels only; kAct == 0 is ReLU)
  float act_param;        // ELU alpha / LeakyReLU negative slope
  // fc_resnet_hidden_packed: the LDS weight image made ahead of time (fc_pack_fragments, FC_PACK_HIDDEN jobs), or null
  const f16x8* image;     // [layer][ks][t][piece][lane] fragments = the layout of `wfrag` below
  const float* image_un;  // [layers] 2^-S of every layer
  const float* image_bias; // [layers][64] biases in accumulator order
};

// feature held by accumulator tile t, register r of a lane in group g
__host__ __device__ constexpr int hid_feat(int t, int g, int r) { return 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r; }

// LDS: [layer][k-step][tile][piece][lane] f16x8 fragments, then bias [layer][g][16], unscale [layer], ids
template <int NB, int K0S, int kCtx>
struct HiddenLds {
  static constexpr int kMain = 1 + 2 * NB;                // initial layer + two per block
  static constexpr int kCtxLayers = kCtx == 1 ? NB : kCtx == 2 ? NB + 1 : 0;   // context products: per block (+ initial)
  static constexpr int kLayers = kMain + kCtxLayers;
  static constexpr int kFrag0 = K0S * 4 * 2;              // fragments of the initial layer
  static constexpr int kFragL = 2 * 4 * 2;                // fragments of a 64 x 64 layer
  static constexpr int kFragG = 1 * 4 * 2;                // fragments of a 64 x C gate layer (C <= 32)
  static constexpr int kFragsMain = kFrag0 + 2 * NB * kFragL;
  static constexpr int kFrags = kFragsMain + kCtxLayers * kFragG;
  static constexpr size_t kBytes = (size_t)kFrags * 64 * 16 + kLayers * 64 * 4 + 16 * 4 + 32 * K0S * 4 + 16 * 8 * 4;
  static_assert(kLayers <= 16, "wun holds 16 entries");
  static_assert(kBytes <= 160 * 1024, "weight fragments exceed the CU's LDS");
};

// kAct: 0 = ReLU (the north-star conditioner; nothing but a v_max), 1 = the activation named by a.act
// BPW: 16-sample blocks a wave pushes through the layers together.  Every layer's weight fragments (16 KB per wave) come
// from LDS once per group of BPW blocks; at one block per wave the 16 waves of a CU ask the LDS pipe for as many cycles
// as their matrix and vector instructions take to issue, and the reads sit right before the MFMAs that need them.
template <int NB, int K0S, int kCtx, int kAct, int BPW>
__global__ __launch_bounds__(512, (kCtx || BPW > 1) ? 2 : 4) void resnet_hidden_kernel(HiddenArgs a) {
  using L = HiddenLds<NB, K0S, kCtx>;
  constexpr bool kPrefetchX = FC_HIDDEN_PREFETCH_X && (K0S == 1 || kCtx != 0 || BPW > 1);   // (16 more live registers spill in the 64-input kernels without a context at one block per wave: 128-register budget)
  extern __shared__ __attribute__((aligned(16))) unsigned char hsmem[];
  f16x8* wfrag = reinterpret_cast<f16x8*>(hsmem);
  float* bias = reinterpret_cast<float*>(hsmem + (size_t)L::kFrags * 64 * 16);   // [layer][g][16]
  float* wun = bias + L::kLayers * 64;                                             // [layer] (padded to 16)
  int* ids = reinterpret_cast<int*>(wun + 16);                                      // [32 K0S]
  float* red = reinterpret_cast<float*>(ids + 32 * K0S);                           // [layer][8 waves]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s16 = lane & 15, g = lane >> 4;
  const int k0 = a.k0, D = a.D, C = kCtx ? a.C : 0;
#ifdef FC_HID_STAMP
  const uint64_t stamp_entry = __builtin_amdgcn_s_memtime();
#endif

  // ---- once per workgroup: scale, split and lay out the weights -----------------------------------
  // operand column i of the initial layer: identity column of x (>= 0), context feature -2 - j, or padding (-1)
  const int C0 = kCtx == 1 ? C : 0;   // context features concatenated into the initial layer (ResidualNet form only)
  for (int i = tid; i < 32 * K0S; i += kHidThreads) ids[i] = i < k0 ? a.id_cols[i] : (i < k0 + C0 ? -2 - (i - k0) : -1);
  if (kCtx == 0 && a.image) {
    // ready-made image: straight into LDS (1 KiB per wave instruction), no arithmetic, one barrier.  Building it
    // here from the f32 weights costs ~34 000 cycles per launch (two rounds of loads, the split, two barriers).
    for (int f = wave; f < L::kFrags; f += kHidThreads / 64)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.image + (size_t)f * 64 + lane),
                                       (__attribute__((address_space(3))) void*)(wfrag + f * 64), 16, 0, 0);
    for (int i = tid; i < L::kLayers * 64; i += kHidThreads) bias[i] = a.image_bias[i];
    if (tid < L::kLayers) wun[tid] = a.image_un[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
  // Two rounds of global loads for ALL layers together (maxima, then fragments) with one barrier pair between them:
  // layer by layer the dependent load latencies and barriers of 5-13 layers cost ~15 us per launch.
  auto layer_src = [&](int l, const float*& w, const float*& b, int& kin, int& nks, int& base) {
    const bool gate = l >= L::kMain;
    w = l == 0 ? a.w0 : gate ? a.wc + (size_t)(l - L::kMain) * kHid * C : a.wb + (size_t)(l - 1) * kHid * kHid;
    b = l == 0 ? a.b0 : gate ? a.bc + (size_t)(l - L::kMain) * kHid : a.bb + (size_t)(l - 1) * kHid;
    kin = l == 0 ? k0 + C0 : gate ? C : kHid;
    nks = l == 0 ? K0S : gate ? 1 : 2;
    base = l == 0 ? 0 : gate ? L::kFragsMain + (l - L::kMain) * L::kFragG : L::kFrag0 + (l - 1) * L::kFragL;
  };
  float wmax[L::kLayers];
#pragma unroll
  for (int l = 0; l < L::kLayers; ++l) {
    const float *w, *b;
    int kin, nks, base;
    layer_src(l, w, b, kin, nks, base);
    float m = 0.f;
    if (kin == kHid) {
      // 64 x 64 layer: 8 consecutive weights per thread as two independent 16-byte loads
      const float4* w4 = reinterpret_cast<const float4*>(w) + 2 * tid;
      const float4 p = w4[0], q = w4[1];
      m = fmaxf(fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fmaxf(fabsf(p.z), fabsf(p.w))),
                fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w))));
    } else {
      // four loads in flight per step (a plain accumulation loop waits for every load in turn)
      const int total = kHid * kin;
      for (int i = tid; i < total; i += 4 * kHidThreads) {
        const float v0 = w[i];
        const float v1 = i + kHidThreads < total ? w[i + kHidThreads] : 0.f;
        const float v2 = i + 2 * kHidThreads < total ? w[i + 2 * kHidThreads] : 0.f;
        const float v3 = i + 3 * kHidThreads < total ? w[i + 3 * kHidThreads] : 0.f;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
      }
    }
    wmax[l] = m;
  }
#pragma unroll
  for (int l = 0; l < L::kLayers; ++l) {
    float m = wmax[l];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[l * (kHidThreads / 64) + wave] = m;
  }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < L::kLayers; ++l) {
    const float *w, *b;
    int kin, nks, base;
    layer_src(l, w, b, kin, nks, base);
    float m = red[l * (kHidThreads / 64)];
#pragma unroll
    for (int i = 1; i < kHidThreads / 64; ++i) m = fmaxf(m, red[l * (kHidThreads / 64) + i]);
    float sc, un;
    pow2_scale(m, sc, un);
    if (tid == 0) wun[l] = un;
    // fragment entry e = (ks * 4 + t) * 64 + lane': W[feat(t, lane' & 15)][32 ks + 8 (lane' >> 4) + j]
    for (int e = tid; e < nks * 4 * 64; e += kHidThreads) {
      const int ln = e & 63, t = (e >> 6) & 3, ks = e >> 8;
      const int rho = ln & 15, f = hid_feat(t, rho >> 2, rho & 3);
      f16x8 hi, lo;
      float v[8];
      if (kin == kHid) {   // 8 consecutive weights of a row, 32-byte aligned: two 16-byte loads
        const float4* w4 = reinterpret_cast<const float4*>(w + (size_t)f * kHid + 32 * ks + 8 * (ln >> 4));
        const float4 p = w4[0], q = w4[1];
        v[0] = p.x, v[1] = p.y, v[2] = p.z, v[3] = p.w, v[4] = q.x, v[5] = q.y, v[6] = q.z, v[7] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 32 * ks + 8 * (ln >> 4) + j;
          v[j] = k < kin ? w[(size_t)f * kin + k] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        _Float16 ph, pl;
        split2(v[j] * sc, ph, pl);
        hi[j] = ph;
        lo[j] = pl;
      }
      wfrag[(base + (ks * 4 + t) * 2 + 0) * 64 + ln] = hi;
      wfrag[(base + (ks * 4 + t) * 2 + 1) * 64 + ln] = lo;
    }
    // bias in accumulator order: [g][t * 4 + r]
    for (int i = tid; i < 64; i += kHidThreads) {
      const int gg = i >> 4, t = (i >> 2) & 3, r = i & 3;
      bias[l * 64 + i] = b[hid_feat(t, gg, r)];
    }
  }
  __syncthreads();
  }   // weights built in the kernel

  // the blocks' activation (resnet.py:42,46) on a lane's 16 values: ReLU, or what the module was built with (one
  // uniform switch per site, a straight 16-element loop inside each case)
  auto activate16 = [&](const f32x4 (&in)[4], f32x4 (&out)[4]) {
#define FC_ACT_LOOP(expr)                                            \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                      \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {                    \
    const float v = in[t][r];                                        \
    out[t][r] = (expr);                                              \
  }
    if constexpr (kAct == 0) {
      FC_ACT_LOOP(fmaxf(v, 0.f))
    } else {
      switch (a.act) {
        case FC_ACT_TANH: FC_ACT_LOOP(tanhf(v)) break;
        case FC_ACT_SILU: FC_ACT_LOOP(div_lean(v, 1.f + exp_lean(fminf(-v, 87.f)))) break;          // x * sigmoid(x)
        case FC_ACT_ELU: FC_ACT_LOOP(v > 0.f ? v : a.act_param * (exp_lean(v) - 1.f)) break;       // ATen: (exp(x) - 1) * alpha
        case FC_ACT_LEAKY_RELU: FC_ACT_LOOP(v > 0.f ? v : v * a.act_param) break;
        case FC_ACT_SIGMOID: FC_ACT_LOOP(div_lean(1.f, 1.f + exp_lean(fminf(-v, 87.f)))) break;
        default: FC_ACT_LOOP(fmaxf(v, 0.f)) break;
      }
    }
#undef FC_ACT_LOOP
  };
  // B operand of one layer from this lane's 16 activations v[t][r]: scale by the row maximum, split
  auto make_operand = [&](const f32x4 (&v)[4], f16x8 (&bh)[2], f16x8 (&bl)[2]) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[t][r]));
    m = rows4_allmax(m, lane);
    float sc, un;
    pow2_scale(m, sc, un);
    u32x4 hh[2], ll[2];     // f16 pairs: tile t, registers 2p, 2p + 1 -> k-step t >> 1, elements 4 (t & 1) + 2p, + 1
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        uint32_t ph, pl;
        split2_pair(v[t][2 * p], v[t][2 * p + 1], sc, ph, pl);
        hh[t >> 1][2 * (t & 1) + p] = ph;
        ll[t >> 1][2 * (t & 1) + p] = pl;
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bh[ks] = __builtin_bit_cast(f16x8, hh[ks]);
      bl[ks] = __builtin_bit_cast(f16x8, ll[ks]);
    }
    return un;
  };
  // acc[b] = (scaled W_l) (scaled act_b)^T for the BPW sample blocks of this wave: three split terms, small ones
  // first; every weight fragment is read from LDS once and serves all blocks
  auto layer = [&](int base, int nks, const f16x8 (&bh)[BPW][2], const f16x8 (&bl)[BPW][2], f32x4 (&acc)[BPW][4]) {
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f16x8* wf = wfrag + base * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      if (ks < nks) {
        f16x8 wl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wl[t] = wf[((ks * 4 + t) * 2 + 1) * 64];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int b = 0; b < BPW; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], bh[b][ks], acc[b][t], 0, 0, 0);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      if (ks < nks) {
        f16x8 wh[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wh[t] = wf[((ks * 4 + t) * 2 + 0) * 64];
        // consecutive MFMAs go to different accumulators
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int b = 0; b < BPW; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], bl[b][ks], acc[b][t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int b = 0; b < BPW; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], bh[b][ks], acc[b][t], 0, 0, 0);
      }
  };
  // Linear output of layer l: undo both scalings and add the bias in one fma (one rounding, as the GEMM's
  // own bias epilogue)
  auto finish = [&](int l, float un_act, const f32x4 (&acc)[4], f32x4 (&out)[4]) {
    const float c = un_act * wun[l];
    const f32x4* bsrc = reinterpret_cast<const f32x4*>(bias + l * 64 + g * 16);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 b = bsrc[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][r] = __builtin_fmaf(acc[t][r], c, b[r]);
    }
  };

  // identity columns this lane reads: k = 32 ks + 8 g + j
  int mycol[K0S][8];
#pragma unroll
  for (int ks = 0; ks < K0S; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) mycol[ks][j] = ids[32 * ks + 8 * g + j];

  const int64_t nwaves = (int64_t)gridDim.x * (kHidThreads / 64);
  // identity (and concatenated context) features of sample s for this lane, laid out like an activation tile
  auto gather = [&](int64_t blk, f32x4 (&xv)[4]) {
    const float* xrow = a.x + (blk * 16 + s16) * D;
    const float* crow_ = kCtx ? a.ctx + (blk * 16 + s16) * C : nullptr;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ks = t >> 1, j = 4 * (t & 1) + r;
        float v = 0.f;
        if (ks < K0S) {
          const int c = mycol[ks < K0S ? ks : 0][j];
          if constexpr (kCtx == 1)
            v = c >= 0 ? xrow[c] : (c <= -2 ? crow_[-2 - c] : 0.f);
          else
            v = c >= 0 ? xrow[c] : 0.f;
        }
        xv[t][r] = v;
      }
  };
  // A wave walks groups of BPW consecutive 16-sample blocks; a group that reaches past the end repeats the last
  // block (computed twice, stored once).
  const int64_t groups = (a.blocks16 + BPW - 1) / BPW;
  const int64_t grp0 = (int64_t)blockIdx.x * (kHidThreads / 64) + wave;
  auto block_of = [&](int64_t grp, int b) {
    const int64_t blk = grp * BPW + b;
    return blk < a.blocks16 ? blk : a.blocks16 - 1;
  };
  f32x4 xpre[kPrefetchX ? BPW : 1][kPrefetchX ? 4 : 1];       // the next group's rows, one iteration ahead
  if constexpr (kPrefetchX) {
    if (grp0 < groups) {
#pragma unroll
      for (int b = 0; b < BPW; ++b) gather(block_of(grp0, b), xpre[b]);
    }
  }
#define FC_EACH_BLOCK _Pragma("unroll") for (int b = 0; b < BPW; ++b)
#ifdef FC_HID_STAMP   // probe builds only (tools/probe/hidden_clock.py): cycles a wave spends in its loop
  const uint64_t stamp0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int64_t grp = grp0; grp < groups; grp += nwaves) {
    // the weight fragments are loop-invariant LDS loads: without this fence the compiler hoists all of them
    // out of the loop and spills
    asm volatile("" ::: "memory");
    f32x4 xin[BPW][4];
    FC_EACH_BLOCK {
      if constexpr (kPrefetchX) {
#pragma unroll
        for (int t = 0; t < 4; ++t) xin[b][t] = xpre[b][t];
      } else {
        gather(block_of(grp, b), xin[b]);
      }
    }
    f16x8 bh[BPW][2], bl[BPW][2];
    f32x4 acc[BPW][4], h[BPW][4], tmid[BPW][4];
    float un[BPW];
    // context row as the B operand of the gate products: lane (s, g) supplies features 8g..8g+7 of sample s; split
    // once per 16-sample block, used by every residual block.  (With a context the weight image, 96 KB at 2 blocks,
    // allows one 8-wave workgroup per CU: 256 registers per wave, these 9 stay live for free.  12- and 16-wave
    // workgroups sharing the image were tried: the compiler spills at 168 / 128 registers and they run 15 % slower.)
    f16x8 ch[BPW][2] = {}, cl[BPW][2] = {};
    float unc[BPW];
    FC_EACH_BLOCK unc[b] = 1.f;
    if constexpr (kCtx) {
      FC_EACH_BLOCK {
        const float* crow = a.ctx + (block_of(grp, b) * 16 + s16) * C;
        float cv[8], m = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * g + j;
          cv[j] = crow[k < C ? k : 0];
          cv[j] = k < C ? cv[j] : 0.f;
          m = fmaxf(m, fabsf(cv[j]));
        }
        m = rows4_allmax(m, lane);
        float sc;
        pow2_scale(m, sc, unc[b]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          _Float16 ph, pl;
          split2(cv[j] * sc, ph, pl);
          ch[b][0][j] = ph;
          cl[b][0][j] = pl;
        }
      }
    }
    FC_EACH_BLOCK un[b] = make_operand(xin[b], bh[b], bl[b]);
    // the next group's rows are requested now and consumed an iteration later (a wave otherwise starts every group
    // with an exposed memory latency)
    if constexpr (kPrefetchX) {
      const int64_t nxt = grp + nwaves < groups ? grp + nwaves : grp;
      FC_EACH_BLOCK gather(block_of(nxt, b), xpre[b]);
    }
    layer(0, K0S, bh, bl, acc);
    FC_EACH_BLOCK finish(0, un[b], acc[b], h[b]);
    if constexpr (kCtx == 2) {
      // made.py:243-244: temps = initial_layer(inputs) + activation(context_layer(context))
      layer(L::kFragsMain, 1, ch, cl, acc);
      FC_EACH_BLOCK {
        f32x4 cpre[4], cact[4];
        finish(L::kMain, unc[b], acc[b], cpre);
        activate16(cpre, cact);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[b][t][r] += cact[t][r];
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      FC_EACH_BLOCK {
        f32x4 act[4];
        activate16(h[b], act);
        un[b] = make_operand(act, bh[b], bl[b]);
      }
      layer(L::kFrag0 + (2 * nb) * L::kFragL, 2, bh, bl, acc);
      FC_EACH_BLOCK finish(1 + 2 * nb, un[b], acc[b], tmid[b]);
      if constexpr (kCtx == 2) {
        // made.py:131-132: temps = linear_layers[0](...) + context_layer(context)
        layer(L::kFragsMain + (1 + nb) * L::kFragG, 1, ch, cl, acc);
        FC_EACH_BLOCK {
          f32x4 cpre[4];
          finish(L::kMain + 1 + nb, unc[b], acc[b], cpre);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) tmid[b][t][r] += cpre[t][r];
        }
      }
      FC_EACH_BLOCK {
        f32x4 act[4];
        activate16(tmid[b], act);
        un[b] = make_operand(act, bh[b], bl[b]);
      }
      layer(L::kFrag0 + (2 * nb + 1) * L::kFragL, 2, bh, bl, acc);
      FC_EACH_BLOCK finish(2 + 2 * nb, un[b], acc[b], tmid[b]);
      if constexpr (kCtx == 1) {
        // resnet.py:48-49: temps = glu(cat(temps, context_layer(context))) = temps * sigmoid(Wc c + bc)
        layer(L::kFragsMain + nb * L::kFragG, 1, ch, cl, acc);
        FC_EACH_BLOCK {
          f32x4 gpre[4];
          finish(L::kMain + nb, unc[b], acc[b], gpre);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              tmid[b][t][r] *= __builtin_amdgcn_rcpf(1.f + exp_lean(fminf(-gpre[t][r], 87.f)));   // sigmoid, ~1 ulp
        }
      }
      FC_EACH_BLOCK {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[b][t][r] += tmid[b][t][r];   // resnet.py:52 `inputs + temps`
      }
    }
    // lane (s, g) holds features 8g..8g+7 (tiles 0, 1) and 32+8g..32+8g+7 (tiles 2, 3) of sample s
    FC_EACH_BLOCK {
      const int64_t blk = grp * BPW + b;
      if (blk < a.blocks16) {
        float4* hrow = reinterpret_cast<float4*>(a.h + (blk * 16 + s16) * kHid);
        hrow[2 * g] = float4{h[b][0][0], h[b][0][1], h[b][0][2], h[b][0][3]};
        hrow[2 * g + 1] = float4{h[b][1][0], h[b][1][1], h[b][1][2], h[b][1][3]};
        hrow[8 + 2 * g] = float4{h[b][2][0], h[b][2][1], h[b][2][2], h[b][2][3]};
        hrow[8 + 2 * g + 1] = float4{h[b][3][0], h[b][3][1], h[b][3][2], h[b][3][3]};
      }
    }
  }
#ifdef FC_HID_STAMP
  if (lane == 0 && grp0 < groups) {
    float* dst = a.h + (grp0 * BPW * 16) * kHid;
    dst[0] = (float)(__builtin_amdgcn_s_memtime() - stamp0);
    dst[1] = (float)((groups - grp0 + nwaves - 1) / nwaves * BPW);   // blocks this wave walked
    dst[2] = (float)(stamp0 - stamp_entry);                          // prologue, cycles
    dst[3] = (float)(__builtin_amdgcn_s_memrealtime() - stamp_r0);   // the loop in 10 ns ticks
  }
#endif
#undef FC_EACH_BLOCK
}

template <int NB, int K0S, int kCtx, int kAct, int BPW>
hipError_t launch_hidden(const HiddenArgs& a, int64_t grid, hipStream_t s) {
  using L = HiddenLds<NB, K0S, kCtx>;
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(
      attr, reinterpret_cast<const void*>(&resnet_hidden_kernel<NB, K0S, kCtx, kAct, BPW>), 160 * 1024);
  if (ea != hipSuccess) return ea;
  hipLaunchKernelGGL((resnet_hidden_kernel<NB, K0S, kCtx, kAct, BPW>), dim3((unsigned)grid), dim3(kHidThreads), L::kBytes,
                     s, a);
  return hipGetLastError();
}

template <int kCtx, int kAct>
hipError_t dispatch_hidden(const HiddenArgs& a, int num_blocks, hipStream_t s) {
  const int cus = device_cu_count();
  const bool wide = a.k0 + (kCtx == 1 ? a.C : 0) > 32;
  // One block per wave, two 512-thread workgroups per CU when two weight images fit in LDS (<= 2 blocks at <= 32
  // inputs, no context).  Two blocks per wave (BPW = 2: one workgroup per CU, 206 registers, half the LDS fragment
  // reads) measured slower in the loop -- 7 110 cycles per block and wave at two waves per SIMD against 12 510 at four,
  // i.e. 3 555 against 3 128 per SIMD (tools/probe/hidden_clock.py) -- so it is built for probes only.
#ifdef FC_HID_FORCE_BPW
  constexpr bool kPairs = kCtx == 0 && kAct == 0;
  const bool pairs = kPairs && FC_HID_FORCE_BPW == 2;
#else
  const bool pairs = false;
#endif
  const size_t frags = (size_t)(wide ? 16 : 8) + (size_t)num_blocks * (2 * 16 + (kCtx ? 8 : 0));   // 1 KB each
  int64_t grid = (int64_t)cus * (!pairs && !kCtx && frags * 1024 + 2048 <= 80 * 1024 ? 2 : 1);
  const int64_t need = ((pairs ? (a.blocks16 + 1) / 2 : a.blocks16) + 7) / 8;
  if (grid > need) grid = need;
#ifdef FC_HID_FORCE_BPW
#define FC_HIDDEN_CASE(NB_, K0S_)                                                            \
  do {                                                                                       \
    if constexpr (kPairs) {                                                                  \
      if (pairs) return launch_hidden<NB_, K0S_, kCtx, kAct, 2>(a, grid, s);                 \
    }                                                                                        \
    return launch_hidden<NB_, K0S_, kCtx, kAct, 1>(a, grid, s);                              \
  } while (0)
#else
#define FC_HIDDEN_CASE(NB_, K0S_) return launch_hidden<NB_, K0S_, kCtx, kAct, 1>(a, grid, s)
#endif
  switch (num_blocks * 2 + (wide ? 1 : 0)) {
    case 0: FC_HIDDEN_CASE(0, 1);
    case 1: FC_HIDDEN_CASE(0, 2);
    case 2: FC_HIDDEN_CASE(1, 1);
    case 3: FC_HIDDEN_CASE(1, 2);
    case 4: FC_HIDDEN_CASE(2, 1);
    case 5: FC_HIDDEN_CASE(2, 2);
    case 6: FC_HIDDEN_CASE(3, 1);
    case 7: FC_HIDDEN_CASE(3, 2);
    default: break;
  }
  if constexpr (!kCtx) {
    if (wide) FC_HIDDEN_CASE(4, 2);
    FC_HIDDEN_CASE(4, 1);
  }
#undef FC_HIDDEN_CASE
  return hipErrorInvalidValue;
}

}  // namespace fc

extern "C" int fc_resnet_hidden(const float* x, float* h, const int32_t* id_cols, const float* w0,
                                const float* b0, const float* wb, const float* bb, int64_t n, int32_t d,
                                int32_t in_features, int32_t hidden, int32_t num_blocks, int32_t activation,
                                float activation_param, void* stream) {
  if (n < 0 || d <= 0 || hidden != fc::kHid || num_blocks < 0 || num_blocks > 4) return hipErrorInvalidValue;
  if (activation < FC_ACT_RELU || activation > FC_ACT_SIGMOID) return hipErrorInvalidValue;
  if (in_features <= 0 || in_features > 64 || in_features > d) return hipErrorInvalidValue;
  if (n % 16 != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !h || !id_cols || !w0 || !b0 || (num_blocks > 0 && (!wb || !bb))) return hipErrorInvalidValue;
  if (((uintptr_t)h & 15u) != 0 || ((uintptr_t)wb & 15u) != 0) return hipErrorInvalidValue;
  fc::HiddenArgs a{x, h, id_cols, w0, b0, wb, bb, n / 16, d, in_features, nullptr, nullptr, nullptr, 0,
                   activation, activation_param, nullptr, nullptr, nullptr};
  if (activation == FC_ACT_RELU) return fc::dispatch_hidden<0, 0>(a, num_blocks, static_cast<hipStream_t>(stream));
  return fc::dispatch_hidden<0, 1>(a, num_blocks, static_cast<hipStream_t>(stream));
}

extern "C" int fc_resnet_hidden_packed(const float* x, float* h, const int32_t* id_cols, const void* w_frag,
                                       const float* w_unscale, const float* bias_acc, int64_t n, int32_t d,
                                       int32_t in_features, int32_t hidden, int32_t num_blocks, int32_t activation,
                                       float activation_param, void* stream) {
  if (n < 0 || d <= 0 || hidden != fc::kHid || num_blocks < 0 || num_blocks > 4) return hipErrorInvalidValue;
  if (activation < FC_ACT_RELU || activation > FC_ACT_SIGMOID) return hipErrorInvalidValue;
  if (in_features <= 0 || in_features > 64 || in_features > d) return hipErrorInvalidValue;
  if (n % 16 != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !h || !id_cols || !w_frag || !w_unscale || !bias_acc) return hipErrorInvalidValue;
  if (((uintptr_t)h & 15u) != 0 || ((uintptr_t)w_frag & 15u) != 0) return hipErrorInvalidValue;
  fc::HiddenArgs a{x, h, id_cols, nullptr, nullptr, nullptr, nullptr, n / 16, d, in_features, nullptr, nullptr, nullptr, 0,
                   activation, activation_param, static_cast<const fc::f16x8*>(w_frag), w_unscale, bias_acc};
  if (activation == FC_ACT_RELU) return fc::dispatch_hidden<0, 0>(a, num_blocks, static_cast<hipStream_t>(stream));
  return fc::dispatch_hidden<0, 1>(a, num_blocks, static_cast<hipStream_t>(stream));
}

extern "C" int fc_resnet_hidden_context(const float* x, const float* context, float* h, const int32_t* id_cols,
                                        const float* w0, const float* b0, const float* wb, const float* bb,
                                        const float* wc, const float* bc, int64_t n, int32_t d,
                                        int32_t in_features, int32_t context_features, int32_t hidden,
                                        int32_t num_blocks, int32_t context_mode, int32_t activation,
                                        float activation_param, void* stream) {
  if (activation < FC_ACT_RELU || activation > FC_ACT_SIGMOID) return hipErrorInvalidValue;
  if (context_mode != FC_CONTEXT_GLU && context_mode != FC_CONTEXT_ADDITIVE) return hipErrorInvalidValue;
  // 3 blocks: weight fragments of 4 blocks + 4 gate layers would need 168 KB of LDS
  if (n < 0 || d <= 0 || hidden != fc::kHid || num_blocks < 0 || num_blocks > 3) return hipErrorInvalidValue;
  if (in_features <= 0 || in_features > d || context_features <= 0 || context_features > 32 ||
      in_features + (context_mode == FC_CONTEXT_GLU ? context_features : 0) > 64)
    return hipErrorInvalidValue;
  if (n % 16 != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !context || !h || !id_cols || !w0 || !b0 || (num_blocks > 0 && (!wb || !bb || !wc || !bc)))
    return hipErrorInvalidValue;
  if (context_mode == FC_CONTEXT_ADDITIVE && (!wc || !bc)) return hipErrorInvalidValue;
  if (((uintptr_t)h & 15u) != 0 || ((uintptr_t)wb & 15u) != 0) return hipErrorInvalidValue;
  fc::HiddenArgs a{x, h, id_cols, w0, b0, wb, bb, n / 16, d, in_features, context, wc, bc, context_features,
                   activation, activation_param, nullptr, nullptr, nullptr};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (context_mode == FC_CONTEXT_GLU)
    return activation == FC_ACT_RELU ? fc::dispatch_hidden<1, 0>(a, num_blocks, s) : fc::dispatch_hidden<1, 1>(a, num_blocks, s);
  return activation == FC_ACT_RELU ? fc::dispatch_hidden<2, 0>(a, num_blocks, s) : fc::dispatch_hidden<2, 1>(a, num_blocks, s);
}
