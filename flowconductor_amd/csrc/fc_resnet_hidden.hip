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
// Weights: both f16 pieces of all layers as ready-made A fragments in LDS (72 KB at <= 32 inputs): copied from an
// image prepared once per weight version by fc_pack_fragments (fc_resnet_hidden_packed, fc_affine_coupling_resnet), or
// built by every workgroup from the row-major f32 weights (fc_resnet_hidden, fc_resnet_hidden_context).
// kTail (fc_affine_coupling_resnet): the final Linear of an affine coupling layer and the bijector itself run as a tail of
// the stack -- one kernel per coupling layer.
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
  // fc_affine_coupling_resnet (kTail): the final Linear and the affine bijector run here too -- y [N, D] is the layer's
  // output and h is not written.  The image then carries one more 64 x 64 layer: rows 0..31 = the final Linear's shift
  // rows of dims 0..31, rows 32..63 = its scale rows (zero rows beyond d_t).
  float* y;                 // [N, D]
  const int32_t* tr_cols;   // [d_t] transformed columns
  float* lad;               // [N]
  int d_t, affine_act, inverse, accumulate;
};

// feature held by accumulator tile t, register r of a lane in group g
__host__ __device__ constexpr int hid_feat(int t, int g, int r) { return 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r; }

// LDS: [layer][k-step][tile][piece][lane] f16x8 fragments, then bias [layer][g][16], unscale [layer], ids
template <int NB, int K0S, int kCtx, int kTail = 0>
struct HiddenLds {
  static constexpr int kMain = 1 + 2 * NB + kTail;        // initial layer + two per block (+ the final Linear)
  static constexpr int kCtxLayers = kCtx == 1 ? NB : kCtx == 2 ? NB + 1 : 0;   // context products: per block (+ initial)
  static constexpr int kLayers = kMain + kCtxLayers;
  static constexpr int kFrag0 = K0S * 4 * 2;              // fragments of the initial layer
  static constexpr int kFragL = 2 * 4 * 2;                // fragments of a 64 x 64 layer
  static constexpr int kFragG = 1 * 4 * 2;                // fragments of a 64 x C gate layer (C <= 32)
  static constexpr int kFragsMain = kFrag0 + (2 * NB + kTail) * kFragL;
  static constexpr int kFrags = kFragsMain + kCtxLayers * kFragG;
  static constexpr size_t kBytes = (size_t)kFrags * 64 * 16 + kLayers * 64 * 4 + 16 * 4 + 32 * K0S * 4 + 16 * 8 * 4;
  static_assert(kLayers <= 16, "wun holds 16 entries");
  static_assert(kBytes <= 160 * 1024, "weight fragments exceed the CU's LDS");
};

// kAct: 0 = ReLU (the north-star conditioner; nothing but a v_max), 1 = the activation named by a.act
// BPW: 16-sample blocks a wave pushes through the layers together.  Every layer's weight fragments (16 KB per wave) come
// from LDS once per group of BPW blocks; at one block per wave the 16 waves of a CU ask the LDS pipe for as many cycles
// as their matrix and vector instructions take to issue, and the reads sit right before the MFMAs that need them.
// kTail: the affine coupling layer's final Linear + bijector as a tail of the stack (one kernel per coupling layer; the
// image is 16 KB larger and every wave stages its 16 rows in LDS: one workgroup per CU).
// XVT: 16-byte pieces of a 16-row chunk of x per lane the tail kernels keep in registers (D <= 16 XVT)
template <int NB, int K0S, int kCtx, int kAct, int BPW, int kTail = 0, int XVT = 8>
__global__ __launch_bounds__(512, (kCtx || BPW > 1 || kTail) ? 2 : 4) void resnet_hidden_kernel(HiddenArgs a) {
  static_assert(!kTail || kCtx == 0, "the coupling tail: no context");
  using L = HiddenLds<NB, K0S, kCtx, kTail>;
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
  } else if constexpr (!kTail) {      // (the coupling-tail kernels only exist with a ready-made image)
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
  // Alternating masks (identity columns 2 k + parity of a [N, 2 k0] input, k0 a whole number of k-steps, rows 16-byte
  // aligned): this lane's 8 inputs of a k-step sit in 16 consecutive floats -- four 16-byte loads instead of eight 4-byte
  // gathers at a stride (prologue 15 100 -> 10 300 cycles, loop 12 000 -> 11 600 cycles per block and wave).  -1: any other
  // mask, the gather.
  int alt = -1;
  if (kCtx == 0 && D == 2 * k0 && k0 == 32 * K0S && ((uintptr_t)a.x & 15u) == 0 && (ids[0] == 0 || ids[0] == 1)) {
    alt = ids[0];
    for (int i = 1; i < k0; ++i)
      if (ids[i] != 2 * i + ids[0]) alt = -1;
  }
  auto gather = [&](int64_t blk, f32x4 (&xv)[4]) {
    const float* xrow = a.x + (blk * 16 + s16) * D;
    const float* crow_ = kCtx ? a.ctx + (blk * 16 + s16) * C : nullptr;
    if (alt >= 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks < K0S) {
          const float4* q = reinterpret_cast<const float4*>(xrow + 64 * ks + 16 * g);
          const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
          xv[2 * ks] = alt ? f32x4{q0.y, q0.w, q1.y, q1.w} : f32x4{q0.x, q0.z, q1.x, q1.z};
          xv[2 * ks + 1] = alt ? f32x4{q2.y, q2.w, q3.y, q3.w} : f32x4{q2.x, q2.z, q3.x, q3.z};
        } else {
          xv[2 * ks] = xv[2 * ks + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      return;
    }
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
  if constexpr (kTail) {
    // ---- one affine coupling layer per launch -------------------------------------------------------------------------
    // The wave's BPW x 16 rows (contiguous 64 D-byte chunks of x) pass through wave-private LDS tiles: coalesced 16-byte
    // loads in, the conditioner's inputs and the transformed columns picked from the tile, the results written back into
    // it, coalesced 16-byte stores out -- the identity columns ride along.  (Per-lane 4-byte gathers / scatters at a
    // column stride cost more address-path time than the whole stack.)  Row stride D | 1: conflict-free column reads.
    // BPW = 2 (round 3): the kernel lives at two waves per SIMD anyway (its weight image leaves room for one workgroup
    // per CU); two blocks per wave give every serial layer chain a second, independent one to overlap with, and every
    // weight fragment read from LDS serves both.
    const int TS = D | 1, wrap = TS - D;
    float* tile = reinterpret_cast<float*>(hsmem + ((L::kBytes + 15) & ~size_t(15))) + (size_t)wave * BPW * 16 * TS;
    const int chunk4 = 4 * D;                        // float4 per 16-row chunk; this lane owns pieces lane + 64 k
    constexpr int XV = XVT;                          // D <= 16 XV (D <= 128)
    int toff[XV], tcol[XV];
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int e = 4 * (lane + 64 * k), r = e / D;
      tcol[k] = e - r * D;
      toff[k] = r * TS + tcol[k];
    }
    auto fetch_rows = [&](int64_t blk, float4 (&raw)[XV]) __attribute__((always_inline)) {
      const float4* src = reinterpret_cast<const float4*>(a.x + blk * 16 * D);
#pragma unroll
      for (int k = 0; k < XV; ++k)
        if (lane + 64 * k < chunk4) raw[k] = src[lane + 64 * k];
    };
    const int64_t groups = (a.blocks16 + BPW - 1) / BPW;
    const int64_t nw = (int64_t)gridDim.x * (kHidThreads / 64);
    const int64_t first = (int64_t)blockIdx.x * (kHidThreads / 64) + wave;
    // block b of group grp; the last group of an odd count repeats its first block (computed twice, stored once)
    auto blk_of = [&](int64_t grp, int b) {
      const int64_t blk = grp * BPW + b;
      return blk < a.blocks16 ? blk : a.blocks16 - 1;
    };
    float4 raw[BPW][XV];
    if (first < groups) {
#pragma unroll
      for (int b = 0; b < BPW; ++b) fetch_rows(blk_of(first, b), raw[b]);
    }
    int tc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int dim = 8 * g + e;
      tc[e] = a.tr_cols[dim < a.d_t ? dim : 0];
    }
    for (int64_t grp = first; grp < groups; grp += nw) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int b = 0; b < BPW; ++b)
#pragma unroll
        for (int k = 0; k < XV; ++k)
          if (lane + 64 * k < chunk4) {
            const float v[4] = {raw[b][k].x, raw[b][k].y, raw[b][k].z, raw[b][k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[b * 16 * TS + toff[k] + j + (tcol[k] + j >= D ? wrap : 0)] = v[j];
          }
      __builtin_amdgcn_wave_barrier();
      if (grp + nw < groups) {                                      // the next group's rows, one iteration ahead
#pragma unroll
        for (int b = 0; b < BPW; ++b) fetch_rows(blk_of(grp + nw, b), raw[b]);
      }
      f32x4 xin[BPW][4];
      float xt[BPW][8];
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        const float* trow = tile + (b * 16 + s16) * TS;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ks = t >> 1, j = 4 * (t & 1) + r;
            float v = 0.f;
            if (ks < K0S) {
              const int c = mycol[ks < K0S ? ks : 0][j];
              v = c >= 0 ? trow[c] : 0.f;
            }
            xin[b][t][r] = v;
          }
#pragma unroll
        for (int e = 0; e < 8; ++e) xt[b][e] = trow[tc[e]];
      }
      f16x8 bh[BPW][2], bl[BPW][2];
      f32x4 acc[BPW][4], h[BPW][4], tmid[BPW][4];
      float un[BPW];
#pragma unroll
      for (int b = 0; b < BPW; ++b) un[b] = make_operand(xin[b], bh[b], bl[b]);
      layer(0, K0S, bh, bl, acc);
#pragma unroll
      for (int b = 0; b < BPW; ++b) finish(0, un[b], acc[b], h[b]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
          f32x4 act[4];
          activate16(h[b], act);
          un[b] = make_operand(act, bh[b], bl[b]);
        }
        layer(L::kFrag0 + (2 * nb) * L::kFragL, 2, bh, bl, acc);
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
          finish(1 + 2 * nb, un[b], acc[b], tmid[b]);
          f32x4 act[4];
          activate16(tmid[b], act);
          un[b] = make_operand(act, bh[b], bl[b]);
        }
        layer(L::kFrag0 + (2 * nb + 1) * L::kFragL, 2, bh, bl, acc);
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
          finish(2 + 2 * nb, un[b], acc[b], tmid[b]);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[b][t][r] += tmid[b][t][r];
        }
      }
      // final Linear: one more 64 x 64 product on the residual stream (no activation in front: resnet.py:99)
#pragma unroll
      for (int b = 0; b < BPW; ++b) un[b] = make_operand(h[b], bh[b], bl[b]);
      layer(L::kFrag0 + 2 * NB * L::kFragL, 2, bh, bl, acc);
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        f32x4 prm[4];
        finish(1 + 2 * NB, un[b], acc[b], prm);
        // lane (s, g): dims 8g + e, e = 4t + r (t < 2): shift = prm[t][r], raw scale = prm[2 + t][r]
        float ladsum = 0.f;
        float* wrow = tile + (b * 16 + s16) * TS;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = 4 * t + r;
            if (8 * g + e < a.d_t) {
              const float xv = xt[b][e], shift = prm[t][r], u = prm[2 + t][r];
              float sc = 1.f, ls = 0.f;
              // (the lean primitives of fc_math.h: hardware exp2 / log2 / rcp + one correction step, <= 1-2 ulp)
              if (a.affine_act == FC_AFFINE_SIGMOID_PLUS2) {
                const float v = u + 2.f;
                const float ex = exp_lean(-fabsf(v));
                const float rcp = div_lean(1.f, 1.f + ex);
                sc = (v >= 0.f ? rcp : ex * rcp) + 1e-3f;
                ls = log_lean(sc);
              } else if (a.affine_act == FC_AFFINE_SOFTPLUS_CLAMP3) {
                const float v = softplus_lean(u, 1.f) + 1e-3f;
                sc = v < 0.f ? 0.f : (v > 3.f ? 3.f : v);      // torch.clamp: a NaN stays a NaN
                ls = log_lean(sc);
              } else if (a.affine_act == FC_AFFINE_MAF_SOFTPLUS) {   // autoregressive.py:97-129 (rows re-ordered by the packer)
                sc = softplus_lean(u, 1.f) + 1e-3f;
                ls = log_lean(sc);
              }
              wrow[tc[e]] = a.inverse ? div_lean(xv - shift, sc) : xv * sc + shift;
              ladsum += a.inverse ? -ls : ls;
            }
          }
        const float l = rows4_allsum(ladsum, lane);
        const bool real = grp * BPW + b < a.blocks16;          // (the repeated block of an odd tail is not stored twice)
        const int64_t row = blk_of(grp, b) * 16 + s16;
        if (g == 0 && real) a.lad[row] = a.accumulate ? a.lad[row] + l : l;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        if (grp * BPW + b >= a.blocks16) continue;
        float4* dst = reinterpret_cast<float4*>(a.y + blk_of(grp, b) * 16 * D);
#pragma unroll
        for (int k = 0; k < XV; ++k)
          if (lane + 64 * k < chunk4) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = tile[b * 16 * TS + toff[k] + j + (tcol[k] + j >= D ? wrap : 0)];
            dst[lane + 64 * k] = float4{v[0], v[1], v[2], v[3]};
          }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }

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

template <int NB, int K0S, int kCtx, int kAct, int BPW, int kTail = 0, int XVT = 8>
hipError_t launch_hidden(const HiddenArgs& a, int64_t grid, hipStream_t s) {
  using L = HiddenLds<NB, K0S, kCtx, kTail>;
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(
      attr, reinterpret_cast<const void*>(&resnet_hidden_kernel<NB, K0S, kCtx, kAct, BPW, kTail, XVT>), 160 * 1024);
  if (ea != hipSuccess) return ea;
  if (kTail && a.D > 16 * XVT) return hipErrorInvalidValue;
  // (tail kernels: + a [16][D | 1] float tile per wave)
  const size_t lds = kTail ? ((L::kBytes + 15) & ~size_t(15)) + (size_t)(kHidThreads / 64) * BPW * 16 * (a.D | 1) * 4 : L::kBytes;
  if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL((resnet_hidden_kernel<NB, K0S, kCtx, kAct, BPW, kTail, XVT>), dim3((unsigned)grid), dim3(kHidThreads), lds,
                     s, a);
  return hipGetLastError();
}

// the affine coupling layer in one kernel: ReLU conditioner, <= 3 blocks (the image of 4 blocks + the final layer
// would not fit in LDS), one workgroup per CU
template <int NB, int K0S>
inline hipError_t launch_coupling_tail(const HiddenArgs& a, hipStream_t s) {
  // two blocks per wave when there is work for them (>= two rounds of single blocks over the chip) and the second row
  // tile fits next to the weight image; the last group of an odd block count computes its block twice
  using L = HiddenLds<NB, K0S, 0, 1>;
  const int64_t cus = device_cu_count();
  const size_t lds2 = ((L::kBytes + 15) & ~size_t(15)) + (size_t)(kHidThreads / 64) * 2 * 16 * (a.D | 1) * 4;
  const bool pairs = a.blocks16 >= 2 * cus * 8 && lds2 <= 160 * 1024 && a.D <= 64;   // (wider rows: the second block's pieces spill)
  const int64_t units = pairs ? (a.blocks16 + 1) / 2 : a.blocks16;
  int64_t grid = cus;
  const int64_t need = (units + 7) / 8;
  if (grid > need) grid = need;
  if (!pairs) return launch_hidden<NB, K0S, 0, 0, 1, 1>(a, grid, s);
  // (the two-block kernels carry 2 x XVT row pieces per lane across the layers: instantiated per input width)
  if (a.D <= 32) return launch_hidden<NB, K0S, 0, 0, 2, 1, 2>(a, grid, s);
  return launch_hidden<NB, K0S, 0, 0, 2, 1, 4>(a, grid, s);
}

// the affine coupling layer in one kernel: ReLU conditioner, <= 3 blocks (the image of 4 blocks + the final layer
// would not fit in LDS), one workgroup per CU
inline hipError_t dispatch_coupling_tail(const HiddenArgs& a, int num_blocks, hipStream_t s) {
  const bool wide = a.k0 > 32;
  switch (num_blocks * 2 + (wide ? 1 : 0)) {
    case 0: return launch_coupling_tail<0, 1>(a, s);
    case 1: return launch_coupling_tail<0, 2>(a, s);
    case 2: return launch_coupling_tail<1, 1>(a, s);
    case 3: return launch_coupling_tail<1, 2>(a, s);
    case 4: return launch_coupling_tail<2, 1>(a, s);
    case 5: return launch_coupling_tail<2, 2>(a, s);
    case 6: return launch_coupling_tail<3, 1>(a, s);
    case 7: return launch_coupling_tail<3, 2>(a, s);
    default: return hipErrorInvalidValue;
  }
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
                   activation, activation_param, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
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
                   activation, activation_param, static_cast<const fc::f16x8*>(w_frag), w_unscale, bias_acc,
                   nullptr, nullptr, nullptr, 0, 0, 0, 0};
  if (activation == FC_ACT_RELU) return fc::dispatch_hidden<0, 0>(a, num_blocks, static_cast<hipStream_t>(stream));
  return fc::dispatch_hidden<0, 1>(a, num_blocks, static_cast<hipStream_t>(stream));
}

extern "C" int fc_affine_coupling_resnet(const float* x, float* y, const int32_t* id_cols, const int32_t* tr_cols,
                                         const void* w_frag, const float* w_unscale, const float* bias_acc,
                                         float* logabsdet, int64_t n, int32_t d, int32_t in_features, int32_t d_t,
                                         int32_t hidden, int32_t num_blocks, int32_t scale_activation, int32_t inverse,
                                         int32_t accumulate, void* stream) {
  if (n < 0 || d <= 0 || hidden != fc::kHid || num_blocks < 0 || num_blocks > 3) return hipErrorInvalidValue;
  // (a coupling layer has in_features + d_t <= d; a masked-autoregressive layer in its density direction reads and transforms
  //  ALL columns: the kernel picks every input from the wave's row tile before it writes any result into it)
  if (in_features <= 0 || in_features > 64 || d_t <= 0 || d_t > 32 || in_features > d || d_t > d || d > 128) return hipErrorInvalidValue;
  if (scale_activation != FC_AFFINE_SIGMOID_PLUS2 && scale_activation != FC_AFFINE_SOFTPLUS_CLAMP3 &&
      scale_activation != FC_AFFINE_ADDITIVE && scale_activation != FC_AFFINE_MAF_SOFTPLUS)
    return hipErrorInvalidValue;
  if (n % 16 != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || x == y || !id_cols || !tr_cols || !w_frag || !w_unscale || !bias_acc || !logabsdet) return hipErrorInvalidValue;
  if (((uintptr_t)w_frag & 15u) != 0) return hipErrorInvalidValue;
  fc::HiddenArgs a{x, nullptr, id_cols, nullptr, nullptr, nullptr, nullptr, n / 16, d, in_features, nullptr, nullptr, nullptr, 0,
                   FC_ACT_RELU, 0.f, static_cast<const fc::f16x8*>(w_frag), w_unscale, bias_acc,
                   y, tr_cols, logabsdet, d_t, scale_activation, inverse ? 1 : 0, accumulate ? 1 : 0};
  return fc::dispatch_coupling_tail(a, num_blocks, static_cast<hipStream_t>(stream));
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
                   activation, activation_param, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (context_mode == FC_CONTEXT_GLU)
    return activation == FC_ACT_RELU ? fc::dispatch_hidden<1, 0>(a, num_blocks, s) : fc::dispatch_hidden<1, 1>(a, num_blocks, s);
  return activation == FC_ACT_RELU ? fc::dispatch_hidden<2, 0>(a, num_blocks, s) : fc::dispatch_hidden<2, 1>(a, num_blocks, s);
}
