// Autoregressive INVERSE of a MADE-conditioned layer as one kernel: the D passes run on the device (round 4), gfx950.
//
// The reference's sampling direction (flowcon/transforms/autoregressive/autoregressive.py:44-53) runs D conditioner
// passes, each recomputing all D x P parameters although pass d only fixes column d; rounds 1-3 cut the final-layer and
// bijector work to column d but kept the loop on the HOST: D x (hidden stack + ~5 launches), ~85 us per pass.  Here a wave
// keeps 16 rows for ALL passes:
//
//   * the pre-masked MADE (made.py:205-283: initial layer + residual blocks, masks multiplied into the weights once)
//     sits in LDS as ready-made matrix-core A fragments, exactly the image and the register dataflow of fc_resnet_hidden.hip
//     (products transposed, the C layout of one layer is the B operand of the next, split-f16 with f32 accumulation);
//   * pass d: hidden stack on the columns found so far (the others still zero: they only meet zeroed weights) -> the
//     P final-layer rows of dim d (their fragments stream from L2: 2 KB per 16-row tile, the whole final layer would
//     not fit in LDS) -> the 16 samples' parameters through a wave-private LDS strip to ONE lane per sample -> the
//     element-wise inverse of column d (affine: autoregressive.py:97-129; rational-quadratic spline: :529-621 via
//     RQOp<0>::eval_core, every K and both tail modes) -> the new column goes back into the lanes that feed it to the
//     initial layer of pass d + 1;
//   * the per-column log-determinants add up in the evaluating lane; rows and logabsdet leave once, after pass D - 1.
//
// hidden <= 64 (zero-padded), <= 3 residual blocks, ReLU, D <= 64, no context, P <= 48 parameters per dim.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "fc_device.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "fc_rq_op.h"
#include "fc_split.h"
#include "../../include/flowcon_hip.h"

namespace fc {

constexpr int kMiThreads = 512;
// floats per sample row of the parameter strip (16 per parameter tile + 4: rows 16 bytes apart in banks)
constexpr int mi_strip_row(int pt) { return 16 * pt + 4; }
constexpr size_t mi_lds_bytes(int nb, int k0s, int pt, int bpw) {
  return (size_t)(k0s * 8 + 2 * nb * 16) * 64 * 16 + (1 + 2 * nb) * 64 * 4 + 16 * 4 + (size_t)8 * bpw * 16 * mi_strip_row(pt) * 4;
}

struct MadeInvArgs {
  const float* z;           // [N, D] inputs of the inverse
  float* y;                 // [N, D] outputs
  float* lad;               // [N]
  const f16x8* image;       // hidden stack: [layer][ks][t][piece][lane] fragments (accumulator row order)
  const float* image_un;    // [layers]
  const float* image_bias;  // [layers][64], accumulator order
  const f16x8* ffrag;       // final layer: [D][ks 2][PT][piece 2][lane] fragments, rows of tile t = parameters 16 t ..
  const float* fun;         // [D] 2^-S of each dim's rows
  const float* fbias;       // [D][16 PT]
  uint32_t* err;
  const int32_t* need;      // [D] hidden units pass d reads (a prefix of the packed unit order), or null: all 64
  int64_t blocks16;
  int D, P, accumulate;
};

// kind: 0 = affine (P = 2: unconstrained scale, shift), 1 = rational-quadratic spline with a run-time bin count (parameters read
// from the strip as they are needed), 8 / 10 = the same with K = 8 / 10 bins fixed at compile time: the lane copies its 3K -/+ 1
// parameters from its strip and runs the unrolled evaluation of the stand-alone kernels (RQOp<K>::eval_core, two-sided knot walk)
// BPW: 16-row blocks a wave carries together -- every pass is a serial chain (five layers, each row maximum -> split ->
// products -> bias), so a second, independent block fills its waits, and each weight fragment read from LDS serves both
template <int NB, int K0S, int PT, int kKind, int BPW>
__global__ __launch_bounds__(kMiThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void made_inverse_kernel(MadeInvArgs a, RQOp<(kKind > 1 ? kKind : 0)> op) {
  constexpr int kLayers = 1 + 2 * NB, kMiPS = mi_strip_row(PT);
  constexpr int kFrag0 = K0S * 4 * 2, kFragL = 2 * 4 * 2, kFrags = kFrag0 + 2 * NB * kFragL;
  extern __shared__ __attribute__((aligned(16))) unsigned char msm[];
  f16x8* wfrag = reinterpret_cast<f16x8*>(msm);
  float* bias = reinterpret_cast<float*>(msm + (size_t)kFrags * 64 * 16);      // [layer][g][16]
  float* wun = bias + kLayers * 64;                                              // [layer] (padded to 16)
  float* strips = wun + 16;                                                      // [8 waves][BPW][16][16 PT + 4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s16 = lane & 15, g = lane >> 4;
  const int D = a.D;
  for (int f = wave; f < kFrags; f += kMiThreads / 64)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.image + (size_t)f * 64 + lane),
                                     (__attribute__((address_space(3))) void*)(wfrag + f * 64), 16, 0, 0);
  for (int i = tid; i < kLayers * 64; i += kMiThreads) bias[i] = a.image_bias[i];
  if (tid < kLayers) wun[tid] = a.image_un[tid];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* strip = strips + (size_t)wave * BPW * 16 * kMiPS;

  // B operand of one layer from this lane's activations v[t][r], t < NT (k = 32 (t >> 1) + 8 g + 4 (t & 1) + r; the
  // tiles beyond NT count as zero): row maximum over the sample's four lanes, power-of-two scale, two f16 pieces
  auto make_operand = [&](auto nt_c, const f32x4 (&v)[4], f16x8 (&bh)[2], f16x8 (&bl)[2]) {
    constexpr int NT = decltype(nt_c)::value;
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[t][r]));
    m = rows4_allmax(m, lane);
    float sc, un;
    pow2_scale(m, sc, un);
#pragma unroll
    for (int ks = 0; ks < (NT + 1) / 2; ++ks) {
      u32x4 hh, ll;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t ph = 0, pl = 0;
        if (2 * ks + (q >> 1) < NT) split2_pair(v[2 * ks + (q >> 1)][2 * (q & 1)], v[2 * ks + (q >> 1)][2 * (q & 1) + 1], sc, ph, pl);
        hh[q] = ph;
        ll[q] = pl;
      }
      bh[ks] = __builtin_bit_cast(f16x8, hh);
      bl[ks] = __builtin_bit_cast(f16x8, ll);
    }
    return un;
  };
  // a layer's products: NKS k-steps of the operand, the first NTS 16-unit tiles of the outputs
  auto layer = [&](auto nts_c, auto nks_c, int base, const f16x8 (&bh)[BPW][2], const f16x8 (&bl)[BPW][2], f32x4 (&acc)[BPW][4]) {
    constexpr int NTS = decltype(nts_c)::value, NKS = decltype(nks_c)::value;
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
      for (int t = 0; t < NTS; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f16x8* wf = wfrag + base * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      f16x8 wl[NTS], wh[NTS];
#pragma unroll
      for (int t = 0; t < NTS; ++t) {
        wl[t] = wf[((ks * 4 + t) * 2 + 1) * 64];
        wh[t] = wf[((ks * 4 + t) * 2 + 0) * 64];
      }
#pragma unroll
      for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int b = 0; b < BPW; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], bh[b][ks], acc[b][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int b = 0; b < BPW; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], bl[b][ks], acc[b][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NTS; ++t)
#pragma unroll
        for (int b = 0; b < BPW; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], bh[b][ks], acc[b][t], 0, 0, 0);
    }
  };
  auto finish = [&](auto nts_c, int l, float un_act, const f32x4 (&acc)[4], f32x4 (&out)[4]) {
    constexpr int NTS = decltype(nts_c)::value;
    const float c = un_act * wun[l];
    const f32x4* bsrc = reinterpret_cast<const f32x4*>(bias + l * 64 + g * 16);
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
      const f32x4 b = bsrc[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][r] = __builtin_fmaf(acc[t][r], c, b[r]);
    }
  };
  auto relu_tiles = [&](auto nts_c, const f32x4 (&in)[4], f32x4 (&out)[4]) {
    constexpr int NTS = decltype(nts_c)::value;
#pragma unroll
    for (int t = 0; t < NTS; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][r] = fmaxf(in[t][r], 0.f);
  };

  uint32_t err = 0;
  const int64_t nwaves = (int64_t)gridDim.x * (kMiThreads / 64);
  const int64_t groups = (a.blocks16 + BPW - 1) / BPW;
  // block b of group grp; the last group of an odd count repeats its first block (computed twice, stored once)
  auto blk_of = [&](int64_t grp, int b) {
    const int64_t blk = grp * BPW + b;
    return blk < a.blocks16 ? blk : a.blocks16 - 1;
  };
#define FC_EACH_BLOCK _Pragma("unroll") for (int b = 0; b < BPW; ++b)
  for (int64_t grp = (int64_t)blockIdx.x * (kMiThreads / 64) + wave; grp < groups; grp += nwaves) {
    asm volatile("" ::: "memory");      // (the weight fragments are loop-invariant LDS loads: do not hoist them)
    const float* zrow[BPW];
    f32x4 xin[BPW][4];                   // the columns found so far, laid out as the initial layer's B operand
    float lad_sum[BPW], znext[BPW];
    FC_EACH_BLOCK {
      zrow[b] = a.z + (blk_of(grp, b) * 16 + s16) * D;
#pragma unroll
      for (int t = 0; t < 4; ++t) xin[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      lad_sum[b] = 0.f;
      znext[b] = zrow[b][0];
    }
    for (int d = 0; d < D; ++d) {
      asm volatile("" ::: "memory");
      float zval[BPW];
      FC_EACH_BLOCK {
        zval[b] = znext[b];
        znext[b] = zrow[b][d + 1 < D ? d + 1 : d];      // next pass's input, one pass ahead
      }
      // the final-layer fragments of dim d: requested now, used after the hidden stack
      f16x8 fh[2][PT], fl[2][PT];
      {
        const f16x8* fr = a.ffrag + (size_t)d * 2 * PT * 2 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int t = 0; t < PT; ++t) {
            fh[ks][t] = fr[((ks * PT + t) * 2 + 0) * 64];
            fl[ks][t] = fr[((ks * PT + t) * 2 + 1) * 64];
          }
      }
      const float f_un = a.fun[d];
      // pass d reads `units` hidden units (the packed order puts them first) and the d columns found so far
      const int units = a.need ? __builtin_amdgcn_readfirstlane(a.need[d]) : 64;
      // ---- hidden stack (fc_resnet_hidden.hip's dataflow) on the first NTS 16-unit tiles, then the P final-layer rows of
      // dim d (no activation in front: made.py:281); one copy of the code per tile count, the pass picks its own
      f32x4 pacc[BPW][PT];
      float un[BPW];
      FC_EACH_BLOCK {
        un[b] = 0.f;
#pragma unroll
        for (int t = 0; t < PT; ++t) pacc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      auto stack = [&](auto nts_c) {
        constexpr int NTS = decltype(nts_c)::value, NKS = (NTS + 1) / 2;
        const std::integral_constant<int, NKS> nks_c;
        f16x8 bh[BPW][2], bl[BPW][2];
        f32x4 acc[BPW][4], h[BPW][4], tmid[BPW][4];
        FC_EACH_BLOCK un[b] = make_operand(std::integral_constant<int, 2 * K0S>{}, xin[b], bh[b], bl[b]);
        layer(nts_c, std::integral_constant<int, K0S>{}, 0, bh, bl, acc);
        FC_EACH_BLOCK finish(nts_c, 0, un[b], acc[b], h[b]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          FC_EACH_BLOCK {
            f32x4 act[4];
            relu_tiles(nts_c, h[b], act);
            un[b] = make_operand(nts_c, act, bh[b], bl[b]);
          }
          layer(nts_c, nks_c, kFrag0 + (2 * nb) * kFragL, bh, bl, acc);
          FC_EACH_BLOCK {
            f32x4 act[4];
            finish(nts_c, 1 + 2 * nb, un[b], acc[b], tmid[b]);
            relu_tiles(nts_c, tmid[b], act);
            un[b] = make_operand(nts_c, act, bh[b], bl[b]);
          }
          layer(nts_c, nks_c, kFrag0 + (2 * nb + 1) * kFragL, bh, bl, acc);
          FC_EACH_BLOCK {
            finish(nts_c, 2 + 2 * nb, un[b], acc[b], tmid[b]);
#pragma unroll
            for (int t = 0; t < NTS; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) h[b][t][r] += tmid[b][t][r];
          }
        }
        FC_EACH_BLOCK un[b] = make_operand(nts_c, h[b], bh[b], bl[b]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
          for (int t = 0; t < PT; ++t)
            FC_EACH_BLOCK pacc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[ks][t], bh[b][ks], pacc[b][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < PT; ++t)
            FC_EACH_BLOCK pacc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[ks][t], bl[b][ks], pacc[b][t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < PT; ++t)
            FC_EACH_BLOCK pacc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[ks][t], bh[b][ks], pacc[b][t], 0, 0, 0);
        }
      };
      switch ((units + 15) >> 4) {
        case 0: break;                 // no unit feeds this dim (dim 0): its parameters are the final layer's biases
        case 1: stack(std::integral_constant<int, 1>{}); break;
        case 2: stack(std::integral_constant<int, 2>{}); break;
        case 3: stack(std::integral_constant<int, 3>{}); break;
        default: stack(std::integral_constant<int, 4>{}); break;
      }
      // lane (s, g) holds parameters 16 t + 4 g + r of sample s: into the strip, one lane per sample reads them back
      {
        const f32x4* fb = reinterpret_cast<const f32x4*>(a.fbias + (size_t)d * 16 * PT + 4 * g);
        FC_EACH_BLOCK {
          const float c = un[b] * f_un;
#pragma unroll
          for (int t = 0; t < PT; ++t) {
            const f32x4 bv = fb[4 * t];
            *reinterpret_cast<float4*>(strip + (b * 16 + s16) * kMiPS + 16 * t + 4 * g) =
                float4{__builtin_fmaf(pacc[b][t][0], c, bv[0]), __builtin_fmaf(pacc[b][t][1], c, bv[1]),
                       __builtin_fmaf(pacc[b][t][2], c, bv[2]), __builtin_fmaf(pacc[b][t][3], c, bv[3])};
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      // the element-wise inverse of column d: lane 16 b + s evaluates sample s of block b (BPW <= 4 blocks: every lane group
      // has a block to serve at BPW = 4, the first BPW groups otherwise)
      float yv = 0.f;
      if (g < BPW) {
        const float* p = strip + (g * 16 + s16) * kMiPS;
        float zmine = zval[0];
#pragma unroll
        for (int b = 1; b < BPW; ++b) zmine = g == b ? zval[b] : zmine;
        float ladv;
        if constexpr (kKind == 0) {
          // autoregressive.py:124-128: scale = softplus(p[0]) + 1e-3, shift = p[1]; inverse (x - shift) / scale
          const float sc = softplus_lean(p[0], 1.f) + 1e-3f;
          yv = div_lean(zmine - p[1], sc);
          ladv = -log_lean(sc);
        } else {
          static_assert(kKind == 1 || PT == 2, "3K -/+ 1 parameters in two 16-row tiles");
          op.template eval_core<false>(p, zmine, yv, ladv, err);     // K static (8 / 10): the unrolled two-sided walk on the strip
        }
#pragma unroll
        for (int b = 0; b < BPW; ++b) lad_sum[b] += g == b ? ladv : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
      // the new column to the lanes of its sample (lane 16 b + s16 holds it), then into the operand slot of feature d
      const int slot = d & 31;                         // k within its k-step: 8 g' + j
      const bool mine = (slot >> 3) == g;
      const int reg = 8 * (d >> 5) + (slot & 7);       // t = 2 (d >> 5) + ((slot & 7) >> 2), r = slot & 3  ->  4 t + r
      FC_EACH_BLOCK {
        const float yb = __shfl(yv, 16 * b + s16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) xin[b][t][r] = (mine && reg == 4 * t + r) ? yb : xin[b][t][r];
      }
    }
    // rows out: lane (s, g) holds columns 32 ks + 8 g + j
    FC_EACH_BLOCK {
      if (grp * BPW + b >= a.blocks16) continue;      // (the repeated block of an odd tail is not stored twice)
      const int64_t row = (grp * BPW + b) * 16 + s16;
      float* yrow = a.y + row * D;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r;
          if ((t >> 1) < K0S && c < D) yrow[c] = xin[b][t][r];
        }
      // the block's log-determinants sit in lane group b
      const float l = __shfl(lad_sum[b], 16 * b + s16);
      if (g == 0) a.lad[row] = a.accumulate ? a.lad[row] + l : l;
    }
  }
#undef FC_EACH_BLOCK
  if (err && a.err) atomicOr(a.err, err);
}

template <int NB, int K0S, int PT, int kKind, int BPW>
hipError_t launch_made_inverse_bpw(const MadeInvArgs& a, const RQOp<0>& op, hipStream_t s) {
  constexpr size_t lds = mi_lds_bytes(NB, K0S, PT, BPW);
  static_assert(lds <= 160 * 1024, "weight image exceeds the CU's LDS");
  static PerDeviceOnce attr;
  const hipError_t ea =
      ensure_max_dynamic_lds(attr, reinterpret_cast<const void*>(&made_inverse_kernel<NB, K0S, PT, kKind, BPW>), 160 * 1024);
  if (ea != hipSuccess) return ea;
  int64_t grid = device_cu_count();
  const int64_t need = ((a.blocks16 + BPW - 1) / BPW + 7) / 8;
  if (grid > need) grid = need;
  RQOp<(kKind > 1 ? kKind : 0)> opk;       // same fields; the static-K form is the type the unrolled evaluation is written on
  opk.q = op.q;
  opk.inv_beta = op.inv_beta;
  opk.inv_div = op.inv_div;
  hipLaunchKernelGGL((made_inverse_kernel<NB, K0S, PT, kKind, BPW>), dim3((unsigned)grid), dim3(kMiThreads), lds, s, a, opk);
  return hipGetLastError();
}

template <int NB, int K0S, int PT, int kKind>
hipError_t launch_made_inverse(const MadeInvArgs& a, const RQOp<0>& op, hipStream_t s) {
  // two blocks per wave once every wave of the chip has a pair to carry (small batches: one block per wave fills more CUs)
  if constexpr (mi_lds_bytes(NB, K0S, PT, 2) <= 160 * 1024) {
    if (a.blocks16 >= 2 * 8 * (int64_t)device_cu_count()) return launch_made_inverse_bpw<NB, K0S, PT, kKind, 2>(a, op, s);
  }
  return launch_made_inverse_bpw<NB, K0S, PT, kKind, 1>(a, op, s);
}

template <int NB, int K0S>
hipError_t dispatch_made_inverse_pt(const MadeInvArgs& a, const RQOp<0>& op, int kind, hipStream_t s) {
  const int pt = (a.P + 15) / 16;
  if (kind == 0) return launch_made_inverse<NB, K0S, 1, 0>(a, op, s);
  if (op.q.K == 8) return launch_made_inverse<NB, K0S, 2, 8>(a, op, s);        // 23 / 25 parameters per dim
  if (op.q.K == 10) return launch_made_inverse<NB, K0S, 2, 10>(a, op, s);      // 29 / 31 (the reference's default bin count)
  switch (pt) {
    case 1: return launch_made_inverse<NB, K0S, 1, 1>(a, op, s);
    case 2: return launch_made_inverse<NB, K0S, 2, 1>(a, op, s);
    case 3: return launch_made_inverse<NB, K0S, 3, 1>(a, op, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace fc

extern "C" int fc_made_inverse(const float* z, float* y, float* logabsdet, const void* hidden_frag,
                               const float* hidden_unscale, const float* hidden_bias, const void* final_frag,
                               const float* final_unscale, const float* final_bias, const int32_t* units_needed,
                               uint32_t* err_flag, int64_t n,
                               int32_t d, int32_t num_blocks, int32_t params_per_dim, int32_t kind,
                               const fc_rq_config* cfg, void* stream) {
  if (n < 0 || d < 1 || d > 64 || num_blocks < 0 || num_blocks > 3 || kind < 0 || kind > 1) return hipErrorInvalidValue;
  if (params_per_dim < 1 || params_per_dim > 48 || (kind == 0 && params_per_dim != 2) || (kind == 1 && !cfg)) return hipErrorInvalidValue;
  if (n % 16 != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!z || !y || !logabsdet || !hidden_frag || !hidden_unscale || !hidden_bias || !final_frag || !final_unscale || !final_bias)
    return hipErrorInvalidValue;
  if ((((uintptr_t)hidden_frag | (uintptr_t)final_frag | (uintptr_t)final_bias) & 15u) != 0) return hipErrorInvalidValue;
  fc::RQOp<0> op{};
  if (kind == 1) {
    fc::RQParams& q = op.q;
    q.K = cfg->num_bins; q.tails = cfg->tails ? 1 : 0; q.inverse = 1;
    if (q.K < 1 || q.K > 16 || params_per_dim != (q.tails ? 3 * q.K - 1 : 3 * q.K + 1)) return hipErrorInvalidValue;
    q.left = cfg->left; q.right = cfg->right; q.bottom = cfg->bottom; q.top = cfg->top;
    q.min_w = (float)cfg->min_bin_width; q.min_h = (float)cfg->min_bin_height; q.min_d = (float)cfg->min_derivative;
    q.cw = (float)(1.0 - cfg->min_bin_width * q.K);
    q.ch = (float)(1.0 - cfg->min_bin_height * q.K);
    fc::rq_finish_params(q);
    q.wh_div = cfg->wh_divisor > 0.f ? cfg->wh_divisor : 1.f;
    q.beta = cfg->softplus_beta;
    q.tail_const = cfg->tail_constant;
    op.inv_div = 1.f / q.wh_div;
    op.inv_beta = 1.f / q.beta;
  }
  fc::MadeInvArgs a{z, y, logabsdet, static_cast<const fc::f16x8*>(hidden_frag), hidden_unscale, hidden_bias,
                    static_cast<const fc::f16x8*>(final_frag), final_unscale, final_bias, err_flag, units_needed, n / 16, d, params_per_dim,
                    (cfg && (cfg->flags & FC_RQ_ACCUMULATE_LOGABSDET)) ? 1 : 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool wide = d > 32;
#define FC_MI(NBV)                                                                                    \
  case NBV:                                                                                           \
    return wide ? fc::dispatch_made_inverse_pt<NBV, 2>(a, op, kind, s) : fc::dispatch_made_inverse_pt<NBV, 1>(a, op, kind, s);
  switch (num_blocks) {
    FC_MI(0) FC_MI(1) FC_MI(2) FC_MI(3)
    default: return hipErrorInvalidValue;
  }
#undef FC_MI
}
