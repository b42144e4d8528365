// Final conditioner layer fused with the RQ-spline coupling bijector, gfx950.
//
//   params[n, :] = h[n, :] @ W^T + b          W: [d_t*(3K-1), H]    (nn.Linear of ResidualNet,
//                                                                     flowcon/nn/nets/resnet.py:91,99)
//   y, logabsdet = rq_spline_coupling(x, params)                     (coupling.py:279-293, 549-582)
//
// Unfused, the [N, 736] f32 parameter tensor is written by the GEMM and read back by the spline
// kernel: 5.9 KB of HBM traffic per sample and layer for cfg 3, 85 % of everything the layer moves.
// Here it only ever exists in LDS.  Specialised for the north-star layer shape: H = 64, d_t = 32,
// K = 8, linear tails (P = 23).
//
// Structure (one 512-thread workgroup per CU, persistent over 32-row tiles):
//   * waves 0-3 ("producers"): exact-f32 matrix cores, v_mfma_f32_32x32x2_f32.  The whole weight matrix
//     lives in their registers as MFMA B fragments for the entire kernel (4 waves x 6 column tiles x
//     32 k-steps = 192 VGPRs each), so weights cost no memory traffic per tile.  A fragments come from
//     the h tile in LDS.  Result tiles go to an LDS parameter buffer.
//   * waves 4-7 ("consumers"): the spline arithmetic (same eval_core as fc_rq_spline) on the VALU,
//     out of that LDS buffer.
//   * the 32 transformed dims are processed as two halves of 16 with a double-buffered LDS parameter
//     buffer: while the consumers evaluate half s, the producers fill half s+1.  MFMA and VALU are
//     separate pipes, one producer and one consumer wave share each SIMD, so the two overlap; one
//     __syncthreads per half-step hands the buffers over.
// Columns are padded from P = 23 to 24 per dim (768 = 24 MFMA tiles of 32, 12 per half); the pad
// column has zero weight.  Per-dim blocks are skewed by 2 floats every 4 dims in LDS so that the
// consumers' stride-24 reads hit 16 distinct banks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fc_tile.h"
#include "fc_math.h"
#include "fc_rq_op.h"
#include "fc_rq_fused.h"
#include "../../include/flowcon_hip.h"

namespace fc {

// Consumer work of one half-step: the 2 elements of this thread, as two independent straight-line chains.
// A separate (noinline) function on purpose: it gets its own register allocation, so the producers' 192
// resident weight registers and this code's ~100 temporaries never compete in hipcc's allocator.
template <bool kInv>
__device__ __attribute__((noinline)) uint32_t fused_consume(RQOp<kK> op, const float* __restrict__ phalf,
                                                            float* __restrict__ xtile, const int* __restrict__ cols_half,
                                                            float* __restrict__ lbuf, int D, int ctid, int first_half,
                                                            int skip) {
  uint32_t err = 0;
  float xin[2], yv[2], lad[2];
  float* xr[2];
  const float* pp[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = ctid + it * 256;
    const int row = e >> 4, jj = e & 15;
    xr[it] = xtile + row * D + cols_half[jj];
    pp[it] = phalf + row * kPRow + skewed(jj * kPP);
    xin[it] = *xr[it];
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (skip) { yv[it] = xin[it] + pp[it][0] * 0.f; lad[it] = 0.f; }
    else op.template eval_tails_straight<kInv>(pp[it], xin[it], yv[it], lad[it], err);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = ctid + it * 256;
    const int row = e >> 4, jj = e & 15;
    *xr[it] = yv[it];
    float l = lad[it];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) l += __shfl_xor(l, o, 16);
    if (jj == 0) {
      if (first_half) lbuf[row] = l; else lbuf[row] += l;
    }
  }
  return err;
}

__global__ __launch_bounds__(512) void rq_fused_linear_kernel(RQOp<kK> op, FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* pbuf = smem;                              // [2][kR][kPRow]
  float* hbuf = pbuf + kPBuf;                      // [2][kR][kHRow]
  float* xbuf = hbuf + kHBuf;                      // [2][kR][D]
  float* lbuf = xbuf + 2 * kR * a.D;               // [kR] logabsdet partials
  int* cs = reinterpret_cast<int*>(lbuf + kR);     // [kDt]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int D = a.D;
  const int64_t stride = gridDim.x;
  const int64_t tile0 = blockIdx.x;
  if (tile0 >= a.tiles) return;  // whole workgroup leaves together

  // The two roles are two separate loops with matching barrier counts (s_barrier only counts arrivals),
  // so that each role gets its own register allocation: the producers' 192 resident weight registers
  // are not live anywhere in the consumers' code.
  if (wave < 4) {
    // ================= producers: exact-f32 MFMA, weights resident in registers =================
    // Weights: all six of this wave's column tiles (2 halves x 3) keep their 32 B-fragments in registers
    // for the whole kernel: 192 VGPRs, zero weight traffic per tile.  (Only possible since the accumulators
    // start from an inline zero: bias-filled start values were hoisted by hipcc, 96 more VGPRs, and spilled.)
    float wreg[2][kTilesPerWave][kSteps];
    float breg[2][kTilesPerWave];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int t = 0; t < kTilesPerWave; ++t) {
        // B fragment of v_mfma_f32_32x32x2_f32: lane l holds Wpad[col = tile*32 + (l & 31)][k = 2s + (l >> 5)]
        const float* wrow = a.wpad + (int64_t)(hf * kHalfCols + (wave * kTilesPerWave + t) * 32 + (lane & 31)) * kH +
                            (lane >> 5);
#pragma unroll
        for (int s = 0; s < kSteps; ++s) wreg[hf][t][s] = wrow[2 * s];
        breg[hf][t] = a.bias[hf * kHalfCols + (wave * kTilesPerWave + t) * 32 + (lane & 31)];
      }
    }

    // h tile: 32 x 64 floats = 512 float4 = two per producer thread
    float4 hv0, hv1;
    auto fetch_h = [&](int64_t t) {
      const float4* hg = reinterpret_cast<const float4*>(a.h + t * kR * kH);
      hv0 = hg[tid];
      hv1 = hg[tid + 256];
    };
    auto park_h = [&](int buf) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = tid + k * 256;
        const float4 v = k == 0 ? hv0 : hv1;
        float* dst = hbuf + (buf * kR + (i * 4) / kH) * kHRow + (i * 4) % kH;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      }
    };
    auto mma_tile = [&](const float* hrow, const float (&w)[kSteps], float b0, int hf, int t) {
      // accumulate from an inline-constant zero and add the bias on the way out: a bias-filled 16-register
      // start value is loop invariant, hipcc hoists all six of them (96 VGPRs) and then spills
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < kSteps; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hrow[2 * s], w[s], acc, 0, 0, 0);
      const int c = (wave * kTilesPerWave + t) * 32 + (lane & 31);
      float* dst = pbuf + hf * kR * kPRow + skewed(c);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[row * kPRow] = acc[r] + b0;
      }
    };
    auto produce = [&](int hb, int hf) {
      if (a.debug & 1) return;
      const float* hrow = hbuf + (hb * kR + (lane & 31)) * kHRow + (lane >> 5);
#pragma unroll
      for (int t = 0; t < kTilesPerWave; ++t) mma_tile(hrow, wreg[hf][t], breg[hf][t], hf, t);
    };

    fetch_h(tile0);
    park_h(0);
    __syncthreads();   // #0: first h / x tiles are in LDS
    produce(0, 0);
    __syncthreads();   // #1: half 0 of the first tile is in pbuf[0]
    int tb = 0;
    for (int64_t tile = tile0; tile < a.tiles; tile += stride) {
      const bool has_next = tile + stride < a.tiles;
      if (has_next) fetch_h(tile + stride);
      produce(tb, 1);
      if (has_next) park_h(tb ^ 1);
      __syncthreads();  // A
      if (has_next) produce(tb ^ 1, 0);
      __syncthreads();  // B
      __syncthreads();  // C (consumers store the finished tile)
      tb ^= 1;
    }
  } else {
    // ================= consumers: spline arithmetic on the VALU out of the LDS parameter buffer =====
    const int ctid = tid - 256;
    if (ctid < kDt) cs[ctid] = a.cols[ctid];
    uint32_t err = 0;
    const int xvec = kR * D / 4;  // <= 1024 float4: up to four per consumer thread
    float4 xv[4];
    auto fetch_x = [&](int64_t t) {
      const float4* xg = reinterpret_cast<const float4*>(a.x + t * kR * D);
#pragma unroll
      for (int k = 0; k < 4; ++k) xv[k] = xg[ctid + k * 256 < xvec ? ctid + k * 256 : 0];
    };
    auto park_x = [&](int buf) {
      float4* xd = reinterpret_cast<float4*>(xbuf + buf * kR * D);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (ctid + k * 256 < xvec) xd[ctid + k * 256] = xv[k];
    };
    const bool inv = op.q.inverse != 0;
    auto consume = [&](int xb, int hf) {
      const float* phalf = pbuf + hf * kR * kPRow;
      float* xtile = xbuf + xb * kR * D;
      const int* ch = cs + hf * kHalfDims;
      err |= inv ? fused_consume<true>(op, phalf, xtile, ch, lbuf, D, ctid, hf == 0, a.debug & 2)
                 : fused_consume<false>(op, phalf, xtile, ch, lbuf, D, ctid, hf == 0, a.debug & 2);
    };

    fetch_x(tile0);
    park_x(0);
    __syncthreads();   // #0
    __syncthreads();   // #1
    int tb = 0;
    for (int64_t tile = tile0; tile < a.tiles; tile += stride) {
      const bool has_next = tile + stride < a.tiles;
      if (has_next) fetch_x(tile + stride);
      consume(tb, 0);
      if (has_next) park_x(tb ^ 1);
      __syncthreads();  // A
      consume(tb, 1);
      __syncthreads();  // B
      {
        float4* yg = reinterpret_cast<float4*>(a.y + tile * kR * D);
        const float4* xd = reinterpret_cast<const float4*>(xbuf + tb * kR * D);
        for (int i = ctid; i < xvec; i += 256) yg[i] = xd[i];
        if (ctid < kR) {
          const float v = lbuf[ctid];
          a.logabsdet[tile * kR + ctid] = op.q.inverse ? v : v;
        }
      }
      __syncthreads();  // C
      tb ^= 1;
    }
    if (err && a.err) atomicOr(a.err, err);
  }
}

}  // namespace fc

extern "C" int fc_rq_spline_fused_linear(const float* x, float* y, const float* h, const float* w_pad,
                                         const float* bias_pad, const int32_t* cols, float* logabsdet,
                                         uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t, int32_t hidden,
                                         const fc_rq_config* cfg, void* stream) {
  if (!cfg || n < 0 || d < d_t) return hipErrorInvalidValue;
  if (hidden != fc::kH || d_t != fc::kDt || cfg->num_bins != fc::kK || cfg->tails != 1 || d % 4 != 0 || d > 128)
    return hipErrorInvalidValue;  // only the north-star layer shape is fused; callers fall back otherwise
  if (n % fc::kR != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !h || !w_pad || !bias_pad || !cols || !logabsdet) return hipErrorInvalidValue;
  if ((((uintptr_t)h | (uintptr_t)x | (uintptr_t)y) & 15u) != 0) return hipErrorInvalidValue;

  fc::RQOp<fc::kK> op;
  fc::RQParams& q = op.q;
  q.K = cfg->num_bins; q.tails = 1; q.inverse = cfg->inverse;
  q.left = cfg->left; q.right = cfg->right; q.bottom = cfg->bottom; q.top = cfg->top;
  q.min_w = (float)cfg->min_bin_width; q.min_h = (float)cfg->min_bin_height; q.min_d = (float)cfg->min_derivative;
  q.cw = (float)(1.0 - cfg->min_bin_width * q.K);
  q.ch = (float)(1.0 - cfg->min_bin_height * q.K);
  q.wh_div = cfg->wh_divisor > 0.f ? cfg->wh_divisor : 1.f;
  q.beta = cfg->softplus_beta;
  q.tail_const = cfg->tail_constant;
  op.inv_div = 1.f / q.wh_div;
  op.inv_beta = 1.f / q.beta;

  const char* dbg = getenv("FC_FUSED_DEBUG");
  fc::FusedArgs a{x, y, h, w_pad, bias_pad, cols, logabsdet, err_flag, n / fc::kR, d, dbg ? atoi(dbg) : 0};
  const size_t lds = sizeof(float) * (size_t)(fc::kPBuf + fc::kHBuf + 2 * fc::kR * d + fc::kR) +
                     sizeof(int) * fc::kDt;
  if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
  // "1": f32 MFMA, producer/consumer waves; "2": f32 MFMA, symmetric waves; default: split-bf16 MFMA
  const char* which = getenv("FC_FUSED_KERNEL");
  int64_t grid0 = fc::device_cu_count();
  if (grid0 > a.tiles) grid0 = a.tiles;
  if (!which || (which[0] != '1' && which[0] != '2')) {
    if (fc::fused3_lds_bytes(d) <= 160 * 1024)
      return fc::launch_fused3(op, a, (unsigned)grid0, static_cast<hipStream_t>(stream));
    which = "2";
  }
  if (which[0] == '2')
    return fc::launch_fused2(op, a, lds, (unsigned)grid0, static_cast<hipStream_t>(stream));
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc::rq_fused_linear_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  int64_t grid = fc::device_cu_count();
  if (grid > a.tiles) grid = a.tiles;
  hipLaunchKernelGGL(fc::rq_fused_linear_kernel, dim3((unsigned)grid), dim3(512), lds,
                     static_cast<hipStream_t>(stream), op, a);
  return hipGetLastError();
}
