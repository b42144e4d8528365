// Piecewise rational-quadratic spline bijector, forward + inverse + log|det J|, gfx950.
//
// Behaviour follows (restated, not copied) flowcon/transforms/splines/rational_quadratic.py:
//   :13-63   unconstrained wrapper (linear tails: identity outside [-B, B], derivative pad)
//   :66-181  knots = softmax -> min + (1 - min*K)*p -> cumsum -> affine -> pinned ends -> diffs,
//            derivatives = min_d + softplus(u, beta), compare-count bin search
//            (utils/torchutils.py:147-149), forward rational-quadratic / inverse quadratic root.
// One thread evaluates one (sample, dim); the K unnormalised widths/heights sit in LDS and
// are walked with static offsets so no runtime-indexed register array (-> scratch) exists.
#include "fc_tile.h"
#include "fc_math.h"
#include "fc_rq_op.h"
#include "../../include/flowcon_hip.h"

namespace fc {

// ---- wave-independent variant: parameters straight into registers ----------------------------------
//
// One wavefront owns G = 64 / d_t consecutive samples: lane (s, j) loads the P raw values of its own
// (sample, dim) -- a contiguous P*4-byte chunk, the 64 chunks of a wave forming ONE contiguous span of
// the conditioner output -- directly into registers.  No LDS image of the parameters, no workgroup
// barrier: waves drift freely, so one wave's HBM wait is another wave's arithmetic.  Only the G input
// rows (G*D floats) pass through a per-wave LDS strip so that identity columns are copied and global
// x / y traffic stays coalesced.  Requires d_t a power of two <= 64 and a compile-time K.
template <int KS, bool kTails>
__global__ __launch_bounds__(256) void rq_wave_kernel(RQOp<KS> op, TileArgs a, int64_t groups) {
  constexpr int P = KS > 0 ? (kTails ? 3 * KS - 1 : 3 * KS + 1) : 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d_t = a.d_t, D = a.D;
  const int sh = __builtin_ctz(d_t);
  const int G = 64 >> sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int* cs = reinterpret_cast<int*>(smem);
  float* xw = smem + round4(d_t) + wave * round4(G * D);
  for (int j = threadIdx.x; j < d_t; j += blockDim.x) cs[j] = a.cols ? a.cols[j] : j;
  __syncthreads();
  const int s = lane >> sh, j = lane & (d_t - 1);
  const int col = cs[j];
  const int row_floats = G * D;
  uint32_t err = 0;
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  // 16-byte loads through a 4-byte aligned vector type (the lane's chunk is P*4 bytes from its
  // neighbour's, so only dword aligned; global memory allows it): ~P/4 load instructions, not P
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  auto load_params = [&](float (&dst)[P], int64_t g) {
    const float* __restrict__ pp = a.params + (g * 64 + lane) * P;  // (n0 * d_t + lane) * P, n0 = g * G
#pragma unroll
    for (int i = 0; i + 4 <= P; i += 4) {
      const f4u v = *reinterpret_cast<const f4u*>(pp + i);
      dst[i] = v.x; dst[i + 1] = v.y; dst[i + 2] = v.z; dst[i + 3] = v.w;
    }
#pragma unroll
    for (int i = P & ~3; i < P; ++i) dst[i] = pp[i];
  };
  // (a register double-buffer prefetching the next group's parameters was measured: +31 VGPRs, no
  //  gain -- 4 resident waves per SIMD already cover the load latency; the kernel sits at ~80 % of the
  //  per-CU load-path rate)
  for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; g < groups; g += wstride) {
    const int64_t n0 = g * G;
    float p[P];
    load_params(p, g);
    const float* __restrict__ xg = a.x + n0 * D;
    if ((row_floats & 255) == 0) {
      for (int i = lane; i < (row_floats >> 2); i += 64)
        reinterpret_cast<float4*>(xw)[i] = reinterpret_cast<const float4*>(xg)[i];
    } else if ((row_floats & 127) == 0) {
      for (int i = lane; i < (row_floats >> 1); i += 64)
        reinterpret_cast<float2*>(xw)[i] = reinterpret_cast<const float2*>(xg)[i];
    } else {
      for (int i = lane; i < row_floats; i += 64) xw[i] = xg[i];
    }
    const float xv = xw[s * D + col];
    float yv, lad;
    op.template eval_core<true>(p, xv, yv, lad, err);
    xw[s * D + col] = yv;
    if (a.logabsdet) {
      const float tot = group_sum_rt(lad, d_t);
      if (j == 0) a.logabsdet[n0 + s] = (a.lad_mode & 2) ? -tot : tot;
    }
    float* __restrict__ yg = a.y + n0 * D;
    if ((row_floats & 255) == 0) {
      for (int i = lane; i < (row_floats >> 2); i += 64)
        reinterpret_cast<float4*>(yg)[i] = reinterpret_cast<const float4*>(xw)[i];
    } else if ((row_floats & 127) == 0) {
      for (int i = lane; i < (row_floats >> 1); i += 64)
        reinterpret_cast<float2*>(yg)[i] = reinterpret_cast<const float2*>(xw)[i];
    } else {
      for (int i = lane; i < row_floats; i += 64) yg[i] = xw[i];
    }
  }
  if (err && a.err) atomicOr(a.err, err);
}

template <int KS>
static hipError_t launch_rq(const RQParams& q, TileArgs a, int flags, hipStream_t stream) {
  RQOp<KS> op;
  op.q = q;
  op.inv_div = 1.f / q.wh_div;
  op.inv_beta = 1.f / q.beta;
  const bool pow2 = (a.d_t & (a.d_t - 1)) == 0 && a.d_t <= 64;
  const bool force_tile = (flags & FC_RQ_FORCE_TILE) != 0;   // pin the LDS-tile structure (A/B measurements)
  if constexpr (KS > 0 && KS <= 10)
  if (pow2 && !a.shared_params && a.N > 0 && a.lad_mode != 1 && a.lad_mode != 3 &&
      !force_tile) {
    const int G = 64 / a.d_t;
    const int64_t groups = a.N / G;
    const bool al = ((G * a.D) % 4 != 0) || (aligned16(a.x) && aligned16(a.y));
    if (groups >= 256 && al) {
      const size_t lds = sizeof(float) * (size_t)(round4(a.d_t) + 4 * round4(G * a.D));
      int64_t grid = (int64_t)device_cu_count() * 6;
      if (grid > (groups + 3) / 4) grid = (groups + 3) / 4;
      TileArgs body = a;
      if (q.tails)
        hipLaunchKernelGGL((rq_wave_kernel<KS, true>), dim3((unsigned)grid), dim3(256), lds, stream, op, body, groups);
      else
        hipLaunchKernelGGL((rq_wave_kernel<KS, false>), dim3((unsigned)grid), dim3(256), lds, stream, op, body, groups);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      const int64_t done = groups * G;
      if (done == a.N) return hipSuccess;
      const int P = q.tails ? 3 * KS - 1 : 3 * KS + 1;
      a.x += done * a.D;
      a.y += done * a.D;
      a.params += done * (int64_t)a.d_t * P;
      if (a.logabsdet) a.logabsdet += done;
      a.N -= done;
    }
  }
  return launch_tile(op, a, stream);
}

}  // namespace fc

extern "C" int fc_rq_spline(const float* x, float* y, const float* params, const int32_t* cols,
                            float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d,
                            int32_t d_t, int32_t shared_params, int32_t lad_mode,
                            const fc_rq_config* cfg, void* stream) {
  if (!cfg || n < 0 || d <= 0 || d_t <= 0 || d_t > d || cfg->num_bins <= 0) return hipErrorInvalidValue;
  if (n > 0 && (!x || !y || !params)) return hipErrorInvalidValue;
  fc::RQParams q;
  q.K = cfg->num_bins;
  q.tails = cfg->tails;
  q.inverse = cfg->inverse;
  q.left = cfg->left; q.right = cfg->right; q.bottom = cfg->bottom; q.top = cfg->top;
  q.min_w = (float)cfg->min_bin_width;
  q.min_h = (float)cfg->min_bin_height;
  q.min_d = (float)cfg->min_derivative;
  q.cw = (float)(1.0 - cfg->min_bin_width * q.K);
  q.ch = (float)(1.0 - cfg->min_bin_height * q.K);
  fc::rq_finish_params(q);
  q.wh_div = cfg->wh_divisor > 0.f ? cfg->wh_divisor : 1.f;
  q.beta = cfg->softplus_beta;
  q.tail_const = cfg->tail_constant;

  fc::TileArgs a{};
  a.x = x; a.y = y; a.params = params; a.cols = cols; a.logabsdet = logabsdet; a.err = err_flag;
  a.N = n; a.D = d; a.d_t = d_t;
  a.rowlen = d_t * (q.tails ? 3 * q.K - 1 : 3 * q.K + 1);
  a.shared_params = shared_params;
  a.lad_mode = lad_mode;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (q.K) {
    case 4: return fc::launch_rq<4>(q, a, cfg->flags, s);
    case 5: return fc::launch_rq<5>(q, a, cfg->flags, s);
    case 8: return fc::launch_rq<8>(q, a, cfg->flags, s);
    case 10: return fc::launch_rq<10>(q, a, cfg->flags, s);
    case 16: return fc::launch_rq<16>(q, a, cfg->flags, s);
    default: return fc::launch_rq<0>(q, a, cfg->flags, s);
  }
}
