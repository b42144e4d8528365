// Backward of the rational-quadratic spline bijector (forward direction), gfx950.
//
// For y, logabsdet = rq_spline(x, params) (flowcon/transforms/splines/rational_quadratic.py:13-181 as called
// from coupling.py:279-293,549-582 / autoregressive.py:583-621) and upstream gradients gy = dL/dy [N, D],
// gl = dL/dlogabsdet [N], one thread per (sample, transformed dim) computes
//     dL/dx        = gy dy/dx + gl dlad/dx
//     dL/dparams_i = gy dy/dp_i + gl dlad/dp_i          for the 3K-1 (3K+1) raw conditioner outputs of its dim.
// This is what torch.autograd produces for the reference's op sequence; the reference trains through it
// (examples/toy_2d.py:57-68), SURVEY section 8(f) #3.
//
// Structure of the derivative.  The element depends on its parameters only through 7 numbers: x, the two
// knots (x_k, y_k), (x_k+1, y_k+1) of its bin and the two knot derivatives d_k, d_k+1.  The closed form of
// (y, lad) in those 7 is differentiated in reverse mode by hand (~60 flops); the chain to the raw parameters
// is analytic:
//     knot x_k = left + span * sum_{i<k} (min_w + cw p_i),  p = softmax(u / wh_div)
//     d x_k / d u_m = span cw p_m (1[m < k] - P_k) / wh_div,   P_k = sum_{i<k} p_i       (ends pinned: no gradient)
//     d_k = min_d + softplus(u, beta)  ->  d d_k / d u = sigmoid(beta u)   (1 beyond the softplus threshold)
// Outside the tail interval the bijector is the identity: dL/dx = gy, no parameter gradient.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_math.h"
#include "fc_rq_op.h"
#include "fc_rq_backward_op.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct RQBackwardArgs {
  const float* x;        // [N, D]
  const float* params;   // [N, d_t * P]
  const int32_t* cols;   // [d_t] or null
  const float* gy;       // [N, D]
  const float* gl;       // [N] or null (treated as zeros)
  float* gx;             // [N, D]: only the transformed columns are written
  float* gp;             // [N, d_t * P]
  int64_t n;
  int d, d_t;
};

// 16-byte accesses through a 4-byte aligned vector type: the P-float chunk of an element is only dword aligned
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int PS>
__device__ __forceinline__ void load_chunk(float (&dst)[PS], const float* __restrict__ src, int P) {
#pragma unroll
  for (int i = 0; i + 4 <= PS; i += 4)
    if (i + 4 <= P) {
      const f4u v = *reinterpret_cast<const f4u*>(src + i);
      dst[i] = v.x; dst[i + 1] = v.y; dst[i + 2] = v.z; dst[i + 3] = v.w;
    }
#pragma unroll
  for (int i = 0; i < PS; ++i)
    if (i >= (P & ~3) && i < P) dst[i] = src[i];
}
template <int PS>
__device__ __forceinline__ void store_chunk(float* __restrict__ dst, const float (&src)[PS], int P) {
#pragma unroll
  for (int i = 0; i + 4 <= PS; i += 4)
    if (i + 4 <= P) *reinterpret_cast<f4u*>(dst + i) = f4u{src[i], src[i + 1], src[i + 2], src[i + 3]};
#pragma unroll
  for (int i = 0; i < PS; ++i)
    if (i >= (P & ~3) && i < P) dst[i] = src[i];
}

// Generic kernel: one thread per (sample, dim), x / gy gathered and gx scattered element by element (the caller
// has copied gy into gx for the identity columns).
template <int KS>
__global__ __launch_bounds__(256) void rq_backward_kernel(RQParams q, float inv_div, RQBackwardArgs a) {
  const int K = KS > 0 ? KS : q.K;
  const int P = q.tails ? 3 * K - 1 : 3 * K + 1;
  const int64_t total = a.n * a.d_t;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / a.d_t;
    const int j = (int)(e - row * a.d_t);
    const int col = a.cols ? a.cols[j] : j;
    const float x = a.x[row * a.d + col];
    const float gy = a.gy[row * a.d + col];
    const float gl = a.gl ? a.gl[row] : 0.f;
    const float* uglob = a.params + (row * a.d_t + j) * P;
    float* gglob = a.gp + (row * a.d_t + j) * P;
    float gx;
    if constexpr (KS > 0) {
      constexpr int PS = 3 * KS + 1;   // register image, large enough for both tail modes
      float ureg[PS], greg[PS];
      load_chunk<PS>(ureg, uglob, P);
      rq_backward_element<KS>(q, inv_div, K, P, ureg, x, gy, gl, gx, greg);
      store_chunk<PS>(gglob, greg, P);
    } else {
      rq_backward_element<KS>(q, inv_div, K, P, uglob, x, gy, gl, gx, gglob);
    }
    a.gx[row * a.d + col] = gx;
  }
}

// Wave kernel (d_t a power of two <= 64, compile-time K): one wavefront owns G = 64 / d_t consecutive samples, as
// in the forward rq_wave_kernel.  Their x and gy rows pass through a per-wave LDS strip so that all row traffic
// is coalesced 16-byte accesses; the gy strip, with the transformed columns overwritten by dL/dx, IS the gx rows
// (identity columns pass through).
template <int KS>
__global__ __launch_bounds__(256) void rq_backward_wave_kernel(RQParams q, float inv_div, RQBackwardArgs a, int64_t groups) {
  constexpr int PS = 3 * KS + 1;
  extern __shared__ __attribute__((aligned(16))) float bsmem[];
  const int d_t = a.d_t, D = a.d;
  const int P = q.tails ? 3 * KS - 1 : 3 * KS + 1;
  const int sh = __builtin_ctz(d_t), G = 64 >> sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowf = G * D, strip = (rowf + 3) & ~3;
  const int chunk = (64 * P + 3) & ~3;   // the 64 gradient chunks of a wave: one contiguous, 16-byte aligned span
  int* cs = reinterpret_cast<int*>(bsmem);
  float* xw = bsmem + ((d_t + 3) & ~3) + wave * (2 * strip + chunk);
  float* gw = xw + strip;
  float* pw_out = gw + strip;
  for (int j = threadIdx.x; j < d_t; j += blockDim.x) cs[j] = a.cols ? a.cols[j] : j;
  __syncthreads();
  const int s = lane >> sh, j = lane & (d_t - 1);
  const int col = cs[j];
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; g < groups; g += wstride) {
    const int64_t n0 = g * G;
    float ureg[PS], greg[PS];
    {
      // parameters: the wave's contiguous span comes in as aligned 16-byte loads, each lane then picks its own
      // P values from LDS (stride P: conflict-free for odd P)
      const float4* src = reinterpret_cast<const float4*>(a.params + g * 64 * P);
      for (int i = lane; i < (64 * P) / 4; i += 64) reinterpret_cast<float4*>(pw_out)[i] = src[i];
#pragma unroll
      for (int i = 0; i < PS; ++i)
        if (i < P) ureg[i] = pw_out[lane * P + i];
    }
    const float* xg = a.x + n0 * D;
    const float* gg = a.gy + n0 * D;
    if ((rowf & 3) == 0) {
      for (int i = lane; i < (rowf >> 2); i += 64) {
        reinterpret_cast<float4*>(xw)[i] = reinterpret_cast<const float4*>(xg)[i];
        reinterpret_cast<float4*>(gw)[i] = reinterpret_cast<const float4*>(gg)[i];
      }
    } else {
      for (int i = lane; i < rowf; i += 64) {
        xw[i] = xg[i];
        gw[i] = gg[i];
      }
    }
    const float x = xw[s * D + col], gy = gw[s * D + col];
    const float gl = a.gl ? a.gl[n0 + s] : 0.f;
    float gx;
    rq_backward_element<KS>(q, inv_div, KS, P, ureg, x, gy, gl, gx, greg);
    gw[s * D + col] = gx;
    // parameter gradients: through the LDS strip (lane stride P floats: conflict-free for odd P), then out as
    // whole 16-byte pieces of the wave's contiguous span -- 4-byte aligned 16-byte stores straight from the
    // registers split into partial-sector writes
#pragma unroll
    for (int i = 0; i < PS; ++i)
      if (i < P) pw_out[lane * P + i] = greg[i];
    {
      float4* dst = reinterpret_cast<float4*>(a.gp + g * 64 * P);
      for (int i = lane; i < (64 * P) / 4; i += 64) dst[i] = reinterpret_cast<const float4*>(pw_out)[i];
    }
    float* og = a.gx + n0 * D;
    if ((rowf & 3) == 0) {
      for (int i = lane; i < (rowf >> 2); i += 64) reinterpret_cast<float4*>(og)[i] = reinterpret_cast<const float4*>(gw)[i];
    } else {
      for (int i = lane; i < rowf; i += 64) og[i] = gw[i];
    }
  }
}

template <int KS>
static hipError_t launch_bwd(const RQParams& q, RQBackwardArgs a, hipStream_t s) {
  const int K = KS > 0 ? KS : q.K;
  const int P = q.tails ? 3 * K - 1 : 3 * K + 1;
  const float inv_div = 1.f / q.wh_div;
  if constexpr (KS > 0) {
    const bool pow2 = (a.d_t & (a.d_t - 1)) == 0 && a.d_t <= 64;
    const int G = pow2 ? 64 / a.d_t : 1;
    const int64_t groups = pow2 ? a.n / G : 0;
    const bool aligned = ((G * a.d) % 4 != 0) ||
                         ((((uintptr_t)a.x | (uintptr_t)a.gy | (uintptr_t)a.gx) & 15u) == 0);
    const size_t lds = sizeof(float) * (size_t)(((a.d_t + 3) & ~3) + 4 * (2 * ((G * a.d + 3) & ~3) + ((64 * P + 3) & ~3)));
    if (groups >= 64 && aligned && lds <= 64 * 1024 && ((((uintptr_t)a.gp | (uintptr_t)a.params) & 15u) == 0)) {
      int64_t grid = 256 * 8;
      if (grid > (groups + 3) / 4) grid = (groups + 3) / 4;
      hipLaunchKernelGGL(rq_backward_wave_kernel<KS>, dim3((unsigned)grid), dim3(256), lds, s, q, inv_div, a, groups);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      const int64_t done = groups * G;
      if (done == a.n) return hipSuccess;
      a.x += done * a.d; a.gy += done * a.d; a.gx += done * a.d;
      a.params += done * (int64_t)a.d_t * P; a.gp += done * (int64_t)a.d_t * P;
      if (a.gl) a.gl += done;
      a.n -= done;
    }
  }
  // generic path: identity columns first (grad_x = grad_y), then one thread per element
  hipError_t e = hipMemcpyAsync(a.gx, a.gy, sizeof(float) * (size_t)a.n * a.d, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return e;
  const int64_t total = a.n * a.d_t;
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(rq_backward_kernel<KS>, dim3((unsigned)grid), dim3(256), 0, s, q, inv_div, a);
  return hipGetLastError();
}

}  // namespace fc

extern "C" int fc_rq_spline_backward(const float* x, const float* params, const int32_t* cols, const float* grad_y,
                                     const float* grad_logabsdet, float* grad_x, float* grad_params, int64_t n,
                                     int32_t d, int32_t d_t, const fc_rq_config* cfg, void* stream) {
  if (!cfg || n < 0 || d <= 0 || d_t <= 0 || d_t > d || cfg->num_bins <= 0 || cfg->num_bins > fc::kMaxBinsBwd)
    return hipErrorInvalidValue;
  if (cfg->inverse) return hipErrorInvalidValue;   // gradients of the forward direction only
  if (n == 0) return hipSuccess;
  if (!x || !params || !grad_y || !grad_x || !grad_params) return hipErrorInvalidValue;
  fc::RQParams q;
  q.K = cfg->num_bins;
  q.tails = cfg->tails;
  q.inverse = 0;
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
  fc::RQBackwardArgs a{x, params, cols, grad_y, grad_logabsdet, grad_x, grad_params, n, d, d_t};
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (q.K) {
    case 4: return fc::launch_bwd<4>(q, a, s);
    case 8: return fc::launch_bwd<8>(q, a, s);
    case 16: return fc::launch_bwd<16>(q, a, s);
    default: return fc::launch_bwd<0>(q, a, s);
  }
}
