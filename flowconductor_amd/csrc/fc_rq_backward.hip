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
#include "../../include/flowcon_hip.h"

namespace fc {

constexpr int kMaxBinsBwd = 32;

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

// softmax probabilities of one axis (p[i], i < K) from the raw logits, as the forward computes them
template <int KS>
__device__ __forceinline__ void softmax_axis(const float* __restrict__ u, int K, float inv_div, float* __restrict__ p) {
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < (KS > 0 ? KS : kMaxBinsBwd); ++i)
    if (i < K) {
      p[i] = u[i] * inv_div;
      m = fmaxf(m, p[i]);
    }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < (KS > 0 ? KS : kMaxBinsBwd); ++i)
    if (i < K) {
      p[i] = exp_lean(p[i] - m);
      sum += p[i];
    }
  const float rs = div_lean(1.f, sum);
#pragma unroll
  for (int i = 0; i < (KS > 0 ? KS : kMaxBinsBwd); ++i)
    if (i < K) p[i] *= rs;
}

// knots of one axis around bin `idx` (found on this axis if kSearch): lower / upper knot and the prefix sums
// of the probabilities below them
template <int KS, bool kSearch>
__device__ __forceinline__ void knots_axis(const float* __restrict__ p, int K, float minb, float c1, float lo, float hi,
                                           float v, int& idx, float& k_lo, float& k_hi, float& pre_lo, float& pre_hi) {
  const float span = hi - lo;
  double cum = 0.0;   // ATen's CPU cumsum accumulates f32 in double (as the forward kernels do)
  float psum = 0.f, prev = lo, prevp = 0.f;
  int found = kSearch ? 0 : idx;
  k_lo = lo; k_hi = lo; pre_lo = 0.f; pre_hi = 0.f;
#pragma unroll
  for (int i = 0; i < (KS > 0 ? KS : kMaxBinsBwd); ++i)
    if (i < K) {
      cum += (double)(minb + c1 * p[i]);
      psum += p[i];
      const float next = (i == K - 1) ? hi : (span * (float)cum + lo);
      const bool take = kSearch ? (v >= prev) : (i == idx);
      if (take) {
        found = i;
        k_lo = prev; k_hi = next; pre_lo = prevp; pre_hi = psum;
      }
      prev = next;
      prevp = psum;
    }
  idx = found;
}

// One element: upstream (gy, gl) -> gx and the P parameter gradients gp[0..P) (u / gp: registers when KS > 0).
template <int KS>
__device__ __forceinline__ void rq_backward_element(const RQParams& q, float inv_div, int K, int P,
                                                    const float* __restrict__ u, float x, float gy, float gl,
                                                    float& gx, float* __restrict__ gp) {
  const bool inside = (x >= q.left) && (x <= q.right);
  if (!inside) {   // identity tails (or, without tails, an input the forward already rejected)
    gx = gy;
#pragma unroll
    for (int i = 0; i < (KS > 0 ? 3 * KS + 1 : 3 * kMaxBinsBwd + 1); ++i)
      if (i < P) gp[i] = 0.f;
    return;
  }
  float pw[KS > 0 ? KS : kMaxBinsBwd], ph[KS > 0 ? KS : kMaxBinsBwd];
  softmax_axis<KS>(u, K, inv_div, pw);
  softmax_axis<KS>(u + K, K, inv_div, ph);
  int idx = 0;
  float xk, xk1, pwk, pwk1, yk, yk1, phk, phk1;
  knots_axis<KS, true>(pw, K, q.min_w, q.cw, q.left, q.right, x, idx, xk, xk1, pwk, pwk1);
  knots_axis<KS, false>(ph, K, q.min_h, q.ch, q.bottom, q.top, x, idx, yk, yk1, phk, phk1);

  // knot derivatives and their slopes with respect to the raw value
  const float* ud = u + 2 * K;
  const int i0 = q.tails ? idx - 1 : idx, i1 = q.tails ? idx : idx + 1;      // positions in ud
  const bool has0 = !q.tails || idx > 0, has1 = !q.tails || idx < K - 1;
  float u0 = q.tail_const, u1 = q.tail_const;
  if constexpr (KS > 0) {   // register image: static indices only
#pragma unroll
    for (int i = 0; i < KS + 1; ++i) {
      if (has0 && i == i0) u0 = ud[i];
      if (has1 && i == i1) u1 = ud[i];
    }
  } else {
    if (has0) u0 = ud[i0];
    if (has1) u1 = ud[i1];
  }
  const float d0v = q.min_d + softplus_lean(u0, q.beta), d1v = q.min_d + softplus_lean(u1, q.beta);
  const float s0 = (u0 * q.beta > 20.f) ? 1.f : div_lean(1.f, 1.f + exp_lean(fminf(-u0 * q.beta, 80.f)));
  const float s1 = (u1 * q.beta > 20.f) ? 1.f : div_lean(1.f, 1.f + exp_lean(fminf(-u1 * q.beta, 80.f)));

  // (y, lad) as functions of (x, x_k, x_k+1, y_k, y_k+1, d_k, d_k+1), rational_quadratic.py:162-181, and their
  // reverse-mode derivative (~60 flops: every intermediate below gets one adjoint, accumulated from its uses)
  const float wk = xk1 - xk, hk = yk1 - yk;
  const float rwk = div_lean(1.f, wk);
  const float delta = hk * rwk;
  const float theta = (x - xk) * rwk;
  const float omt = 1.f - theta, t1 = theta * omt, th2 = theta * theta;
  const float a1 = delta * th2 + d0v * t1;            // y = y_k + h_k a1 / den
  const float num = hk * a1;
  const float sd = d0v + d1v - 2.f * delta;
  const float den = delta + sd * t1;
  const float b1 = d1v * th2 + 2.f * delta * t1 + d0v * (omt * omt);   // lad = log(delta^2 b1) - 2 log(den)
  const float dnum = delta * delta * b1;
  const float rden = div_lean(1.f, den);
  const float g_num = gy * rden;
  const float g_den = -(gy * num * rden + 2.f * gl) * rden;
  const float g_dnum = gl * div_lean(1.f, dnum);
  const float g_b1 = g_dnum * delta * delta;
  const float g_a1 = g_num * hk;
  const float g_sd = g_den * t1;
  float g_delta = g_dnum * 2.f * delta * b1 + g_b1 * 2.f * t1 + g_den - 2.f * g_sd + g_a1 * th2;
  const float g_d1 = g_b1 * th2 + g_sd;
  const float g_d0 = g_b1 * (omt * omt) + g_sd + g_a1 * t1;
  const float g_th2 = g_b1 * d1v + g_a1 * delta;
  const float g_t1 = g_b1 * 2.f * delta + g_den * sd + g_a1 * d0v;
  const float g_omt = g_b1 * d0v * 2.f * omt + g_t1 * theta;
  const float g_theta = g_th2 * 2.f * theta + g_t1 * omt - g_omt;
  float g_hk = g_num * a1 + g_delta * rwk;
  const float g_dx = g_theta * rwk;
  const float g_wk = -(g_theta * theta + g_delta * delta) * rwk;
  float gq[7];
  gq[0] = g_dx;                 // x
  gq[1] = -g_dx - g_wk;         // x_k
  gq[2] = g_wk;                 // x_k+1
  gq[3] = gy - g_hk;            // y_k
  gq[4] = g_hk;                 // y_k+1
  gq[5] = g_d0;
  gq[6] = g_d1;
  gx = gq[0];

  // chain to the raw parameters
  const float cx = (q.right - q.left) * q.cw * inv_div, cy = (q.top - q.bottom) * q.ch * inv_div;
  const float gxk = idx > 0 ? gq[1] : 0.f, gxk1 = idx + 1 < K ? gq[2] : 0.f;   // pinned end knots
  const float gyk = idx > 0 ? gq[3] : 0.f, gyk1 = idx + 1 < K ? gq[4] : 0.f;
#pragma unroll
  for (int m = 0; m < (KS > 0 ? KS : kMaxBinsBwd); ++m)
    if (m < K) {
      const float below_lo = m < idx ? 1.f : 0.f, below_hi = m < idx + 1 ? 1.f : 0.f;
      gp[m] = cx * pw[m] * (gxk * (below_lo - pwk) + gxk1 * (below_hi - pwk1));
      gp[K + m] = cy * ph[m] * (gyk * (below_lo - phk) + gyk1 * (below_hi - phk1));
    }
  const int nd = P - 2 * K;
#pragma unroll
  for (int i = 0; i < (KS > 0 ? KS + 1 : kMaxBinsBwd + 1); ++i)
    if (i < nd) {
      float v = 0.f;
      if (has0 && i == i0) v += gq[5] * s0;
      if (has1 && i == i1) v += gq[6] * s1;
      gp[2 * K + i] = v;
    }
}

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
