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
// (y, lad) in those 7 is differentiated by forward-mode dual numbers (7 partials carried through ~30
// operations); the chain to the raw parameters is analytic:
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

// value + partials with respect to (x, x_k, x_k+1, y_k, y_k+1, d_k, d_k+1)
struct Dual7 {
  float v;
  float g[7];
};
__device__ __forceinline__ Dual7 dconst(float c) {
  Dual7 r;
  r.v = c;
#pragma unroll
  for (int i = 0; i < 7; ++i) r.g[i] = 0.f;
  return r;
}
__device__ __forceinline__ Dual7 dvar(float v, int which) {
  Dual7 r = dconst(v);
  r.g[which] = 1.f;
  return r;
}
__device__ __forceinline__ Dual7 operator+(const Dual7& a, const Dual7& b) {
  Dual7 r;
  r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < 7; ++i) r.g[i] = a.g[i] + b.g[i];
  return r;
}
__device__ __forceinline__ Dual7 operator-(const Dual7& a, const Dual7& b) {
  Dual7 r;
  r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < 7; ++i) r.g[i] = a.g[i] - b.g[i];
  return r;
}
__device__ __forceinline__ Dual7 operator*(const Dual7& a, const Dual7& b) {
  Dual7 r;
  r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < 7; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i];
  return r;
}
__device__ __forceinline__ Dual7 operator*(float s, const Dual7& a) {
  Dual7 r;
  r.v = s * a.v;
#pragma unroll
  for (int i = 0; i < 7; ++i) r.g[i] = s * a.g[i];
  return r;
}
__device__ __forceinline__ Dual7 operator/(const Dual7& a, const Dual7& b) {
  Dual7 r;
  const float inv = div_lean(1.f, b.v);
  r.v = a.v * inv;
#pragma unroll
  for (int i = 0; i < 7; ++i) r.g[i] = (a.g[i] - r.v * b.g[i]) * inv;
  return r;
}
__device__ __forceinline__ Dual7 dlog(const Dual7& a) {
  Dual7 r;
  const float inv = div_lean(1.f, a.v);
  r.v = log_lean(a.v);
#pragma unroll
  for (int i = 0; i < 7; ++i) r.g[i] = a.g[i] * inv;
  return r;
}

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
    // The P raw values of this element are one contiguous chunk, consecutive threads own consecutive chunks:
    // with a compile-time K they are moved as 16-byte accesses through a 4-byte aligned vector type
    // (~P/4 memory instructions instead of P), parameters and gradients both living in registers.
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int PS = KS > 0 ? 3 * KS + 1 : 1;   // register image (large enough for both tail modes)
    float ureg[PS], greg[PS];
    const float* uglob = a.params + (row * a.d_t + j) * P;
    float* gglob = a.gp + (row * a.d_t + j) * P;
    const float* u = uglob;
    float* gp = gglob;
    if constexpr (KS > 0) {
#pragma unroll
      for (int i = 0; i + 4 <= PS; i += 4)
        if (i + 4 <= P) {
          const f4u v = *reinterpret_cast<const f4u*>(uglob + i);
          ureg[i] = v.x; ureg[i + 1] = v.y; ureg[i + 2] = v.z; ureg[i + 3] = v.w;
        }
#pragma unroll
      for (int i = 0; i < PS; ++i)
        if (i >= (P & ~3) && i < P) ureg[i] = uglob[i];
      u = ureg;
      gp = greg;
    }

    const bool inside = (x >= q.left) && (x <= q.right);
    auto flush = [&]() {   // register gradients -> memory
      if constexpr (KS > 0) {
#pragma unroll
        for (int i = 0; i + 4 <= PS; i += 4)
          if (i + 4 <= P) *reinterpret_cast<f4u*>(gglob + i) = f4u{greg[i], greg[i + 1], greg[i + 2], greg[i + 3]};
#pragma unroll
        for (int i = 0; i < PS; ++i)
          if (i >= (P & ~3) && i < P) gglob[i] = greg[i];
      }
    };
    if (!inside) {   // identity tails (or, without tails, an input the forward already rejected)
      a.gx[row * a.d + col] = gy;
      if constexpr (KS > 0) {
#pragma unroll
        for (int i = 0; i < PS; ++i) greg[i] = 0.f;
      } else {
        for (int i = 0; i < P; ++i) gp[i] = 0.f;
      }
      flush();
      continue;
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

    // (y, lad) as functions of (x, x_k, x_k+1, y_k, y_k+1, d_k, d_k+1): rational_quadratic.py:162-181
    const Dual7 X = dvar(x, 0), XK = dvar(xk, 1), XK1 = dvar(xk1, 2), YK = dvar(yk, 3), YK1 = dvar(yk1, 4);
    const Dual7 D0 = dvar(d0v, 5), D1 = dvar(d1v, 6);
    const Dual7 wk = XK1 - XK, hk = YK1 - YK;
    const Dual7 delta = hk / wk;
    const Dual7 theta = (X - XK) / wk;
    const Dual7 omt = dconst(1.f) - theta;
    const Dual7 t1 = theta * omt;
    const Dual7 th2 = theta * theta;
    const Dual7 num = hk * (delta * th2 + D0 * t1);
    const Dual7 den = delta + (D0 + D1 - 2.f * delta) * t1;
    const Dual7 y = YK + num / den;
    const Dual7 dnum = (delta * delta) * (D1 * th2 + 2.f * (delta * t1) + D0 * (omt * omt));
    const Dual7 lad = dlog(dnum) - 2.f * dlog(den);

    float gq[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) gq[i] = gy * y.g[i] + gl * lad.g[i];
    a.gx[row * a.d + col] = gq[0];

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
    flush();
  }
}

template <int KS>
static hipError_t launch_bwd(const RQParams& q, const RQBackwardArgs& a, hipStream_t s) {
  const int64_t total = a.n * a.d_t;
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(rq_backward_kernel<KS>, dim3((unsigned)grid), dim3(256), 0, s, q, 1.f / q.wh_div, a);
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
