// Backward of the piecewise linear / quadratic / cubic spline bijectors (forward direction), gfx950.
//
// What torch.autograd yields for flowcon/transforms/splines/linear.py:38-105, quadratic.py:55-159 and cubic.py:63-267
// (+ their `unconstrained_*` wrappers): grad_x [N, D] (transformed columns; the caller adds the identity columns) and
// grad_params [N, d_t P] from grad_y [N, D] and grad_logabsdet [N].
//
// These splines renormalise their parameters several times (softmax, trapezoid area, boundary heights, monotone knot
// slopes), so instead of a hand-derived adjoint per spline the kernel differentiates the forward evaluation itself in
// FORWARD mode: one thread per (element, direction) runs the element's evaluation on dual numbers (value, tangent) with
// the tangent seeded on ONE of the element's P raw parameters or on x, and writes that one gradient entry
//     grad = grad_y dy/dtheta + grad_logabsdet dlad/dtheta.
// P + 1 evaluations per element instead of one, but every thread is independent (no atomics, no reductions), the writes
// are coalesced, and the derivative is the derivative of exactly the arithmetic the forward kernel performs (the value
// parts restate fc_splines.hip's forward branches).  Working set: P dual numbers per thread in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_device.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct Dual {
  float v, d;
};
__device__ __forceinline__ Dual mk(float v) { return Dual{v, 0.f}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual{a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const float q = a.v / b.v;
  return Dual{q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator+(Dual a, float b) { return Dual{a.v + b, a.d}; }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return Dual{a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, float b) { return Dual{a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return Dual{a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return Dual{a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return Dual{a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, float b) { return Dual{a.v / b, a.d / b}; }
__device__ __forceinline__ Dual dexp(Dual a) {
  const float e = expf(a.v);
  return Dual{e, e * a.d};
}
__device__ __forceinline__ Dual dlog(Dual a) { return Dual{logf(a.v), a.d / a.v}; }
__device__ __forceinline__ Dual dabs(Dual a) { return a.v < 0.f ? Dual{-a.v, -a.d} : a; }
__device__ __forceinline__ Dual dmin(Dual a, Dual b) { return b.v < a.v ? b : a; }
__device__ __forceinline__ float dsgn(Dual a) { return (float)((a.v > 0.f) - (a.v < 0.f)); }
// torch.clamp(x, 0, 1): zero slope outside
__device__ __forceinline__ Dual dclamp01(Dual a) { return a.v < 0.f ? mk(0.f) : (a.v > 1.f ? mk(1.f) : a); }
// F.softplus (threshold 20) and torch.sigmoid
__device__ __forceinline__ Dual dsoftplus(Dual a) {
  if (a.v > 20.f) return a;
  const float e = expf(a.v);
  return Dual{log1pf(e), a.d * (e / (1.f + e))};
}
__device__ __forceinline__ Dual dsigmoid(Dual a) {
  const float s = 1.f / (1.f + expf(-a.v));
  return Dual{s, a.d * s * (1.f - s)};
}

struct SplineBwdArgs {
  const float* x;        // [N, D]
  const float* params;   // [N, d_t P]
  const int32_t* cols;   // [d_t] or null
  const float* gy;       // [N, D]
  const float* gl;       // [N] or null
  float* gx;             // [N, D] (transformed columns written)
  float* gp;             // [N, d_t P]
  int64_t total;         // N d_t (P + 1)
  int D, d_t, P, K;
  int kind, tails;
  float left, right, bottom, top;
  float min_w, min_h, cw, ch, w_div, h_div;
};

// p[i] = floor + c1 softmax(p / div)_i in place
__device__ __forceinline__ void dsoftmax(Dual* __restrict__ p, int K, float div, float floor_v, float c1) {
  float m = -INFINITY;
  for (int i = 0; i < K; ++i) {
    p[i] = p[i] / div;
    m = fmaxf(m, p[i].v);
  }
  Dual sum = mk(0.f);
  for (int i = 0; i < K; ++i) {
    p[i] = dexp(p[i] - m);
    sum = sum + p[i];
  }
  for (int i = 0; i < K; ++i) p[i] = floor_v + c1 * (p[i] / sum);
}

// linear.py:38-75 (forward): out = cdf[idx] + alpha pdf[idx], lad = log pdf[idx] + log K
__device__ __forceinline__ void linear_forward(const SplineBwdArgs& a, Dual* __restrict__ p, Dual xn, Dual& out, Dual& lad) {
  const int K = a.K;
  dsoftmax(p, K, 1.f, 0.f, 1.f);
  const Dual bin_pos = xn * (float)K;
  int idx = (int)floorf(bin_pos.v);
  idx = idx >= K ? K - 1 : (idx < 0 ? 0 : idx);
  const Dual alpha = bin_pos - (float)idx;
  Dual cum = mk(0.f);
  for (int i = 0; i < idx; ++i) cum = cum + p[i];
  out = dclamp01(cum + alpha * p[idx]);
  lad = dlog(p[idx]) - logf(1.f / (float)K);
}

// quadratic.py:55-159 (forward)
__device__ __forceinline__ void quadratic_forward(const SplineBwdArgs& a, Dual* __restrict__ p, Dual xn, Dual& out, Dual& lad) {
  const int K = a.K, nh = a.tails ? K - 1 : K + 1;
  Dual* w = p;
  Dual* h = p + K;
  dsoftmax(w, K, a.w_div, a.min_w, a.cw);
  for (int i = 0; i < nh; ++i) h[i] = dsoftplus(h[i] / a.h_div) + 1e-3f;
  Dual edge = mk(0.f);
  if (a.tails) {
    const Dual first_w = 0.5f * w[0], last_w = 0.5f * w[K - 1];
    Dual mid = mk(0.f);
    for (int i = 0; i + 1 < K - 1; ++i) mid = mid + ((h[i] + h[i + 1]) / 2.f) * w[i + 1];
    const Dual numer = 0.5f * first_w * h[0] + 0.5f * last_w * h[K - 2] + mid;
    edge = numer / (1.f - 0.5f * first_w - 0.5f * last_w);
  }
  auto knot_h = [&](int i) { return !a.tails ? h[i] : ((i == 0 || i == K) ? edge : h[i - 1]); };
  Dual area = mk(0.f);
  for (int i = 0; i < K; ++i) area = area + ((knot_h(i) + knot_h(i + 1)) / 2.f) * w[i];
  const float oh = 1.f - a.min_h;
  Dual cum_cdf = mk(0.f), cum_loc = mk(0.f), lo_cdf = mk(0.f), lo_loc = mk(0.f);
  Dual hl_prev = a.min_h + oh * (knot_h(0) / area);
  Dual loc = mk(0.f), wk = w[0], lc = mk(0.f), hl = hl_prev, hr = hl_prev;
  for (int i = 0; i < K; ++i) {
    const Dual hr_i = a.min_h + oh * (knot_h(i + 1) / area);
    cum_cdf = cum_cdf + ((hl_prev + hr_i) / 2.f) * w[i];
    cum_loc = cum_loc + w[i];
    const Dual hi_cdf = (i == K - 1) ? mk(1.f) : cum_cdf;
    const Dual hi_loc = (i == K - 1) ? mk(1.f) : cum_loc;
    if (xn.v >= lo_loc.v) {
      loc = lo_loc;
      wk = w[i];
      lc = lo_cdf;
      hl = hl_prev;
      hr = hr_i;
    }
    lo_cdf = hi_cdf;
    lo_loc = hi_loc;
    hl_prev = hr_i;
  }
  const Dual qa = 0.5f * (hr - hl) * wk, qb = hl * wk;
  const Dual alpha = (xn - loc) / wk;
  out = dclamp01(qa * (alpha * alpha) + qb * alpha + lc);
  lad = dlog(alpha * (hr - hl) + hl);
}

// cubic.py:63-267 (forward)
__device__ __forceinline__ void cubic_forward(const SplineBwdArgs& a, Dual* __restrict__ p, Dual xn, Dual& out, Dual& lad) {
  const int K = a.K;
  Dual* w = p;
  Dual* h = p + K;
  dsoftmax(w, K, a.w_div, a.min_w, a.cw);
  dsoftmax(h, K, a.h_div, a.min_h, a.ch);
  Dual cwid = mk(0.f), chgt = mk(0.f), lo_w = mk(0.f), lo_h = mk(0.f), left_w = mk(0.f), dco = mk(0.f);
  int idx = 0;
  for (int i = 0; i < K; ++i) {
    cwid = cwid + w[i];
    chgt = chgt + h[i];
    const Dual hi_w = (i == K - 1) ? mk(1.f) : cwid;
    const Dual hi_h = (i == K - 1) ? mk(1.f) : chgt;
    if (xn.v >= lo_w.v) {
      idx = i;
      left_w = lo_w;
      dco = lo_h;
    }
    lo_w = hi_w;
    lo_h = hi_h;
  }
  auto interior = [&](int i) {   // knot slope between bins i and i + 1 (cubic.py:113-131)
    const Dual s0 = h[i] / w[i], s1 = h[i + 1] / w[i + 1];
    const Dual m1 = dmin(dabs(s0), dabs(s1));
    const Dual m2 = 0.5f * (w[i + 1] * s0 + w[i] * s1) / (w[i] + w[i + 1]);
    return dmin(m1, m2) * (dsgn(s0) + dsgn(s1));
  };
  const Dual wk = w[idx];
  const Dual s = h[idx] / wk;
  const Dual dl = idx == 0 ? dsigmoid(p[2 * K]) * 3.f * (h[0] / w[0]) : interior(idx - 1);
  const Dual dr = idx == K - 1 ? dsigmoid(p[2 * K + 1]) * 3.f * (h[K - 1] / w[K - 1]) : interior(idx);
  const Dual ca = (dl + dr - 2.f * s) / (wk * wk);
  const Dual cb = (3.f * s - 2.f * dl - dr) / wk;
  const Dual t = xn - left_w;
  out = ca * (t * t * t) + cb * (t * t) + dl * t + dco;
  lad = dlog(3.f * ca * (t * t) + 2.f * cb * t + dl);
}

__global__ __launch_bounds__(256) void splines_backward_kernel(SplineBwdArgs a) {
  extern __shared__ Dual dual_smem[];
  const int P = a.P, dirs = P + 1;
  Dual* p = dual_smem + (size_t)threadIdx.x * P;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t elem = t / dirs;
    const int dir = (int)(t - elem * dirs);              // < P: raw parameter `dir`;  == P: the input
    const int64_t n = elem / a.d_t;
    const int j = (int)(elem - n * a.d_t);
    const int col = a.cols ? a.cols[j] : j;
    const float x = a.x[n * a.D + col];
    const float gy = a.gy[n * a.D + col], gl = a.gl ? a.gl[n] : 0.f;
    const bool inside = (x >= a.left) && (x <= a.right);
    float g;
    if (!inside) {
      g = dir == P ? gy : 0.f;                            // identity outside the interval, no logabsdet
    } else {
      const float* raw = a.params + elem * P;
      for (int i = 0; i < P; ++i) p[i] = Dual{raw[i], i == dir ? 1.f : 0.f};
      const float span_in = a.right - a.left, span_out = a.top - a.bottom;
      const Dual xn = Dual{(x - a.left) / span_in, dir == P ? 1.f / span_in : 0.f};
      Dual out, lad;
      if (a.kind == FC_SPLINE_LINEAR) linear_forward(a, p, xn, out, lad);
      else if (a.kind == FC_SPLINE_QUADRATIC) quadratic_forward(a, p, xn, out, lad);
      else cubic_forward(a, p, xn, out, lad);
      g = gy * (out.d * span_out) + gl * lad.d;
    }
    if (dir == P) a.gx[n * a.D + col] = g;
    else a.gp[elem * P + dir] = g;
  }
}

}  // namespace fc

extern "C" int fc_piecewise_spline_backward(const float* x, const float* params, const int32_t* cols, const float* grad_y,
                                            const float* grad_logabsdet, float* grad_x, float* grad_params, int64_t n,
                                            int32_t d, int32_t d_t, const fc_spline_config* cfg, void* stream) {
  if (!cfg || n < 0 || d <= 0 || d_t <= 0 || d_t > d || cfg->num_bins <= 0 || cfg->inverse) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !params || !grad_y || !grad_x || !grad_params) return hipErrorInvalidValue;
  const int K = cfg->num_bins;
  int P;
  switch (cfg->kind) {
    case FC_SPLINE_LINEAR: P = K; break;
    case FC_SPLINE_QUADRATIC: P = cfg->tails ? 2 * K - 1 : 2 * K + 1; break;
    case FC_SPLINE_CUBIC: P = 2 * K + 2; break;
    default: return hipErrorInvalidValue;
  }
  const size_t lds = (size_t)256 * P * sizeof(fc::Dual);
  if (lds > 64 * 1024) return hipErrorInvalidValue;      // P <= 32 parameters per element
  fc::SplineBwdArgs a{};
  a.x = x; a.params = params; a.cols = cols; a.gy = grad_y; a.gl = grad_logabsdet; a.gx = grad_x; a.gp = grad_params;
  a.total = n * (int64_t)d_t * (P + 1);
  a.D = d; a.d_t = d_t; a.P = P; a.K = K; a.kind = cfg->kind; a.tails = cfg->tails;
  a.left = cfg->left; a.right = cfg->right; a.bottom = cfg->bottom; a.top = cfg->top;
  a.min_w = (float)cfg->min_bin_width; a.min_h = (float)cfg->min_bin_height;
  a.cw = (float)(1.0 - cfg->min_bin_width * K); a.ch = (float)(1.0 - cfg->min_bin_height * K);
  a.w_div = cfg->width_divisor > 0.f ? cfg->width_divisor : 1.f;
  a.h_div = cfg->height_divisor > 0.f ? cfg->height_divisor : 1.f;
  int64_t grid = (a.total + 255) / 256;
  const int64_t cap = (int64_t)fc::device_cu_count() * 8;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(fc::splines_backward_kernel, dim3((unsigned)grid), dim3(256), lds, static_cast<hipStream_t>(stream), a);
  return hipGetLastError();
}
