// Piecewise linear / quadratic / cubic spline bijectors (the RQ spline's siblings), gfx950.
//
// Restates (not copies):
//   flowcon/transforms/splines/linear.py:38-105      softmax pdf, uniform bins, cdf by cumsum
//   flowcon/transforms/splines/quadratic.py:55-159   trapezoid-normalised heights, quadratic segments
//   flowcon/transforms/splines/cubic.py:63-267       Steffen-style monotone cubic, Blinn's inverse
// and their `unconstrained_*` wrappers (identity outside [-B, B]).
// Same tile machinery as the RQ kernel; bins are walked with runtime-K loops over LDS.  Each
// op's `prepare` turns the raw conditioner outputs of one (sample, dim) into widths / heights
// in place in LDS, once, so the evaluation passes read ready values.
#include "fc_tile.h"
#include "fc_math.h"
#include "../../include/flowcon_hip.h"

namespace fc {

// torch.clamp(x, 0, 1) as ATen computes it: a NaN (e.g. the square root of a discriminant that rounding pushed
// below zero, splines/quadratic.py:139) stays a NaN instead of turning into 0 the way fminf / fmaxf would make it
__device__ __forceinline__ float clamp01(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }


struct SplineParams {
  int K;
  int tails;       // 0 none, 1 linear
  int inverse;
  float left, right, bottom, top;
  float min_w, min_h;   // float casts of the python floats
  float cw, ch;         // (float)(1 - min_w*K), (float)(1 - min_h*K)
  float w_div, h_div;   // divisors applied to unnormalised widths / heights (1 = off)
  float eps;            // cubic: root-in-bin slack (1e-5)
  float quad_thresh;    // cubic: |a| below this -> quadratic fallback (1e-3)
};

// shared: domain gate + normalisation of the input to [0, 1]
__device__ __forceinline__ bool spline_gate(const SplineParams& q, float x, float& xn, float& y, float& lad,
                                            uint32_t& err) {
  const bool inside = (x >= q.left) && (x <= q.right);
  if (!inside) {
    y = x;
    lad = 0.f;
    if (!q.tails) err |= kErrOutsideDomain;
    return false;
  }
  xn = q.inverse ? (x - q.bottom) / (q.top - q.bottom) : (x - q.left) / (q.right - q.left);
  return true;
}

__device__ __forceinline__ float spline_out(const SplineParams& q, float v) {
  return q.inverse ? v * (q.right - q.left) + q.left : v * (q.top - q.bottom) + q.bottom;
}

// softmax of K LDS values in place: p[i] = floor + c1 * softmax_i   (floor = 0, c1 = 1: plain softmax)
__device__ __forceinline__ void softmax_inplace(float* __restrict__ p, int K, float div, float floor_v,
                                                float c1) {
  float m = -INFINITY;
  for (int i = 0; i < K; ++i) {
    const float t = p[i] / div;
    p[i] = t;
    m = fmaxf(m, t);
  }
  float sum = 0.f;
  for (int i = 0; i < K; ++i) {
    const float e = expf(p[i] - m);
    p[i] = e;
    sum += e;
  }
  const float rs = 1.f / sum;
  for (int i = 0; i < K; ++i) p[i] = floor_v + c1 * (p[i] * rs);
}

// ---- linear -----------------------------------------------------------------------------------------

// torch.linspace(0, 1, K + 1)[i] in float32 (symmetric evaluation used by ATen)
__device__ __forceinline__ float linspace01(int i, int K) {
  const float step = 1.f / (float)K;
  const int steps = K + 1;
  return i < steps / 2 ? step * (float)i : 1.f - step * (float)(steps - 1 - i);
}

struct LinearSplineOp {
  static constexpr bool kHasPrepare = true;
  SplineParams q;

  __device__ __forceinline__ void prepare(float* __restrict__ prow, int j, int d_t) const {
    softmax_inplace(prow + j * q.K, q.K, 1.f, 0.f, 1.f);  // pdf
  }

  __device__ __forceinline__ void eval(const float* __restrict__ prow, int j, int d_t, float x, float& y,
                                       float& lad, uint32_t& err) const {
    const int K = q.K;
    const float* pdf = prow + j * K;
    float xn;
    if (!spline_gate(q, x, xn, y, lad, err)) return;
    if (!q.inverse) {
      const float bin_pos = xn * (float)K;
      int idx = (int)floorf(bin_pos);
      if (idx >= K) idx = K - 1;
      if (idx < 0) idx = 0;
      const float alpha = bin_pos - (float)idx;
      double cum = 0.0;
      for (int i = 0; i < idx; ++i) cum += (double)pdf[i];
      const float p = pdf[idx];
      float out = (float)cum + alpha * p;
      out = clamp01(out);
      lad = logf(p) - logf(1.f / (float)K);
      y = spline_out(q, out);
    } else {
      // cdf knots: 0, c_0, ..., c_{K-2}, 1 (+1e-6 after the reference's in-place searchsorted nudge)
      double cum = 0.0;
      float lo = 0.f, knot_lo = 0.f, knot_hi = 0.f;
      int idx = 0;
      for (int i = 0; i < K; ++i) {
        cum += (double)pdf[i];
        const float hi = (i == K - 1) ? (1.0f + 1e-6f) : (float)cum;
        if (xn >= lo) {
          idx = i;
          knot_lo = lo;
          knot_hi = hi;
        }
        lo = hi;
      }
      const float b0 = linspace01(idx, K), b1 = linspace01(idx + 1, K);
      const float slope = (knot_hi - knot_lo) / (b1 - b0);
      const float offset = knot_hi - slope * b1;
      float out = (xn - offset) / slope;
      out = clamp01(out);
      lad = -logf(slope);
      y = spline_out(q, out);
    }
  }
};

// ---- quadratic --------------------------------------------------------------------------------------

struct QuadraticSplineOp {
  static constexpr bool kHasPrepare = true;
  SplineParams q;

  __device__ __forceinline__ int nh() const { return q.tails ? q.K - 1 : q.K + 1; }

  __device__ __forceinline__ void prepare(float* __restrict__ prow, int j, int d_t) const {
    float* p = prow + j * (q.K + nh());
    softmax_inplace(p, q.K, q.w_div, q.min_w, q.cw);  // widths
    float* h = p + q.K;
    for (int i = 0; i < nh(); ++i) h[i] = softplus1(h[i] / q.h_div) + 1e-3f;  // quadratic.py:85
  }

  // un-normalised height at knot i (0..K): boundary knots get `edge` when only K-1 were given
  __device__ __forceinline__ float knot_h(const float* __restrict__ h, int i, float edge) const {
    if (!q.tails) return h[i];
    return (i == 0 || i == q.K) ? edge : h[i - 1];
  }

  __device__ __forceinline__ void eval(const float* __restrict__ prow, int j, int d_t, float x, float& y,
                                       float& lad, uint32_t& err) const {
    const int K = q.K;
    const float* w = prow + j * (K + nh());
    const float* h = w + K;
    float xn;
    if (!spline_gate(q, x, xn, y, lad, err)) return;

    float edge = 0.f;
    if (q.tails) {
      // boundary heights such that the normalised boundary density is 1 (quadratic.py:87-102)
      const float first_w = 0.5f * w[0], last_w = 0.5f * w[K - 1];
      float mid = 0.f;
      for (int i = 0; i + 1 < K - 1; ++i) mid += ((h[i] + h[i + 1]) / 2.f) * w[i + 1];
      const float numer = 0.5f * first_w * h[0] + 0.5f * last_w * h[K - 2] + mid;
      edge = numer / (1.f - 0.5f * first_w - 0.5f * last_w);
    }
    float area = 0.f;
    for (int i = 0; i < K; ++i) area += ((knot_h(h, i, edge) + knot_h(h, i + 1, edge)) / 2.f) * w[i];
    const float oh = 1.f - q.min_h;

    // walk: cumulative cdf / locations, last entries pinned to 1 (+1e-6 nudge on the searched axis)
    double cum_cdf = 0.0, cum_loc = 0.0;
    float lo_cdf = 0.f, lo_loc = 0.f;
    float hl_prev = q.min_h + oh * (knot_h(h, 0, edge) / area);
    int idx = 0;
    float loc = 0.f, wk = w[0], lc = 0.f, hl = hl_prev, hr = hl_prev;
    for (int i = 0; i < K; ++i) {
      const float hr_i = q.min_h + oh * (knot_h(h, i + 1, edge) / area);
      cum_cdf += (double)(((hl_prev + hr_i) / 2.f) * w[i]);
      cum_loc += (double)w[i];
      const float hi_cdf = (i == K - 1) ? 1.f : (float)cum_cdf;
      const float hi_loc = (i == K - 1) ? 1.f : (float)cum_loc;
      const bool take = q.inverse ? (xn >= lo_cdf) : (xn >= lo_loc);
      if (take) {
        idx = i;
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
    (void)idx;
    const float a = 0.5f * (hr - hl) * wk;
    const float b = hl * wk;
    float out;
    if (q.inverse) {
      const float c_ = lc - xn;
      const float alpha = (-b + sqrtf(b * b - 4.f * a * c_)) / (2.f * a);
      out = alpha * wk + loc;
      out = clamp01(out);
      lad = -logf(alpha * (hr - hl) + hl);
    } else {
      const float alpha = (xn - loc) / wk;
      out = a * (alpha * alpha) + b * alpha + lc;
      out = clamp01(out);
      lad = logf(alpha * (hr - hl) + hl);
    }
    y = spline_out(q, out);
  }
};

// ---- cubic --------------------------------------------------------------------------------------------

__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

__device__ __forceinline__ float cbrt_ref(float v) {  // torchutils.cbrt: sign * exp(log|v| / 3)
  return sgn(v) * expf(logf(fabsf(v)) / 3.f);
}

struct CubicSplineOp {
  static constexpr bool kHasPrepare = true;
  SplineParams q;

  __device__ __forceinline__ void prepare(float* __restrict__ prow, int j, int d_t) const {
    float* p = prow + j * (2 * q.K + 2);
    softmax_inplace(p, q.K, q.w_div, q.min_w, q.cw);
    softmax_inplace(p + q.K, q.K, q.h_div, q.min_h, q.ch);
  }

  // interior knot derivative between bins i and i+1 (cubic.py:113-131)
  __device__ __forceinline__ float interior(const float* __restrict__ w, const float* __restrict__ h,
                                            int i) const {
    const float s0 = h[i] / w[i], s1 = h[i + 1] / w[i + 1];
    const float m1 = fminf(fabsf(s0), fabsf(s1));
    const float m2 = 0.5f * (w[i + 1] * s0 + w[i] * s1) / (w[i] + w[i + 1]);
    return fminf(m1, m2) * (sgn(s0) + sgn(s1));
  }

  __device__ __forceinline__ void eval(const float* __restrict__ prow, int j, int d_t, float x, float& y,
                                       float& lad, uint32_t& err) const {
    const int K = q.K;
    const float* w = prow + j * (2 * K + 2);
    const float* h = w + K;
    float xn;
    if (!spline_gate(q, x, xn, y, lad, err)) return;

    double cw = 0.0, chh = 0.0;
    float lo_w = 0.f, lo_h = 0.f;
    int idx = 0;
    float left_w = 0.f, right_w = 1.f, dco = 0.f;
    for (int i = 0; i < K; ++i) {
      cw += (double)w[i];
      chh += (double)h[i];
      const float hi_w = (i == K - 1) ? 1.f : (float)cw;
      const float hi_h = (i == K - 1) ? 1.f : (float)chh;
      const bool take = q.inverse ? (xn >= lo_h) : (xn >= lo_w);
      if (take) {
        idx = i;
        left_w = lo_w;
        right_w = hi_w;
        dco = lo_h;
      }
      lo_w = hi_w;
      lo_h = hi_h;
    }
    // (right_w is only read by the inverse, where the searched -- and nudged -- axis is cumheights,
    //  so cumwidths[idx + 1] is the un-nudged knot, cubic.py:139-151)

    const float wk = w[idx];
    const float s = h[idx] / wk;
    const float dl = idx == 0 ? sigmoidf(w[2 * K]) * 3.f * (h[0] / w[0]) : interior(w, h, idx - 1);
    const float dr = idx == K - 1 ? sigmoidf(w[2 * K + 1]) * 3.f * (h[K - 1] / w[K - 1]) : interior(w, h, idx);
    const float ca = (dl + dr - 2.f * s) / (wk * wk);
    const float cb = (3.f * s - 2.f * dl - dr) / wk;
    const float cc = dl;

    float out;
    if (!q.inverse) {
      const float t = xn - left_w;
      out = ca * (t * t * t) + cb * (t * t) + cc * t + dco;
      lad = logf(3.f * ca * (t * t) + 2.f * cb * t + cc);
    } else {
      // Blinn's closed form (cubic.py:152-244), evaluated without its two cancellations.  The reference's float32
      // sequence is correct in exact arithmetic but loses the answer in two places (measured against its own float64
      // evaluation: max error 1e-3 .. 8e-3 on random parameters, tools/probe/cubic_inverse_accuracy.py):
      //  * one real root: q^3 = (-dep1 - sqrt(-disc)) / 2 cancels to a few ulps when |delta_1|^3 << dep1^2 (an almost
      //    quadratic bin just above the |a| < 1e-3 fallback) and the cube root amplifies what is left by ~1e3.  Here the
      //    cube of larger magnitude is taken by the sum that does not cancel and the other from p q = -delta_1.
      //  * a -> 0 fallback: (-b + sqrt(b^2 - 4 a c)) / (2 a) cancels (and is 0 / 0 at a = 0); 2 c / (-b - sqrt(..)) is
      //    the same root (b = left knot derivative > 0, c <= 0 inside the bin).
      // Two Newton steps on the bin's cubic then remove the rounding of the closed form itself; a step is taken only
      // if it stays inside the bin (+- eps) with a positive slope.  The a -> 0 fallback is NOT polished: there the
      // reference DEFINES the inverse through the quadratic part alone, and that definition is kept.
      // Against float64 the result is never further off than the reference's float32 result (same probe).
      const float b_ = (cb / ca) / 3.f;
      const float c_ = (cc / ca) / 3.f;
      const float d_ = (dco - xn) / ca;
      const float delta_1 = -(b_ * b_) + c_;
      const float delta_2 = -c_ * b_ + d_;
      const float delta_3 = b_ * d_ - c_ * c_;
      const float disc = 4.f * delta_1 * delta_3 - delta_2 * delta_2;
      const float dep1 = -2.f * b_ * delta_1 + delta_2;
      const float dep2 = delta_1;
      if (disc >= 0.f) {
        float theta = atan2f(sqrtf(disc), -dep1);
        theta /= 3.f;
        const float c1 = cosf(theta), s1 = sinf(theta);
        const float k3 = 0.5f * 1.7320508075688772f;  // 0.5 * math.sqrt(3)
        float r1 = c1, r2 = -0.5f * c1 - k3 * s1, r3 = -0.5f * c1 + k3 * s1;
        const float scale = 2.f * sqrtf(-dep2);
        const float shift = -b_ + left_w;
        r1 = r1 * scale + shift;
        r2 = r2 * scale + shift;
        r3 = r3 * scale + shift;
        const bool m1 = ((left_w - q.eps) < r1) && (r1 < (right_w + q.eps));
        const bool m2 = ((left_w - q.eps) < r2) && (r2 < (right_w + q.eps));
        const bool m3 = ((left_w - q.eps) < r3) && (r3 < (right_w + q.eps));
        // argsort(masks, descending)[0]: first root whose mask is set (stable), else the first
        out = m1 ? r1 : (m2 ? r2 : (m3 ? r3 : r1));
      } else {
        const float sq = sqrtf(-disc);
        const float big = (-dep1 + (dep1 <= 0.f ? sq : -sq)) / 2.f;
        const float p = cbrt_ref(big);
        const float qq = p != 0.f ? -dep2 / p : 0.f;
        out = (p + qq) - b_ + left_w;
      }
      if (fabsf(ca) < q.quad_thresh) {
        const float a2 = cb, b2 = cc, c2 = dco - xn;
        const float alpha = (2.f * c2) / (-b2 - sqrtf(b2 * b2 - 4.f * a2 * c2));
        out = alpha + left_w;
      } else {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const float t = out - left_w;
          const float fv = ((ca * t + cb) * t + cc) * t + (dco - xn);
          const float fp = (3.f * ca * t + 2.f * cb) * t + cc;
          const float nxt = out - fv / fp;
          // (comparisons are false for NaN: a failed step keeps the closed-form root)
          if (fp > 0.f && nxt > left_w - q.eps && nxt < right_w + q.eps) out = nxt;
        }
      }
      const float t = out - left_w;
      lad = -logf(3.f * ca * (t * t) + 2.f * cb * t + cc);
    }
    y = spline_out(q, out);
  }
};

}  // namespace fc

extern "C" int fc_piecewise_spline(const float* x, float* y, const float* params, const int32_t* cols,
                                   float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t,
                                   int32_t shared_params, int32_t lad_mode, const fc_spline_config* cfg,
                                   void* stream) {
  if (!cfg || n < 0 || d <= 0 || d_t <= 0 || d_t > d || cfg->num_bins <= 0) return hipErrorInvalidValue;
  if (n > 0 && (!x || !y || !params)) return hipErrorInvalidValue;
  fc::SplineParams q;
  q.K = cfg->num_bins;
  q.tails = cfg->tails;
  q.inverse = cfg->inverse;
  q.left = cfg->left; q.right = cfg->right; q.bottom = cfg->bottom; q.top = cfg->top;
  q.min_w = (float)cfg->min_bin_width;
  q.min_h = (float)cfg->min_bin_height;
  q.cw = (float)(1.0 - cfg->min_bin_width * q.K);
  q.ch = (float)(1.0 - cfg->min_bin_height * q.K);
  q.w_div = cfg->width_divisor > 0.f ? cfg->width_divisor : 1.f;
  q.h_div = cfg->height_divisor > 0.f ? cfg->height_divisor : 1.f;
  q.eps = cfg->cubic_eps;
  q.quad_thresh = cfg->cubic_quadratic_threshold;

  fc::TileArgs a{};
  a.x = x; a.y = y; a.params = params; a.cols = cols; a.logabsdet = logabsdet; a.err = err_flag;
  a.N = n; a.D = d; a.d_t = d_t;
  a.shared_params = shared_params;
  a.lad_mode = lad_mode;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (cfg->kind) {
    case FC_SPLINE_LINEAR: {
      a.rowlen = d_t * q.K;
      return fc::launch_tile(fc::LinearSplineOp{q}, a, s);
    }
    case FC_SPLINE_QUADRATIC: {
      a.rowlen = d_t * (q.tails ? 2 * q.K - 1 : 2 * q.K + 1);
      return fc::launch_tile(fc::QuadraticSplineOp{q}, a, s);
    }
    case FC_SPLINE_CUBIC: {
      a.rowlen = d_t * (2 * q.K + 2);
      return fc::launch_tile(fc::CubicSplineOp{q}, a, s);
    }
    default:
      return hipErrorInvalidValue;
  }
}
