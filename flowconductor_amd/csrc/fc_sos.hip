// Sum-of-sigmoids monotone bijector (+ extended softplus), forward and numerical inverse, gfx950.
//
// Restates (not copies):
//   flowcon/transforms/adaptive_sigmoids.py:108-142   sum_of_sigmoids / get_params / forward
//   flowcon/transforms/nonlinearities.py:519-552      ExtendedSoftplus
//   flowcon/transforms/no_analytic_inv/base.py:23-83  bracket -> bisection -> 2 Newton steps
// Row layout per dim: [S shift | S log_scale | S raw_softmax | 1 softplus shift] = 3S + 1 raw
// values.  They are turned into derived parameters (10 tanh, 0.1 + 9.9 sigmoid, renormalised
// softmax + 1e-6, softplus + 0.1) once, in place in LDS, so the ~55 function evaluations of the
// inverse read ready values.  The inverse brackets per element (the reference expands one
// batch-global bracket, base.py:46-60, which makes its result depend on batch composition);
// both converge to the same root.
#include "fc_tile.h"
#include "fc_math.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct SoSOp {
  static constexpr bool kHasPrepare = true;
  int S;
  int inverse;
  int iterations;      // bisection steps (SumOfSigmoids: 50)
  float lim;           // initial bracket half-width (SumOfSigmoids: 120)
  float ratio_mult;    // 1.5
  float offset;        // forward returns z - offset, inverse consumes inputs + offset (AR: 0.5)
  float log_post;      // log_scale_postact (0)

  __device__ __forceinline__ void prepare(float* __restrict__ prow, int j, int d_t) const {
    float* p = prow + j * (3 * S + 1);
    for (int k = 0; k < S; ++k) p[k] = tanhf(p[k]) * 10.f;
    for (int k = 0; k < S; ++k) p[S + k] = sigmoidf(p[S + k]) * 9.9f + 0.1f;
    float m = -INFINITY;
    for (int k = 0; k < S; ++k) m = fmaxf(m, p[2 * S + k]);
    float sum = 0.f;
    for (int k = 0; k < S; ++k) {
      const float e = expf(p[2 * S + k] - m);
      p[2 * S + k] = e;
      sum += e;
    }
    float tot = 0.f;
    for (int k = 0; k < S; ++k) {
      const float w = p[2 * S + k] / sum + 1e-6f;
      p[2 * S + k] = w;
      tot += w;
    }
    const float scale = expf(log_post);
    for (int k = 0; k < S; ++k) p[2 * S + k] = scale * (p[2 * S + k] / tot);
    p[3 * S] = softplus1(p[3 * S]) + 0.1f;
  }

  // value only (bisection)
  __device__ __forceinline__ float value(const float* __restrict__ p, float x) const {
    float acc = 0.f, wsum = 0.f;
    for (int k = 0; k < S; ++k) {
      const float pre = p[S + k] * (x - p[k]);
      acc += p[2 * S + k] * sigmoidf(pre);
      wsum += p[2 * S + k];
    }
    const float sh = p[3 * S];
    return acc / wsum + (softplus1(x - sh) - softplus1(-(x + sh)));
  }

  // value and log-derivative
  __device__ __forceinline__ void value_lad(const float* __restrict__ p, float x, float& val,
                                            float& lad) const {
    float acc = 0.f, wsum = 0.f, m = -INFINITY;
    for (int k = 0; k < S; ++k) {
      const float pre = p[S + k] * (x - p[k]);
      const float w = p[2 * S + k];
      acc += w * sigmoidf(pre);
      wsum += w;
      const float lj = logf(w) + logf(p[S + k]) + (pre - 2.f * softplus1(pre));
      m = fmaxf(m, lj);
    }
    float se = 0.f;
    for (int k = 0; k < S; ++k) {
      const float pre = p[S + k] * (x - p[k]);
      const float lj = logf(p[2 * S + k]) + logf(p[S + k]) + (pre - 2.f * softplus1(pre));
      se += expf(lj - m);
    }
    const float lj_sos = m + logf(se);
    const float sh = p[3 * S];
    const float lj_pos = -logaddexpf(sh, x) + x;
    const float lj_neg = -softplus1(sh + x);
    const float lj_esp = logaddexpf(lj_pos, lj_neg);
    val = acc / wsum + (softplus1(x - sh) - softplus1(-(x + sh)));
    lad = logaddexpf(lj_sos, lj_esp);
  }

  __device__ __forceinline__ void eval(const float* __restrict__ prow, int j, int d_t, float x,
                                       float& y, float& lad, uint32_t& err) const {
    const float* p = prow + j * (3 * S + 1);
    if (!inverse) {
      float v, l;
      value_lad(p, x, v, l);
      y = v - offset;
      lad = l;
      return;
    }
    const float z = x + offset;
    // bracket: expand until f(hi) >= z and f(lo) <= z (bounded number of expansions)
    float hi = lim, lo = -lim;
    for (int it = 0; it < 64; ++it) {
      const float fh = value(p, hi);
      if (!(fh < z)) break;
      hi = hi * ratio_mult * fmaxf(z / fh, 1.f);
    }
    hi += 1.f;
    for (int it = 0; it < 64; ++it) {
      const float fl = value(p, lo);
      if (!(fl > z)) break;
      lo = lo * ratio_mult * fmaxf(z / fl, 1.f);
    }
    lo -= 1.f;
    for (int it = 0; it < iterations; ++it) {
      const float mid = (hi + lo) * 0.5f;
      const float fm = value(p, mid);
      if (fm > z) hi = mid;
      else if (fm < z) lo = mid;
      else { hi = mid; lo = mid; }
    }
    float xg = (hi + lo) * 0.5f;
    // two Newton steps, x -= f / (f' + 1e-7)  (base.py:27-33); f' = exp(log-derivative)
    float v, l;
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      value_lad(p, xg, v, l);
      xg = xg - (v - z) / (expf(l) + 1e-7f);
    }
    value_lad(p, xg, v, l);
    if (!isfinite(xg)) err |= kErrNonFinite;
    y = xg;
    lad = -l;
  }
};

}  // namespace fc

extern "C" int fc_sum_of_sigmoids(const float* x, float* y, const float* params, const int32_t* cols,
                                  float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t,
                                  int32_t n_sigmoids, int32_t inverse, int32_t bisection_iterations,
                                  float bisection_lim, float offset, float log_scale_postact,
                                  int32_t shared_params, int32_t lad_mode, void* stream) {
  if (n < 0 || d <= 0 || d_t <= 0 || d_t > d || n_sigmoids <= 0) return hipErrorInvalidValue;
  if (n > 0 && (!x || !y || !params)) return hipErrorInvalidValue;
  fc::SoSOp op;
  op.S = n_sigmoids;
  op.inverse = inverse;
  op.iterations = bisection_iterations;
  op.lim = bisection_lim;
  op.ratio_mult = 1.5f;
  op.offset = offset;
  op.log_post = log_scale_postact;
  fc::TileArgs a{};
  a.x = x; a.y = y; a.params = params; a.cols = cols; a.logabsdet = logabsdet; a.err = err_flag;
  a.N = n; a.D = d; a.d_t = d_t;
  a.rowlen = d_t * (3 * n_sigmoids + 1);
  a.shared_params = shared_params;
  a.lad_mode = lad_mode;
  return fc::launch_tile(op, a, static_cast<hipStream_t>(stream));
}
