// Sum-of-sigmoids monotone bijector (+ extended softplus), forward and numerical inverse, gfx950.
//
// Restates (not copies):
//   flowcon/transforms/adaptive_sigmoids.py:108-142   sum_of_sigmoids / get_params / forward
//   flowcon/transforms/nonlinearities.py:519-552      ExtendedSoftplus
//   flowcon/transforms/no_analytic_inv/base.py:23-83  bracket -> bisection -> 2 Newton steps
// Row layout per dim: [S shift | S log_scale | S raw_softmax | 1 softplus shift] = 3S + 1 raw
// values.  They are turned into derived parameters (10 tanh, 0.1 + 9.9 sigmoid, renormalised
// softmax + 1e-6, softplus + 0.1) once, in place in LDS, so the function evaluations of the inverse (a bracket, a
// safeguarded Newton search: ~8-12 evaluations; the reference spends ~55) read ready values.  The inverse brackets per element (the reference expands one
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
  int iterations;      // cap on the root-search steps (SumOfSigmoids: 50)
  int bisect;          // 1: the reference's plain bisection with `iterations` steps (A/B measurements); 0: safeguarded Newton
  float lim;           // initial bracket half-width (SumOfSigmoids: 120)
  float ratio_mult;    // 1.5
  float offset;        // forward returns z - offset, inverse consumes inputs + offset (AR: 0.5)
  float log_post;      // log_scale_postact (0)

  // (Derived parameters and the forward evaluation on the lean primitives of fc_math.h -- hardware exp2 / log2 / rcp plus
  //  one correction step, 1-2 ulp -- instead of libm: ~2 100 instead of ~15 000 instructions per element at S = 30,
  //  1.4 -> 0.3 ms per 2^18 x 8 elements; the golden fixtures need 0.00 of their float32-noise margin either way.)
  __device__ __forceinline__ void prepare(float* __restrict__ prow, int j, int d_t) const {
    float* p = prow + j * (3 * S + 1);
    for (int k = 0; k < S; ++k) p[k] = tanh_lean(p[k]) * 10.f;
    for (int k = 0; k < S; ++k) p[S + k] = sigmoid_lean(p[S + k]) * 9.9f + 0.1f;
    float m = -INFINITY;
    for (int k = 0; k < S; ++k) m = fmaxf(m, p[2 * S + k]);
    float sum = 0.f;
    for (int k = 0; k < S; ++k) {
      const float e = exp_lean(p[2 * S + k] - m);
      p[2 * S + k] = e;
      sum += e;
    }
    const float rs = div_lean(1.f, sum);
    float tot = 0.f;
    for (int k = 0; k < S; ++k) {
      const float w = p[2 * S + k] * rs + 1e-6f;
      p[2 * S + k] = w;
      tot += w;
    }
    const float scale = div_lean(expf(log_post), tot);
    for (int k = 0; k < S; ++k) p[2 * S + k] = scale * p[2 * S + k];
    p[3 * S] = softplus_lean(p[3 * S], 1.f) + 0.1f;
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

  // value and derivative on the lean primitives (fc_math.h): the root search only needs them to ~1e-6 -- the two closing
  // Newton steps and the returned logabsdet use the reference-order functions below
  __device__ __forceinline__ void value_deriv(const float* __restrict__ p, float x, float& f, float& df) const {
    float acc = 0.f, dacc = 0.f, wsum = 0.f;
    for (int k = 0; k < S; ++k) {
      const float a = p[S + k], w = p[2 * S + k];
      const float pre = a * (x - p[k]);
      const float e = exp_lean(-fabsf(pre));                 // in (0, 1]
      const float r = __builtin_amdgcn_rcpf(1.f + e);
      const float sg = pre >= 0.f ? r : e * r;               // sigmoid(pre)
      acc += w * sg;
      dacc += (w * a) * (e * r * r);                         // sigmoid'(pre) = e / (1 + e)^2
      wsum += w;
    }
    const float sh = p[3 * S];
    const float u = x - sh, v = -(x + sh);
    const float eu = exp_lean(-fabsf(u)), ev = exp_lean(-fabsf(v));
    const float ru = __builtin_amdgcn_rcpf(1.f + eu), rv = __builtin_amdgcn_rcpf(1.f + ev);
    const float spu = fmaxf(u, 0.f) + log_lean(1.f + eu), spv = fmaxf(v, 0.f) + log_lean(1.f + ev);   // softplus
    const float rw = __builtin_amdgcn_rcpf(wsum);
    f = acc * rw + (spu - spv);
    df = dacc * rw + (u >= 0.f ? ru : eu * ru) + (v >= 0.f ? rv : ev * rv);     // + sigmoid(u) + sigmoid(v)
  }

  // value and log-derivative: lj_sos = logsumexp_k(log(w_k a_k) + pre_k - 2 softplus(pre_k)) in ONE pass over the sigmoids
  // (running maximum, the partial sum rescaled when it moves)
  __device__ __forceinline__ void value_lad(const float* __restrict__ p, float x, float& val,
                                            float& lad) const {
    float acc = 0.f, wsum = 0.f, m = -1e30f, se = 0.f;     // (a finite floor: exp_lean(-inf) is not defined)
    for (int k = 0; k < S; ++k) {
      const float a = p[S + k], w = p[2 * S + k];
      const float pre = a * (x - p[k]);
      const float e = exp_lean(-fabsf(pre));                 // in (0, 1]
      const float r = div_lean(1.f, 1.f + e);
      acc += w * (pre >= 0.f ? r : e * r);                   // w sigmoid(pre)
      wsum += w;
      // pre - 2 softplus(pre) = -|pre| - 2 log1p(exp(-|pre|)): log(sigmoid'(pre)), symmetric in pre
      const float lj = log_lean(w * a) - fabsf(pre) - 2.f * log1p_lean_pos(e);
      const float mn = fmaxf(m, lj);
      se = se * exp_lean(m - mn) + exp_lean(lj - mn);
      m = mn;
    }
    const float lj_sos = m + log_lean(se);
    const float sh = p[3 * S];
    const float lj_pos = -logaddexpf(sh, x) + x;
    const float lj_neg = -softplus1(sh + x);
    const float lj_esp = logaddexpf(lj_pos, lj_neg);
    val = div_lean(acc, wsum) + (softplus1(x - sh) - softplus1(-(x + sh)));
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
    // bracket: expand until f(hi) >= z and f(lo) <= z (bounded number of expansions), base.py:40-60 per element
    float hi = lim, lo = -lim, fv, dv;
    for (int it = 0; it < 64; ++it) {
      value_deriv(p, hi, fv, dv);
      if (!(fv < z)) break;
      hi = hi * ratio_mult * fmaxf(z / fv, 1.f);
    }
    hi += 1.f;
    for (int it = 0; it < 64; ++it) {
      value_deriv(p, lo, fv, dv);
      if (!(fv > z)) break;
      lo = lo * ratio_mult * fmaxf(z / fv, 1.f);
    }
    lo -= 1.f;
    // Safeguarded Newton inside the bracket instead of the reference's `iterations` (50) bisection steps
    // (base.py:65-79): every evaluation tightens the bracket by the sign of f - z; the Newton iterate is taken when it
    // falls strictly inside, the midpoint otherwise.  f is smooth and strictly increasing (f' >= the extended
    // softplus' slope > 0), so the iteration converges quadratically: 5-9 evaluations where bisection spends 50.
    // `iterations` stays the cap.
    float xg = fminf(fmaxf(z, lo), hi);          // the extended softplus makes f(x) ~ x: z itself is a good start
    if (bisect) {
      for (int it = 0; it < iterations; ++it) {
        const float mid = (hi + lo) * 0.5f;
        const float fm = value(p, mid);
        if (fm > z) hi = mid;
        else if (fm < z) lo = mid;
        else { hi = mid; lo = mid; }
      }
      xg = (hi + lo) * 0.5f;
    }
    for (int it = 0; it < (bisect ? 0 : iterations); ++it) {
      value_deriv(p, xg, fv, dv);
      const float r = fv - z;
      if (r > 0.f) hi = xg;
      else if (r < 0.f) lo = xg;
      else break;
      float xn = xg - r * __builtin_amdgcn_rcpf(dv);
      if (!(xn > lo && xn < hi)) xn = 0.5f * (lo + hi);
      const bool done = fabsf(xn - xg) <= 2e-7f * fmaxf(1.f, fabsf(xg));
      xg = xn;
      if (done) break;
    }
    // two closing Newton steps, x -= f / (f' + 1e-7)  (base.py:27-33); the returned logabsdet is the forward's
    float v, l;
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      if (bisect) {
        value_lad(p, xg, v, l);
        xg = xg - (v - z) / (expf(l) + 1e-7f);
      } else {
        value_deriv(p, xg, fv, dv);
        xg = xg - (fv - z) / (dv + 1e-7f);
      }
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
  op.iterations = bisection_iterations < 0 ? -bisection_iterations : bisection_iterations;
  op.bisect = bisection_iterations < 0 ? 1 : 0;
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
