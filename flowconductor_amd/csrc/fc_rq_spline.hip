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
#include "../../include/flowcon_hip.h"

namespace fc {

struct RQParams {
  int K;
  int tails;        // 0: none (domain [left,right] x [bottom,top]), 1: linear
  int inverse;
  float left, right, bottom, top;
  float min_w, min_h, min_d;
  float cw, ch;     // (float)(1 - min_w*K), (float)(1 - min_h*K), evaluated in double on the host
  float wh_div;     // unnormalised widths/heights are divided by this (coupling.py:554-559); 1 = off
  float beta;       // softplus beta: 1, or ln2/(1-min_d) with enable_identity_init
  float tail_const; // (float)log(exp(1 - min_d) - 1): padded end derivatives for linear tails
};

// Walk the K bins of one cumulative axis. u -> LDS pointer to K unnormalised values.
// search: idx = last bin whose lower knot <= v (== compare-count - 1 for monotone knots).
// select: take bin `idx`. Returns lower knot and bin size of the chosen bin.
template <int KS, bool kSearch>
__device__ __forceinline__ void walk_axis(const float* __restrict__ u, int K, float inv_scale,
                                          bool scale_by_mul, float scale_div, float minb, float c1,
                                          float lo, float hi, float v, int& idx, float& knot_lo,
                                          float& bin_size) {
  const float span = hi - lo;
  if (KS > 0) {
    float t[KS > 0 ? KS : 1];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      t[i] = scale_by_mul ? u[i] * inv_scale : u[i] / scale_div;
      m = fmaxf(m, t[i]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      t[i] = expf(t[i] - m);
      sum += t[i];
    }
    const float rs = 1.f / sum;
    double cum = 0.0;  // at::cumsum on the CPU accumulates f32 in double
    float prev = lo;
    int found = kSearch ? 0 : idx;
    float klo = lo, bsz = 0.f;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const float p = t[i] * rs;
      const float w = minb + c1 * p;
      cum += (double)w;
      const float next = (i == KS - 1) ? hi : (span * (float)cum + lo);
      const bool take = kSearch ? (v >= prev) : (i == idx);
      if (take) {
        found = i;
        klo = prev;
        bsz = next - prev;
      }
      prev = next;
    }
    idx = found;
    knot_lo = klo;
    bin_size = bsz;
  } else {
    float m = -INFINITY;
    for (int i = 0; i < K; ++i) {
      const float t = scale_by_mul ? u[i] * inv_scale : u[i] / scale_div;
      m = fmaxf(m, t);
    }
    float sum = 0.f;
    for (int i = 0; i < K; ++i) {
      const float t = scale_by_mul ? u[i] * inv_scale : u[i] / scale_div;
      sum += expf(t - m);
    }
    const float rs = 1.f / sum;
    double cum = 0.0;
    float prev = lo;
    int found = kSearch ? 0 : idx;
    float klo = lo, bsz = 0.f;
    for (int i = 0; i < K; ++i) {
      const float t = scale_by_mul ? u[i] * inv_scale : u[i] / scale_div;
      const float p = expf(t - m) * rs;
      const float w = minb + c1 * p;
      cum += (double)w;
      const float next = (i == K - 1) ? hi : (span * (float)cum + lo);
      const bool take = kSearch ? (v >= prev) : (i == idx);
      if (take) {
        found = i;
        klo = prev;
        bsz = next - prev;
      }
      prev = next;
    }
    idx = found;
    knot_lo = klo;
    bin_size = bsz;
  }
}

template <int KS>
struct RQOp {
  static constexpr bool kHasPrepare = false;
  __device__ void prepare(float*, int, int) const {}
  RQParams q;
  float inv_div;
  bool mul_exact;

  __device__ __forceinline__ void eval(const float* __restrict__ prow, int j, int d_t, float x,
                                       float& y, float& lad, uint32_t& err) const {
    const int K = KS > 0 ? KS : q.K;
    const int P = q.tails ? 3 * K - 1 : 3 * K + 1;
    const float* p = prow + j * P;

    // rational_quadratic.py:26-38 / :81-82
    const bool inside = (x >= q.left) && (x <= q.right);
    if (!inside) {
      y = x;
      lad = 0.f;
      if (!q.tails) err |= kErrOutsideDomain;
      return;
    }

    int idx = 0;
    float xk, wk, yk, hk;
    if (!q.inverse) {
      walk_axis<KS, true>(p, K, inv_div, mul_exact, q.wh_div, q.min_w, q.cw, q.left, q.right, x,
                          idx, xk, wk);
      walk_axis<KS, false>(p + K, K, inv_div, mul_exact, q.wh_div, q.min_h, q.ch, q.bottom, q.top,
                           x, idx, yk, hk);
    } else {
      walk_axis<KS, true>(p + K, K, inv_div, mul_exact, q.wh_div, q.min_h, q.ch, q.bottom, q.top,
                          x, idx, yk, hk);
      walk_axis<KS, false>(p, K, inv_div, mul_exact, q.wh_div, q.min_w, q.cw, q.left, q.right, x,
                           idx, xk, wk);
    }

    // derivatives at the two knots of the bin (rational_quadratic.py:33-36, :100-104)
    const float* ud = p + 2 * K;
    float u0, u1;
    if (q.tails) {
      u0 = idx == 0 ? q.tail_const : ud[idx - 1];
      u1 = idx == K - 1 ? q.tail_const : ud[idx];
    } else {
      u0 = ud[idx];
      u1 = ud[idx + 1];
    }
    const float d0 = q.min_d + softplus_b(u0, q.beta);
    const float d1 = q.min_d + softplus_b(u1, q.beta);
    const float delta = hk / wk;
    const float dsum = d0 + d1 - 2.f * delta;

    float theta;
    if (!q.inverse) {
      theta = (x - xk) / wk;
    } else {
      // rational_quadratic.py:133-146
      const float r = x - yk;
      const float qa = r * dsum + hk * (delta - d0);
      const float qb = hk * d0 - r * dsum;
      const float qc = -delta * r;
      const float disc = qb * qb - 4.f * qa * qc;
      if (!(disc >= 0.f)) err |= kErrDiscriminant;
      theta = (2.f * qc) / (-qb - sqrtf(disc));
    }
    const float t1mt = theta * (1.f - theta);
    const float den = delta + dsum * t1mt;
    const float omt = 1.f - theta;
    const float dnum = (delta * delta) * (d1 * (theta * theta) + 2.f * delta * t1mt + d0 * (omt * omt));
    const float l = logf(dnum) - 2.f * logf(den);
    if (!q.inverse) {
      const float num = hk * (delta * (theta * theta) + d0 * t1mt);
      y = yk + num / den;
      lad = l;
    } else {
      y = theta * wk + xk;
      lad = -l;
    }
  }
};

template <int KS>
static hipError_t launch_rq(const RQParams& q, const TileArgs& a, hipStream_t stream) {
  RQOp<KS> op;
  op.q = q;
  op.inv_div = 1.f / q.wh_div;
  // x / d == x * (1/d) bit-for-bit only when d is a power of two
  int ex;
  op.mul_exact = frexpf(q.wh_div, &ex) == 0.5f;
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
    case 4: return fc::launch_rq<4>(q, a, s);
    case 5: return fc::launch_rq<5>(q, a, s);
    case 8: return fc::launch_rq<8>(q, a, s);
    case 10: return fc::launch_rq<10>(q, a, s);
    case 16: return fc::launch_rq<16>(q, a, s);
    default: return fc::launch_rq<0>(q, a, s);
  }
}
