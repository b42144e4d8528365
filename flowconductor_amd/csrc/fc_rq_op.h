// The rational-quadratic spline evaluation shared by fc_rq_spline.hip (stand-alone bijector kernels)
// and fc_rq_fused.hip (final conditioner layer fused in).  See fc_rq_spline.hip for the reference
// citations (flowcon/transforms/splines/rational_quadratic.py:13-181).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_tile.h"
#include "fc_math.h"

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
  // knot constants of walk_both (static bin counts <= 16), formed in double by rq_finish_params:
  //   sc1 = span c1;  kc_i = lo + span min (i + 1) for i < K / 2,  hi - span min (K - 1 - i) for K / 2 <= i <= K - 2
  float sc1x, sc1y;
  float kcx[15], kcy[15];
};

// host: call after left .. top, min_w / min_h, cw / ch and K are set
inline void rq_finish_params(RQParams& q) {
  const double sx = (double)q.right - (double)q.left, sy = (double)q.top - (double)q.bottom;
  q.sc1x = (float)(sx * (double)q.cw);
  q.sc1y = (float)(sy * (double)q.ch);
  const int K = q.K < 16 ? q.K : 16, H = K / 2;
  for (int i = 0; i < 15; ++i) {
    if (i > K - 2) {
      q.kcx[i] = q.right;
      q.kcy[i] = q.top;
    } else if (i < H) {
      q.kcx[i] = (float)((double)q.left + sx * (double)q.min_w * (double)(i + 1));
      q.kcy[i] = (float)((double)q.bottom + sy * (double)q.min_h * (double)(i + 1));
    } else {
      q.kcx[i] = (float)((double)q.right - sx * (double)q.min_w * (double)(K - 1 - i));
      q.kcy[i] = (float)((double)q.top - sy * (double)q.min_h * (double)(K - 1 - i));
    }
  }
}

// Walk the K bins of one cumulative axis. u -> LDS pointer to K unnormalised values.
// search: idx = last bin whose lower knot <= v (== compare-count - 1 for monotone knots).
// select: take bin `idx`. Returns lower knot and bin size of the chosen bin.
template <int KS, bool kSearch>
__device__ __forceinline__ void walk_axis(const float* __restrict__ u, int K, float inv_scale,
                                          float minb, float c1,
                                          float lo, float hi, float v, int& idx, float& knot_lo,
                                          float& bin_size) {
  const float span = hi - lo;
  if (KS > 0) {
    float t[KS > 0 ? KS : 1];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      t[i] = u[i] * inv_scale;
      m = fmaxf(m, t[i]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      t[i] = exp_lean(t[i] - m);
      sum += t[i];
    }
    const float rs = div_lean(1.f, sum);
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
      const float t = u[i] * inv_scale;
      m = fmaxf(m, t);
    }
    float sum = 0.f;
    for (int i = 0; i < K; ++i) {
      const float t = u[i] * inv_scale;
      sum += exp_lean(t - m);
    }
    const float rs = div_lean(1.f, sum);
    double cum = 0.0;
    float prev = lo;
    int found = kSearch ? 0 : idx;
    float klo = lo, bsz = 0.f;
    for (int i = 0; i < K; ++i) {
      const float t = u[i] * inv_scale;
      const float p = exp_lean(t - m) * rs;
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

typedef float f2 __attribute__((ext_vector_type(2)));

// exp(x) for x <= 0 in the softmax: 2^(x*log2e) without the product-error compensation of exp_lean.
// The relative error grows as |x| * 6e-8, i.e. only on bins whose softmax weight is already small.
__device__ __forceinline__ float exp_softmax(float x) {
  return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

// Both cumulative axes in one pass, widths in .x and heights in .y (packed v_pk_{add,fma}_f32: two lanes of work per
// VALU slot).  The knots are affine in partial sums of the softmax NUMERATORS e_i = exp(u_i - max):
//   knot_{i+1} = lo + span sum_{j<=i} (min + c1 e_j / total) = kc_i + l_i (span c1 / total),  l_i = e_0 + .. + e_i       (i <  K/2)
//              = hi - span sum_{j>i}  (min + c1 e_j / total) = kc_i - r_i (span c1 / total),  r_i = e_{i+1} + .. + e_{K-1} (i >= K/2)
// so a knot pair costs one packed fma instead of normalising, offsetting, accumulating (in double) and scaling each bin;
// summing from the nearer end keeps every partial sum <= K/2 - 1 float additions deep and about half of the total in
// size (error against float64 below the float32 reference's own: tools/probe/fused_accuracy.py).  The bin is searched on
// one axis (kSearchX: widths, forward; else heights, inverse); knots are monotone, so "last bin whose lower knot <= v" is
// tracked by one predicate that selects on both axes.
template <int KS, bool kSearchX>
__device__ __forceinline__ void walk_both(const float* __restrict__ uw, const float* __restrict__ uh,
                                          float inv_scale, const RQParams& q, float v, int& idx,
                                          f2& knot_lo, f2& bin_size) {
  constexpr int H = KS / 2;
  f2 t[KS > 0 ? KS : 1];
  float mx = -INFINITY, my = -INFINITY;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    t[i] = f2{uw[i], uh[i]};
    mx = fmaxf(mx, t[i].x);
    my = fmaxf(my, t[i].y);
  }
  const f2 m = {mx, my};
  const float c2 = inv_scale * 1.4426950408889634f;      // exp(d / wh_div) = exp2(d c2); inv_scale > 0 keeps the maximum
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const f2 d = (t[i] - m) * c2;
    t[i] = f2{__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
  }
  f2 part[KS > 0 ? KS : 1];       // part[i] = l_i (i < H), r_i (H - 1 <= i <= KS - 2; slot H - 1 holds r until the total is formed)
  f2 run = t[0];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    if (i > 0) run += t[i];
    part[i] = run;
  }
  const f2 l_last = run;
  run = t[KS - 1];
#pragma unroll
  for (int i = KS - 2; i >= H; --i) {
    part[i] = run;
    run += t[i];
  }
  const f2 tot = l_last + run;     // run = e_H + .. + e_{K-1}
  const f2 g = f2{q.sc1x, q.sc1y} * f2{div_lean(1.f, tot.x), div_lean(1.f, tot.y)};
  const f2 lo = {q.left, q.bottom}, hi = {q.right, q.top};
  f2 prev = lo, sel_lo = lo, sel_hi = lo;
  int found = 0;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    f2 next = hi;
    if (i < KS - 1) {
      const f2 kc = {q.kcx[i], q.kcy[i]};
      next = i < H ? __builtin_elementwise_fma(part[i], g, kc) : __builtin_elementwise_fma(part[i], -g, kc);
    }
    const bool take = v >= (kSearchX ? prev.x : prev.y);
    sel_lo.x = take ? prev.x : sel_lo.x;
    sel_lo.y = take ? prev.y : sel_lo.y;
    sel_hi.x = take ? next.x : sel_hi.x;
    sel_hi.y = take ? next.y : sel_hi.y;
    found = take ? i : found;
    prev = next;
  }
  idx = found;
  knot_lo = sel_lo;
  bin_size = sel_hi - sel_lo;
}

template <int KS>
struct RQOp {
  static constexpr bool kHasPrepare = false;
  __device__ void prepare(float*, int, int) const {}
  RQParams q;
  float inv_beta; // 1 / softplus beta (the fused kernel multiplies instead of dividing; exact at the default 1)
  float inv_div;  // unnormalised widths/heights are multiplied by 1/wh_div (exact for the usual
                  // power-of-two sqrt(hidden_features); otherwise within 1 ulp of the reference's division)

  __device__ __forceinline__ void eval(const float* __restrict__ prow, int j, int d_t, float x,
                                       float& y, float& lad, uint32_t& err) const {
    const int K = KS > 0 ? KS : q.K;
    const int P = q.tails ? 3 * K - 1 : 3 * K + 1;
    eval_core<false>(prow + j * P, x, y, lad, err);
  }

  // Branch-free evaluation for linear tails: no early return for out-of-interval inputs (evaluated on a
  // clamped copy, selected away at the end), select-form softplus, direction fixed at compile time.  This is
  // the readable form of what the fused kernel executes: fc_rq_fused3_eval.inc is this function written out
  // statement by statement (tools/gen_fused_eval.py) with MFMA hook points in between and the parameters
  // taken from accumulator registers instead of `p`.
  template <bool kInverse>
  __device__ __forceinline__ void eval_tails_straight(const float* __restrict__ p, float x, float& y, float& lad,
                                                      uint32_t& err) const {
    static_assert(KS > 0, "static bin count required");
    const bool inside = (x >= q.left) && (x <= q.right);
    const float xc = inside ? x : q.left;
    int idx;
    f2 klo, bsz;
    walk_both<KS, !kInverse>(p, p + KS, inv_div, q, xc, idx, klo, bsz);
    const float xk = klo.x, yk = klo.y, wk = bsz.x, hk = bsz.y;
    const float* ud = p + 2 * KS;
    const float r0 = ud[idx > 0 ? idx - 1 : 0], r1 = ud[idx < KS - 1 ? idx : KS - 2];
    const float u0 = idx == 0 ? q.tail_const : r0;
    const float u1 = idx == KS - 1 ? q.tail_const : r1;
    const float d0 = q.min_d + softplus_plain(u0, q.beta, inv_beta);
    const float d1 = q.min_d + softplus_plain(u1, q.beta, inv_beta);
    const float delta = div_lean(hk, wk);
    const float dsum = d0 + d1 - 2.f * delta;
    float theta;
    if constexpr (!kInverse) {
      theta = div_lean(xc - xk, wk);
    } else {
      const float r = xc - yk;
      const float qa = r * dsum + hk * (delta - d0);
      const float qb = hk * d0 - r * dsum;
      const float qc = -delta * r;
      const float disc = qb * qb - 4.f * qa * qc;
      if (inside && !(disc >= 0.f)) err |= kErrDiscriminant;
      theta = div_lean(2.f * qc, -qb - sqrt_lean(disc));
    }
    const float t1mt = theta * (1.f - theta);
    const float den = delta + dsum * t1mt;
    const float omt = 1.f - theta;
    const float dnum = (delta * delta) * (d1 * (theta * theta) + 2.f * delta * t1mt + d0 * (omt * omt));
    const float l = __builtin_fmaf(-2.f, __builtin_amdgcn_logf(den), __builtin_amdgcn_logf(dnum)) * 0.6931471805599453f;
    float ys;
    if constexpr (!kInverse) {
      const float num = hk * (delta * (theta * theta) + d0 * t1mt);
      ys = yk + div_lean(num, den);
    } else {
      ys = theta * wk + xk;
    }
    y = inside ? ys : x;
    lad = inside ? (kInverse ? -l : l) : 0.f;
  }

  // p -> the P raw values of one (sample, dim): an LDS pointer, or (kRegs) a register array that
  // must only be indexed statically, so the two derivatives are picked with a select chain.
  template <bool kRegs>
  __device__ __forceinline__ void eval_core(const float* __restrict__ p, float x, float& y, float& lad,
                                            uint32_t& err) const {
    const int K = KS > 0 ? KS : q.K;
#ifdef FC_PROBE_SKIP_EVAL  // tools/ ablation build only: data movement without the spline arithmetic
    y = x + p[0] * 0.f;
    lad = 0.f;
    return;
#endif

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
    if constexpr (KS > 0) {
      f2 klo, bsz;
      if (!q.inverse)
        walk_both<KS, true>(p, p + K, inv_div, q, x, idx, klo, bsz);
      else
        walk_both<KS, false>(p, p + K, inv_div, q, x, idx, klo, bsz);
      xk = klo.x; yk = klo.y; wk = bsz.x; hk = bsz.y;
    } else if (!q.inverse) {
      walk_axis<KS, true>(p, K, inv_div, q.min_w, q.cw, q.left, q.right, x,
                          idx, xk, wk);
      walk_axis<KS, false>(p + K, K, inv_div, q.min_h, q.ch, q.bottom, q.top,
                           x, idx, yk, hk);
    } else {
      walk_axis<KS, true>(p + K, K, inv_div, q.min_h, q.ch, q.bottom, q.top,
                          x, idx, yk, hk);
      walk_axis<KS, false>(p, K, inv_div, q.min_w, q.cw, q.left, q.right, x,
                           idx, xk, wk);
    }

    // derivatives at the two knots of the bin (rational_quadratic.py:33-36, :100-104)
    const float* ud = p + 2 * K;
    float u0, u1;
    if (kRegs && KS > 0) {
      // padded derivative row: linear tails [c, ud_0..ud_{K-2}, c]; none [ud_0..ud_K]
      u0 = q.tails ? q.tail_const : ud[0];
      u1 = q.tails ? q.tail_const : ud[KS];
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const float lo_i = q.tails ? (i == 0 ? q.tail_const : ud[i > 0 ? i - 1 : 0]) : ud[i];
        const float hi_i = q.tails ? (i == KS - 1 ? q.tail_const : ud[i < KS - 1 ? i : 0]) : ud[i + 1];
        u0 = idx == i ? lo_i : u0;
        u1 = idx == i ? hi_i : u1;
      }
    } else if (q.tails) {
      u0 = idx == 0 ? q.tail_const : ud[idx - 1];
      u1 = idx == K - 1 ? q.tail_const : ud[idx];
    } else {
      u0 = ud[idx];
      u1 = ud[idx + 1];
    }
    const float d0 = q.min_d + softplus_plain(u0, q.beta, inv_beta);
    const float d1 = q.min_d + softplus_plain(u1, q.beta, inv_beta);
    const float delta = div_lean(hk, wk);
    const float dsum = d0 + d1 - 2.f * delta;

    float theta;
    if (!q.inverse) {
      theta = div_lean(x - xk, wk);
    } else {
      // rational_quadratic.py:133-146
      const float r = x - yk;
      const float qa = r * dsum + hk * (delta - d0);
      const float qb = hk * d0 - r * dsum;
      const float qc = -delta * r;
      const float disc = qb * qb - 4.f * qa * qc;
      if (!(disc >= 0.f)) err |= kErrDiscriminant;
      theta = div_lean(2.f * qc, -qb - sqrt_lean(disc));
    }
    const float t1mt = theta * (1.f - theta);
    const float den = delta + dsum * t1mt;
    const float omt = 1.f - theta;
    const float dnum = (delta * delta) * (d1 * (theta * theta) + 2.f * delta * t1mt + d0 * (omt * omt));
    const float l = __builtin_fmaf(-2.f, __builtin_amdgcn_logf(den), __builtin_amdgcn_logf(dnum)) * 0.6931471805599453f;
    if (!q.inverse) {
      const float num = hk * (delta * (theta * theta) + d0 * t1mt);
      y = yk + div_lean(num, den);
      lad = l;
    } else {
      y = theta * wk + xk;
      lad = -l;
    }
  }
};

}  // namespace fc
