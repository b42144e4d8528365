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
};

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

// Both cumulative axes in one pass, widths in .x and heights in .y so that the mul/add chain maps to
// packed v_pk_{mul,add}_f32 (2 lanes of work per VALU slot).  The bin is searched on one axis
// (kSearchX: widths, forward; else heights, inverse); knots are monotone, so "last bin whose lower
// knot <= v" is tracked by one predicate that selects on both axes.
template <int KS, bool kSearchX>
__device__ __forceinline__ void walk_both(const float* __restrict__ uw, const float* __restrict__ uh,
                                          float inv_scale, f2 minb, f2 c1, f2 lo, f2 hi, float v, int& idx,
                                          f2& knot_lo, f2& bin_size) {
  f2 t[KS > 0 ? KS : 1];
  float mx = -INFINITY, my = -INFINITY;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    t[i] = f2{uw[i], uh[i]} * inv_scale;
    mx = fmaxf(mx, t[i].x);
    my = fmaxf(my, t[i].y);
  }
  const f2 m = {mx, my};
  f2 sum = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const f2 d = t[i] - m;
    t[i] = f2{exp_softmax(d.x), exp_softmax(d.y)};
    sum += t[i];
  }
  const f2 rs = {div_lean(1.f, sum.x), div_lean(1.f, sum.y)};
  const f2 span = hi - lo;
  double cx = 0.0, cy = 0.0;  // at::cumsum on the CPU accumulates f32 in double
  f2 prev = lo, sel_lo = lo, sel_hi = lo;
  int found = 0;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const f2 p = t[i] * rs;
    const f2 w = minb + c1 * p;
    cx += (double)w.x;
    cy += (double)w.y;
    const f2 cum = {(float)cx, (float)cy};
    const f2 next = (i == KS - 1) ? hi : (span * cum + lo);
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
    const f2 minb = {q.min_w, q.min_h}, c1 = {q.cw, q.ch};
    const f2 lo = {q.left, q.bottom}, hi = {q.right, q.top};
    int idx;
    f2 klo, bsz;
    walk_both<KS, !kInverse>(p, p + KS, inv_div, minb, c1, lo, hi, xc, idx, klo, bsz);
    const float xk = klo.x, yk = klo.y, wk = bsz.x, hk = bsz.y;
    const float* ud = p + 2 * KS;
    const float r0 = ud[idx > 0 ? idx - 1 : 0], r1 = ud[idx < KS - 1 ? idx : KS - 2];
    const float u0 = idx == 0 ? q.tail_const : r0;
    const float u1 = idx == KS - 1 ? q.tail_const : r1;
    const float d0 = q.min_d + softplus_lean_sel(u0, q.beta);
    const float d1 = q.min_d + softplus_lean_sel(u1, q.beta);
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
    const float l = log_lean(dnum) - 2.f * log_lean(den);
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
      const f2 minb = {q.min_w, q.min_h}, c1 = {q.cw, q.ch};
      const f2 lo = {q.left, q.bottom}, hi = {q.right, q.top};
      f2 klo, bsz;
      if (!q.inverse)
        walk_both<KS, true>(p, p + K, inv_div, minb, c1, lo, hi, x, idx, klo, bsz);
      else
        walk_both<KS, false>(p, p + K, inv_div, minb, c1, lo, hi, x, idx, klo, bsz);
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
    const float d0 = q.min_d + softplus_lean(u0, q.beta);
    const float d1 = q.min_d + softplus_lean(u1, q.beta);
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
    const float l = log_lean(dnum) - 2.f * log_lean(den);
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
