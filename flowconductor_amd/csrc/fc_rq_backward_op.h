// Element-level backward of the rational-quadratic spline (forward direction): shared by fc_rq_backward.hip (parameters
// read from HBM) and fc_rq_fused_backward.h (parameters recomputed on the matrix cores).  See fc_rq_backward.hip for the
// derivation and the reference citations (flowcon/transforms/splines/rational_quadratic.py:13-181).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_math.h"
#include "fc_rq_op.h"

namespace fc {

constexpr int kMaxBinsBwd = 32;

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
  double cum = 0.0;   // ATen's CPU cumsum accumulates f32 in double (as the run-time-K forward walk does)
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

// The same element for compile-time K and tail mode on REGISTER parameters (fc_rq_fused_backward.h), arranged for
// the matrix-core kernels' tight register budget and VALU-bound loops:
//   * both cumulative axes walked together in packed f32 pairs (.x widths, .y heights), softmax through v_exp with
//     the log2(e) factor folded in (as the forward's walk_both);
//   * plain f32 running sums for the knots (the static-K forward forms its knots from two-sided float partial sums, the run-time-K walk accumulates in double like ATen's CPU cumsum; the gradient does
//     not need the knots bit for bit -- an input within 1e-7 of a knot may differentiate the neighbouring bin, whose
//     value and slope agree there);
//   * softplus and its slope (the sigmoid) of a knot derivative from ONE exponential;
//   * the chain to the 2K softmax logits as  g_m = c p_m (S_m - C)  with S_m in {A, g_hi, 0} by bin position.
template <int K, bool kTails>
__device__ __forceinline__ void rq_backward_element_fast(const RQParams& q, float inv_div, const float* __restrict__ u,
                                                         float x, float gy, float gl, float& gx, float* __restrict__ gp) {
  constexpr int P = kTails ? 3 * K - 1 : 3 * K + 1;
  const bool inside = (x >= q.left) && (x <= q.right);
  if (!inside) {
    gx = gy;
#pragma unroll
    for (int i = 0; i < P; ++i) gp[i] = 0.f;
    return;
  }
  f2 p[K];
  {
    float mx = -INFINITY, my = -INFINITY;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      p[i] = f2{u[i], u[K + i]} * inv_div;
      mx = fmaxf(mx, p[i].x);
      my = fmaxf(my, p[i].y);
    }
    const f2 m = {mx, my};
    f2 sum = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const f2 d = (p[i] - m) * f2{1.4426950408889634f, 1.4426950408889634f};
      p[i] = f2{__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
      sum += p[i];
    }
    const f2 rs = {div_lean(1.f, sum.x), div_lean(1.f, sum.y)};
#pragma unroll
    for (int i = 0; i < K; ++i) p[i] = p[i] * rs;
  }
  // knots around the bin of x (searched on the width axis) and the prefix sums of the probabilities below them
  const f2 minb = {q.min_w, q.min_h}, c1 = {q.cw, q.ch}, lo = {q.left, q.bottom}, hi = {q.right, q.top};
  const f2 span = hi - lo;
  f2 cum = {0.f, 0.f}, prev = lo, prevp = {0.f, 0.f};
  f2 k_lo = lo, k_hi = lo, p_lo = {0.f, 0.f}, p_hi = {0.f, 0.f};
  int idx = 0;
  {
    f2 psum = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < K; ++i) {
      cum += minb + c1 * p[i];
      psum += p[i];
      const f2 next = (i == K - 1) ? hi : (span * cum + lo);
      const bool take = x >= prev.x;
      idx = take ? i : idx;
      k_lo.x = take ? prev.x : k_lo.x;   k_lo.y = take ? prev.y : k_lo.y;
      k_hi.x = take ? next.x : k_hi.x;   k_hi.y = take ? next.y : k_hi.y;
      p_lo.x = take ? prevp.x : p_lo.x;  p_lo.y = take ? prevp.y : p_lo.y;
      p_hi.x = take ? psum.x : p_hi.x;   p_hi.y = take ? psum.y : p_hi.y;
      prev = next;
      prevp = psum;
    }
  }
  // knot derivatives d = min_d + softplus(u, beta) and their slopes sigmoid(beta u), from one exponential each
  const float* ud = u + 2 * K;
  const int i0 = kTails ? idx - 1 : idx, i1 = kTails ? idx : idx + 1;
  const bool has0 = !kTails || idx > 0, has1 = !kTails || idx < K - 1;
  float u0 = q.tail_const, u1 = q.tail_const;
#pragma unroll
  for (int i = 0; i < P - 2 * K; ++i) {
    u0 = (has0 && i == i0) ? ud[i] : u0;
    u1 = (has1 && i == i1) ? ud[i] : u1;
  }
  auto knot_derivative = [&](float uu, float& dv, float& slope) __attribute__((always_inline)) {
    const float xb = uu * q.beta;
    const float e = exp_lean(-fabsf(fminf(xb, 80.f)));
    const float r = div_lean(1.f, 1.f + e);
    const float sp = fmaxf(xb, 0.f) + log1p_lean_pos(e);
    const float spb = q.beta == 1.f ? sp : div_lean(sp, q.beta);
    dv = q.min_d + (xb > 20.f ? uu : spb);
    slope = xb > 20.f ? 1.f : (xb >= 0.f ? r : e * r);
  };
  float d0v, d1v, s0, s1;
  knot_derivative(u0, d0v, s0);
  knot_derivative(u1, d1v, s1);

  // (y, lad) as functions of (x, x_k, x_k+1, y_k, y_k+1, d_k, d_k+1) and their reverse-mode derivative (see above)
  const float xk = k_lo.x, yk = k_lo.y;
  const float wk = k_hi.x - xk, hk = k_hi.y - yk;
  const float rwk = div_lean(1.f, wk);
  const float delta = hk * rwk;
  const float theta = (x - xk) * rwk;
  const float omt = 1.f - theta, t1 = theta * omt, th2 = theta * theta;
  const float a1 = delta * th2 + d0v * t1;
  const float num = hk * a1;
  const float sd = d0v + d1v - 2.f * delta;
  const float den = delta + sd * t1;
  const float b1 = d1v * th2 + 2.f * delta * t1 + d0v * (omt * omt);
  const float dnum = delta * delta * b1;
  const float rden = div_lean(1.f, den);
  const float g_num = gy * rden;
  const float g_den = -(gy * num * rden + 2.f * gl) * rden;
  const float g_dnum = gl * div_lean(1.f, dnum);
  const float g_b1 = g_dnum * delta * delta;
  const float g_a1 = g_num * hk;
  const float g_sd = g_den * t1;
  const float g_delta = g_dnum * 2.f * delta * b1 + g_b1 * 2.f * t1 + g_den - 2.f * g_sd + g_a1 * th2;
  const float g_d1 = g_b1 * th2 + g_sd;
  const float g_d0 = g_b1 * (omt * omt) + g_sd + g_a1 * t1;
  const float g_th2 = g_b1 * d1v + g_a1 * delta;
  const float g_t1 = g_b1 * 2.f * delta + g_den * sd + g_a1 * d0v;
  const float g_omt = g_b1 * d0v * 2.f * omt + g_t1 * theta;
  const float g_theta = g_th2 * 2.f * theta + g_t1 * omt - g_omt;
  const float g_hk = g_num * a1 + g_delta * rwk;
  const float g_dx = g_theta * rwk;
  const float g_wk = -(g_theta * theta + g_delta * delta) * rwk;
  gx = g_dx;
  // adjoints of the four knots; the interval ends are pinned (no gradient)
  const bool lo_free = idx > 0, hi_free = idx + 1 < K;
  const f2 g_lo = {lo_free ? -g_dx - g_wk : 0.f, lo_free ? gy - g_hk : 0.f};     // (x_k, y_k)
  const f2 g_hi = {hi_free ? g_wk : 0.f, hi_free ? g_hk : 0.f};                  // (x_k+1, y_k+1)
  // knot = lo + span sum_{i<k} (min + c p_i):  d knot_k / d u_m = span c p_m (1[m < k] - P_k) / wh_div
  const f2 cxy = span * c1 * inv_div;
  const f2 both = g_lo + g_hi;
  const f2 corr = g_lo * p_lo + g_hi * p_hi;
#pragma unroll
  for (int m = 0; m < K; ++m) {
    f2 sel;
    sel.x = m < idx ? both.x : (m == idx ? g_hi.x : 0.f);
    sel.y = m < idx ? both.y : (m == idx ? g_hi.y : 0.f);
    const f2 gm = (cxy * p[m]) * (sel - corr);
    gp[m] = gm.x;
    gp[K + m] = gm.y;
  }
  const float gd0 = g_d0 * s0, gd1 = g_d1 * s1;
#pragma unroll
  for (int i = 0; i < P - 2 * K; ++i) gp[2 * K + i] = ((has0 && i == i0) ? gd0 : 0.f) + ((has1 && i == i1) ? gd1 : 0.f);
}

// rq_backward_element_fast as ONE basic block (fc_rq_fused_backward512.h interleaves it with matrix-core instructions of
// another block; a divergent early exit would cut the scheduling region in two).
template <int K, bool kTails>
__device__ __forceinline__ void rq_backward_element_flat(const RQParams& q, float inv_div, const float* __restrict__ u,
                                                         float x_in, float gy_in, float gl_in, float& gx, float* __restrict__ gp) {
  constexpr int P = kTails ? 3 * K - 1 : 3 * K + 1;
  // no early exit: an element outside the interval (identity tails) is evaluated at the clamped input with ZERO upstream
  // gradients -- every parameter gradient comes out as an exact zero -- and takes gx = gy at the end
  const bool inside = (x_in >= q.left) && (x_in <= q.right);
  const float x = fminf(fmaxf(x_in, q.left), q.right);
  const float gy = inside ? gy_in : 0.f, gl = inside ? gl_in : 0.f;
  f2 p[K];
  {
    float mx = -INFINITY, my = -INFINITY;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      p[i] = f2{u[i], u[K + i]} * inv_div;
      mx = fmaxf(mx, p[i].x);
      my = fmaxf(my, p[i].y);
    }
    const f2 m = {mx, my};
    f2 sum = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const f2 d = (p[i] - m) * f2{1.4426950408889634f, 1.4426950408889634f};
      p[i] = f2{__builtin_amdgcn_exp2f(d.x), __builtin_amdgcn_exp2f(d.y)};
      sum += p[i];
    }
    const f2 rs = {div_lean(1.f, sum.x), div_lean(1.f, sum.y)};
#pragma unroll
    for (int i = 0; i < K; ++i) p[i] = p[i] * rs;
  }
  // knots around the bin of x (searched on the width axis) and the prefix sums of the probabilities below them
  const f2 minb = {q.min_w, q.min_h}, c1 = {q.cw, q.ch}, lo = {q.left, q.bottom}, hi = {q.right, q.top};
  const f2 span = hi - lo;
  f2 cum = {0.f, 0.f}, prev = lo, prevp = {0.f, 0.f};
  f2 k_lo = lo, k_hi = lo, p_lo = {0.f, 0.f}, p_hi = {0.f, 0.f};
  int idx = 0;
  {
    f2 psum = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < K; ++i) {
      cum += minb + c1 * p[i];
      psum += p[i];
      const f2 next = (i == K - 1) ? hi : (span * cum + lo);
      const bool take = x >= prev.x;
      idx = take ? i : idx;
      k_lo.x = take ? prev.x : k_lo.x;   k_lo.y = take ? prev.y : k_lo.y;
      k_hi.x = take ? next.x : k_hi.x;   k_hi.y = take ? next.y : k_hi.y;
      p_lo.x = take ? prevp.x : p_lo.x;  p_lo.y = take ? prevp.y : p_lo.y;
      p_hi.x = take ? psum.x : p_hi.x;   p_hi.y = take ? psum.y : p_hi.y;
      prev = next;
      prevp = psum;
    }
  }
  // knot derivatives d = min_d + softplus(u, beta) and their slopes sigmoid(beta u), from one exponential each
  const float* ud = u + 2 * K;
  const int i0 = kTails ? idx - 1 : idx, i1 = kTails ? idx : idx + 1;
  const bool has0 = !kTails || idx > 0, has1 = !kTails || idx < K - 1;
  float u0 = q.tail_const, u1 = q.tail_const;
#pragma unroll
  for (int i = 0; i < P - 2 * K; ++i) {
    u0 = (has0 && i == i0) ? ud[i] : u0;
    u1 = (has1 && i == i1) ? ud[i] : u1;
  }
  auto knot_derivative = [&](float uu, float& dv, float& slope) __attribute__((always_inline)) {
    const float xb = uu * q.beta;
    const float e = exp_lean(-fabsf(fminf(xb, 80.f)));
    const float r = div_lean(1.f, 1.f + e);
    const float sp = fmaxf(xb, 0.f) + log1p_lean_pos_flat(e);
    const float spb = q.beta == 1.f ? sp : div_lean(sp, q.beta);
    dv = q.min_d + (xb > 20.f ? uu : spb);
    slope = xb > 20.f ? 1.f : (xb >= 0.f ? r : e * r);
  };
  float d0v, d1v, s0, s1;
  knot_derivative(u0, d0v, s0);
  knot_derivative(u1, d1v, s1);

  // (y, lad) as functions of (x, x_k, x_k+1, y_k, y_k+1, d_k, d_k+1) and their reverse-mode derivative (see above)
  const float xk = k_lo.x, yk = k_lo.y;
  const float wk = k_hi.x - xk, hk = k_hi.y - yk;
  const float rwk = div_lean(1.f, wk);
  const float delta = hk * rwk;
  const float theta = (x - xk) * rwk;
  const float omt = 1.f - theta, t1 = theta * omt, th2 = theta * theta;
  const float a1 = delta * th2 + d0v * t1;
  const float num = hk * a1;
  const float sd = d0v + d1v - 2.f * delta;
  const float den = delta + sd * t1;
  const float b1 = d1v * th2 + 2.f * delta * t1 + d0v * (omt * omt);
  const float dnum = delta * delta * b1;
  const float rden = div_lean(1.f, den);
  const float g_num = gy * rden;
  const float g_den = -(gy * num * rden + 2.f * gl) * rden;
  const float g_dnum = gl * div_lean(1.f, dnum);
  const float g_b1 = g_dnum * delta * delta;
  const float g_a1 = g_num * hk;
  const float g_sd = g_den * t1;
  const float g_delta = g_dnum * 2.f * delta * b1 + g_b1 * 2.f * t1 + g_den - 2.f * g_sd + g_a1 * th2;
  const float g_d1 = g_b1 * th2 + g_sd;
  const float g_d0 = g_b1 * (omt * omt) + g_sd + g_a1 * t1;
  const float g_th2 = g_b1 * d1v + g_a1 * delta;
  const float g_t1 = g_b1 * 2.f * delta + g_den * sd + g_a1 * d0v;
  const float g_omt = g_b1 * d0v * 2.f * omt + g_t1 * theta;
  const float g_theta = g_th2 * 2.f * theta + g_t1 * omt - g_omt;
  const float g_hk = g_num * a1 + g_delta * rwk;
  const float g_dx = g_theta * rwk;
  const float g_wk = -(g_theta * theta + g_delta * delta) * rwk;
  gx = inside ? g_dx : gy_in;
  // adjoints of the four knots; the interval ends are pinned (no gradient)
  const bool lo_free = idx > 0, hi_free = idx + 1 < K;
  const f2 g_lo = {lo_free ? -g_dx - g_wk : 0.f, lo_free ? gy - g_hk : 0.f};     // (x_k, y_k)
  const f2 g_hi = {hi_free ? g_wk : 0.f, hi_free ? g_hk : 0.f};                  // (x_k+1, y_k+1)
  // knot = lo + span sum_{i<k} (min + c p_i):  d knot_k / d u_m = span c p_m (1[m < k] - P_k) / wh_div
  const f2 cxy = span * c1 * inv_div;
  const f2 both = g_lo + g_hi;
  const f2 corr = g_lo * p_lo + g_hi * p_hi;
#pragma unroll
  for (int m = 0; m < K; ++m) {
    f2 sel;
    sel.x = m < idx ? both.x : (m == idx ? g_hi.x : 0.f);
    sel.y = m < idx ? both.y : (m == idx ? g_hi.y : 0.f);
    const f2 gm = (cxy * p[m]) * (sel - corr);
    gp[m] = gm.x;
    gp[K + m] = gm.y;
  }
  const float gd0 = g_d0 * s0, gd1 = g_d1 * s1;
#pragma unroll
  for (int i = 0; i < P - 2 * K; ++i) gp[2 * K + i] = ((has0 && i == i0) ? gd0 : 0.f) + ((has1 && i == i1) ? gd1 : 0.f);
}


}  // namespace fc
