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

}  // namespace fc
