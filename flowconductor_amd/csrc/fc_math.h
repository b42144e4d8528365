// Scalar f32 helpers that follow the rounding order of the ATen ops the reference calls.
// Built with -ffp-contract=off so `a * b + c` stays two roundings like the reference's
// separate mul / add tensor ops.
#pragma once
#include <hip/hip_runtime.h>

namespace fc {

// ---- lean f32 primitives ----------------------------------------------------------------------------
// ocml's expf / logf / log1pf / IEEE division spend most of their instructions on denormal, overflow
// and special-value handling (15 / 14 / 121 / 12 VALU ops on gfx950).  The bijector hot loop only
// meets finite, normal-range arguments, so these versions keep the same hardware transcendental plus
// the error-compensation step and drop the rest; each stays within ~1 ulp (tools/noise_floor.py
// checks the end-to-end error distribution against float64).

// exp(x) = 2^(x*log2e): v_exp_f32 on the rounded product, then a first-order correction for the
// product's rounding error (hi/lo split of log2e).  6 VALU ops.  Valid for x <= ~88.
__device__ __forceinline__ float exp_lean(float x) {
  const float kLog2eHi = 1.4426950216293335f, kLog2eLo = 1.925963033500011e-8f, kLn2 = 0.6931471805599453f;
  const float hi = x * kLog2eHi;
  float lo = __builtin_fmaf(x, kLog2eHi, -hi);
  lo = __builtin_fmaf(x, kLog2eLo, lo);
  const float e = __builtin_amdgcn_exp2f(hi);
  return __builtin_fmaf(e, lo * kLn2, e);
}

// log(x) = log2(x) * ln2 with a hi/lo split of ln2.  4 VALU ops.  x normal and positive.
__device__ __forceinline__ float log_lean(float x) {
  const float kLn2Hi = 0.6931471824645996f, kLn2Lo = -1.904654323148236e-9f;
  const float l2 = __builtin_amdgcn_logf(x);
  const float hi = l2 * kLn2Hi;
  const float err = __builtin_fmaf(l2, kLn2Hi, -hi);
  return hi + __builtin_fmaf(l2, kLn2Lo, err);
}

// a / b by v_rcp_f32 plus one residual correction: 4 VALU ops, <= 1 ulp for normal-range operands.
__device__ __forceinline__ float div_lean(float a, float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float q = a * r;
  const float e = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(e, r, q);
}

// log1p(t) for t >= 0 (t = exp(.) in softplus): log(u) * t / (u - 1) with u = 1 + t undoes the
// rounding of 1 + t (Kahan).  ~12 VALU ops instead of 121.
__device__ __forceinline__ float log1p_lean_pos(float t) {
  const float u = 1.f + t;
  const float d = u - 1.f;
  return d == 0.f ? t : log_lean(u) * div_lean(t, d);
}

// log1p_lean_pos without control flow: the quotient form is evaluated for every lane (behind an opaque asm, so that the
// compiler cannot turn the select back into a branch around it) -- for code that must stay one basic block
__device__ __forceinline__ float log1p_lean_pos_flat(float t) {
  const float u = 1.f + t;
  const float d = u - 1.f;
  float r = log_lean(u) * div_lean(t, d == 0.f ? 1.f : d);
  asm volatile("" : "+v"(r));
  return d == 0.f ? t : r;
}

// sqrt(x) by v_sqrt_f32 plus one Newton correction of the residual; x normal and non-negative.
__device__ __forceinline__ float sqrt_lean(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  if (!(s > 0.f)) return s;
  const float e = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(e, 0.5f * __builtin_amdgcn_rcpf(s), s);
}

// F.softplus(x, beta, threshold=20) on the lean primitives
__device__ __forceinline__ float softplus_lean(float x, float beta) {
  const float xb = x * beta;
  if (xb > 20.f) return x;
  const float v = log1p_lean_pos(exp_lean(xb));
  return beta == 1.f ? v : div_lean(v, beta);
}

// the same without control flow (both sides evaluated, select at the end): lets the scheduler interleave
// several independent element evaluations in one basic block
__device__ __forceinline__ float softplus_lean_sel(float x, float beta) {
  const float xb = x * beta;
  const float t = exp_lean(fminf(xb, 20.f));
  const float u = 1.f + t;
  const float d = u - 1.f;
  const float l1p = d == 0.f ? t : log_lean(u) * div_lean(t, d == 0.f ? 1.f : d);
  const float v = beta == 1.f ? l1p : div_lean(l1p, beta);
  return xb > 20.f ? x : v;
}

// The same on the bare hardware exp2 / log2 (no hi / lo compensation of the log2(e) / ln(2) products): the relative error
// of the result grows by <= 4e-8 |x beta| (<= 2e-7 where the softplus is not yet linear), which measured as no change of
// the RQ kernels' error against float64 (tools/probe/fused_accuracy.py); log1p(e) = log(u) + (e - (u - 1)) / u with
// u = fl(1 + e) keeps tiny e exact.  11 VALU + 3 transcendental instructions instead of 20 + 3.
__device__ __forceinline__ float softplus_plain(float x, float beta, float inv_beta) {
  const float xb = x * beta;
  const float ex = __builtin_amdgcn_exp2f(fminf(xb, 20.f) * 1.4426950408889634f);
  const float up = 1.f + ex;
  const float rr = ex - (up - 1.f);
  const float l1p = __builtin_fmaf(__builtin_amdgcn_logf(up), 0.6931471805599453f, rr * __builtin_amdgcn_rcpf(up));
  return xb > 20.f ? x : l1p * inv_beta;
}

// sigmoid and tanh on the lean primitives (one exponential each, ~1-2 ulp; no overflow for any finite x)
__device__ __forceinline__ float sigmoid_lean(float v) {
  const float e = exp_lean(-fabsf(v));
  const float r = div_lean(1.f, 1.f + e);
  return v >= 0.f ? r : e * r;
}
__device__ __forceinline__ float tanh_lean(float v) {
  const float e = exp_lean(-2.f * fabsf(v));
  const float t = div_lean(1.f - e, 1.f + e);
  return v >= 0.f ? t : -t;
}

// F.softplus(x, beta, threshold=20): x*beta > 20 ? x : log1p(exp(x*beta)) / beta
__device__ __forceinline__ float softplus_b(float x, float beta) {
  const float xb = x * beta;
  return xb > 20.f ? x : log1pf(expf(xb)) / beta;
}

__device__ __forceinline__ float softplus1(float x) {
  return x > 20.f ? x : log1pf(expf(x));
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// F.logsigmoid(x) = min(x, 0) - log1p(exp(-|x|))
__device__ __forceinline__ float logsigmoidf(float x) {
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// torch.logaddexp(a, b) for finite inputs
__device__ __forceinline__ float logaddexpf(float a, float b) {
  const float m = fmaxf(a, b);
  return m + log1pf(expf(-fabsf(a - b)));
}

}  // namespace fc
