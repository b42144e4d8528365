// Scalar f32 helpers that follow the rounding order of the ATen ops the reference calls.
// Built with -ffp-contract=off so `a * b + c` stays two roundings like the reference's
// separate mul / add tensor ops.
#pragma once
#include <hip/hip_runtime.h>

namespace fc {

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
