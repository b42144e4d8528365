// Parameter-free (or scalar-parameter) element-wise bijectors with a per-row log|det J| sum.
//
// Restates (not copies) flowcon/transforms/nonlinearities.py:
//   Exp :18-32, Tanh :35-48, LogTanh :51-112, LeakyReLU :115-136, Sigmoid :139-169,
//   Softplus :172-189, GatedLinearUnit :197-209, CauchyCDF :212-231, ExtendedSoftplus :519-552.
// x, y are viewed as [n, m] (m = elements per batch item); T lanes cooperate on a row and
// butterfly-reduce the row's logabsdet.  Streaming: x read once, y written once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_math.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct EwArgs {
  const float* x;
  float* y;
  float* lad_row;        // [n] or nullptr
  float* lad_elem;       // [n, m] or nullptr (ExtendedSoftplus, GatedLinearUnit)
  const float* aux;      // kind-specific device data (temperature[1], log_slope[1], shift[m], context[n,m])
  uint32_t* err;
  int64_t n;
  int64_t m;
  int kind;
  int inverse;
  float p0, p1, p2, p3;  // kind-specific host scalars
};

__device__ __forceinline__ void ew_eval(const EwArgs& a, int64_t row, int64_t j, float x, float& y,
                                        float& lad, uint32_t& err) {
  switch (a.kind) {
    case FC_EW_EXP:
      if (!a.inverse) { y = expf(x); lad = x; }
      else { if (x <= 0.f) err |= 1u; y = logf(x); lad = -y; }
      break;
    case FC_EW_TANH:
      if (!a.inverse) {
        // nonlinearities.py:40-43 from ONE exponential e = exp(-2|x|) (round 4: libm's tanhf + logf made this a vector-bound
        // kernel at 0.23 of the HBM peak):  tanh = (1 - e) / (1 + e),  log(1 - tanh^2) = log(4 e / (1 + e)^2)
        // = 2 (ln 2 - |x| - log1p(e)) -- no cancellation near saturation, where log(1 - y^2) on a rounded y is off by
        // 1e-4 at |x| = 4 and by 0.3 at |x| = 8 in ANY float32 evaluation, the reference's included (tools/probe/
        // tanh_accuracy.py: reference 3.0e-1, this 2.2e-5 against float64 over 4096 x 64 inputs ~ 2 N(0, 1)).  Where a float32
        // tanh rounds to +-1 the reference's log(0) = -inf is kept.
        const float ax = fabsf(x);
        const float e = exp_lean(-2.f * ax);
        const float t = div_lean(1.f - e, 1.f + e);
        y = x >= 0.f ? t : -t;
        // (9.0109: where 1 - tanh(x) drops below half an ulp of 1, i.e. where a correctly rounded float32 tanh returns 1)
        lad = ax < 9.010913f ? 2.f * ((0.6931471805599453f - ax) - log1p_lean_pos(e)) : -INFINITY;
      } else {
        if (x <= -1.f || x >= 1.f) err |= 1u;
        y = 0.5f * logf((1.f + x) / (1.f - x));
        lad = -logf(1.f - x * x);
      }
      break;
    case FC_EW_LOGTANH: {
      const float cut = a.p0, alpha = a.p1, beta = a.p2, inv_cut = a.p3;
      if (!a.inverse) {
        if (x > cut) { y = alpha * logf(beta * x); lad = logf(alpha / x); }
        else if (x < -cut) { y = alpha * -logf(-beta * x); lad = logf(-alpha / x); }
        else { y = tanhf(x); lad = logf(1.f - y * y); }
      } else {
        const float nlab = -logf(alpha * beta);
        if (x > inv_cut) { y = expf(x / alpha) / beta; lad = nlab + x / alpha; }
        else if (x < -inv_cut) { y = -expf(-x / alpha) / beta; lad = nlab - x / alpha; }
        else { y = 0.5f * logf((1.f + x) / (1.f - x)); lad = -logf(1.f - x * x); }
      }
      break;
    }
    case FC_EW_LEAKY_RELU: {
      const float log_slope = a.aux[0];
      const float slope = a.inverse ? a.p1 : a.p0;  // p0 = slope, p1 = 1/slope
      y = x > 0.f ? x : x * slope;
      lad = x < 0.f ? (a.inverse ? -log_slope : log_slope) : 0.f;
      break;
    }
    case FC_EW_SIGMOID: {
      const float temp = a.aux[0];
      if (!a.inverse) {
        const float t = temp * x;
        y = sigmoidf(t);
        lad = logf(temp) - softplus1(-t) - softplus1(t);
      } else {
        if (x < 0.f || x > 1.f) err |= 1u;
        const float eps = a.p0;
        const float xc = x < eps ? eps : (x > 1.f - eps ? 1.f - eps : x);   // torch.clamp: NaN stays NaN
        y = (1.f / temp) * (logf(xc) - log1pf(-xc));
        lad = -(logf(temp) - softplus1(-temp * y) - softplus1(temp * y));
      }
      break;
    }
    case FC_EW_SOFTPLUS: {
      const float thr = a.p0, eps = a.p1;
      if (!a.inverse) {
        y = (x > thr ? x : log1pf(expf(x))) + eps;
        lad = logsigmoidf(x);
      } else {
        const float xi = x - eps;
        y = xi > thr ? xi : logf(expm1f(xi));
        lad = -logf(-expm1f(-xi));
      }
      break;
    }
    case FC_EW_CAUCHY_CDF: {
      const float inv_pi = 0.3183098861837907f, pi = 3.141592653589793f, nlog_pi = -1.1447298858494002f;
      if (!a.inverse) {
        y = inv_pi * atanf(x) + 0.5f;
        lad = nlog_pi - logf(1.f + x * x);
      } else {
        if (x < 0.f || x > 1.f) err |= 1u;
        y = tanf(pi * (x - 0.5f));
        lad = -(nlog_pi - logf(1.f + y * y));
      }
      break;
    }
    case FC_EW_EXTENDED_SOFTPLUS: {
      const float sh = softplus1(a.aux[j]) + 1e-1f;
      y = softplus1(x - sh) + -softplus1(-(x + sh));
      const float lj_pos = -logaddexpf(sh, x) + x;
      const float lj_neg = -softplus1(sh + x);
      lad = logaddexpf(lj_pos, lj_neg);
      break;
    }
    default: {  // FC_EW_GLU: gate = sigmoid(context)
      const float gate = sigmoidf(a.aux[row * a.m + j]);
      if (!a.inverse) { y = x * gate; lad = logf(gate); }
      else { y = x / gate; lad = -logf(gate); }
      break;
    }
  }
}

template <int T>
__global__ __launch_bounds__(256) void elementwise_kernel(EwArgs a) {
  const int rows_per_block = 256 / T;
  const int lane = threadIdx.x % T;
  const int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / T;
  float acc = 0.f;
  uint32_t err = 0;
  if (row < a.n) {
    const float* xr = a.x + row * a.m;
    float* yr = a.y + row * a.m;
    for (int64_t j = lane; j < a.m; j += T) {
      float y, lad;
      ew_eval(a, row, j, xr[j], y, lad, err);
      yr[j] = y;
      if (a.lad_elem) a.lad_elem[row * a.m + j] = lad;
      acc += lad;
    }
  }
#pragma unroll
  for (int o = T >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, T);
  if (row < a.n && lane == 0 && a.lad_row) a.lad_row[row] = acc;
  if (err && a.err) atomicOr(a.err, err);
}

// Rows whose length is a multiple of 4 (16-byte aligned buffers): a lane moves float4 pieces -- four times the bytes in
// flight per lane and a quarter of the butterfly steps of the one-element-per-lane form above.
template <int T>
__global__ __launch_bounds__(256) void elementwise_kernel_v4(EwArgs a) {
  const int rows_per_block = 256 / T;
  const int lane = threadIdx.x % T;
  const int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / T;
  float acc = 0.f;
  uint32_t err = 0;
  if (row < a.n) {
    const float4* xr = reinterpret_cast<const float4*>(a.x + row * a.m);
    float4* yr = reinterpret_cast<float4*>(a.y + row * a.m);
    float4* lr = a.lad_elem ? reinterpret_cast<float4*>(a.lad_elem + row * a.m) : nullptr;
    const int64_t m4 = a.m >> 2;
    for (int64_t j4 = lane; j4 < m4; j4 += T) {
      const float4 xv = xr[j4];
      float4 yv, lv;
      ew_eval(a, row, 4 * j4 + 0, xv.x, yv.x, lv.x, err);
      ew_eval(a, row, 4 * j4 + 1, xv.y, yv.y, lv.y, err);
      ew_eval(a, row, 4 * j4 + 2, xv.z, yv.z, lv.z, err);
      ew_eval(a, row, 4 * j4 + 3, xv.w, yv.w, lv.w, err);
      yr[j4] = yv;
      if (lr) lr[j4] = lv;
      acc += (lv.x + lv.y) + (lv.z + lv.w);
    }
  }
#pragma unroll
  for (int o = T >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, T);
  if (row < a.n && lane == 0 && a.lad_row) a.lad_row[row] = acc;
  if (err && a.err) atomicOr(a.err, err);
}

}  // namespace fc

extern "C" int fc_elementwise(const float* x, float* y, float* logabsdet_row, float* logabsdet_elem,
                              const float* aux, uint32_t* err_flag, int64_t n, int64_t m, int32_t kind,
                              int32_t inverse, float p0, float p1, float p2, float p3, void* stream) {
  if (n < 0 || m <= 0 || kind < 0 || kind > FC_EW_GLU) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y) return hipErrorInvalidValue;
  const bool needs_aux = kind == FC_EW_LEAKY_RELU || kind == FC_EW_SIGMOID ||
                         kind == FC_EW_EXTENDED_SOFTPLUS || kind == FC_EW_GLU;
  if (needs_aux && !aux) return hipErrorInvalidValue;
  fc::EwArgs a{x, y, logabsdet_row, logabsdet_elem, aux, err_flag, n, m, kind, inverse, p0, p1, p2, p3};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (m % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)logabsdet_elem) & 15u) == 0) {
    int t4 = 1;
    while (t4 < m / 4 && t4 < 64) t4 <<= 1;
    const int rows_per_block4 = 256 / t4;
    const int64_t grid4 = (n + rows_per_block4 - 1) / rows_per_block4;
    if (grid4 > 0x7fffffffLL) return hipErrorInvalidConfiguration;
    dim3 g4((unsigned)grid4), b4(256);
    switch (t4) {
      case 1: hipLaunchKernelGGL(fc::elementwise_kernel_v4<1>, g4, b4, 0, s, a); break;
      case 2: hipLaunchKernelGGL(fc::elementwise_kernel_v4<2>, g4, b4, 0, s, a); break;
      case 4: hipLaunchKernelGGL(fc::elementwise_kernel_v4<4>, g4, b4, 0, s, a); break;
      case 8: hipLaunchKernelGGL(fc::elementwise_kernel_v4<8>, g4, b4, 0, s, a); break;
      case 16: hipLaunchKernelGGL(fc::elementwise_kernel_v4<16>, g4, b4, 0, s, a); break;
      case 32: hipLaunchKernelGGL(fc::elementwise_kernel_v4<32>, g4, b4, 0, s, a); break;
      default: hipLaunchKernelGGL(fc::elementwise_kernel_v4<64>, g4, b4, 0, s, a); break;
    }
    return hipGetLastError();
  }
  int t = 1;
  while (t < m && t < 64) t <<= 1;
  const int rows_per_block = 256 / t;
  const int64_t grid = (n + rows_per_block - 1) / rows_per_block;
  if (grid > 0x7fffffffLL) return hipErrorInvalidConfiguration;
  dim3 g((unsigned)grid), b(256);
  switch (t) {
    case 1: hipLaunchKernelGGL(fc::elementwise_kernel<1>, g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL(fc::elementwise_kernel<2>, g, b, 0, s, a); break;
    case 4: hipLaunchKernelGGL(fc::elementwise_kernel<4>, g, b, 0, s, a); break;
    case 8: hipLaunchKernelGGL(fc::elementwise_kernel<8>, g, b, 0, s, a); break;
    case 16: hipLaunchKernelGGL(fc::elementwise_kernel<16>, g, b, 0, s, a); break;
    case 32: hipLaunchKernelGGL(fc::elementwise_kernel<32>, g, b, 0, s, a); break;
    default: hipLaunchKernelGGL(fc::elementwise_kernel<64>, g, b, 0, s, a); break;
  }
  return hipGetLastError();
}
