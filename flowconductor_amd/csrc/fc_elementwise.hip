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
      if (!a.inverse) { y = tanhf(x); lad = logf(1.f - y * y); }
      else {
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
  int t = 1;
  while (t < m && t < 64) t <<= 1;
  const int rows_per_block = 256 / t;
  const int64_t grid = (n + rows_per_block - 1) / rows_per_block;
  if (grid > 0x7fffffffLL) return hipErrorInvalidConfiguration;
  hipStream_t s = static_cast<hipStream_t>(stream);
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
