// Row reductions that follow the bijector stack.
//
// fc_standard_normal_log_prob restates StandardNormal._log_prob
// (flowcon/distributions/normal.py:23-33): -0.5 * sum_j z_j^2 - (float)log_z, and can fold in
// the `+ logabsdet` of Flow._log_prob (flowcon/flows/base.py:48) so the [N] vector is touched
// once.  HBM-bound: reads [N, D] once, writes [N].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flowcon_hip.h"

namespace fc {

// T lanes (power of two, <= 64) cooperate on one row; a 256-thread block covers 256/T rows.
template <int T>
__global__ __launch_bounds__(256) void std_normal_kernel(const float* __restrict__ z,
                                                         const float* __restrict__ add,
                                                         float* __restrict__ out, int64_t n, int d,
                                                         float log_z) {
  const int rows_per_block = 256 / T;
  const int lane = threadIdx.x % T;
  const int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / T;
  float acc = 0.f;
  if (row < n) {
    const float* r = z + row * d;
    for (int j = lane; j < d; j += T) {
      const float v = r[j];
      acc += v * v;
    }
  }
#pragma unroll
  for (int o = T >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, T);
  if (row < n && lane == 0) {
    float v = -0.5f * acc - log_z;
    if (add) v += add[row];
    out[row] = v;
  }
}

}  // namespace fc

extern "C" int fc_standard_normal_log_prob(const float* z, const float* add, float* out, int64_t n,
                                           int32_t d, float log_z, void* stream) {
  if (n < 0 || d <= 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!z || !out) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int t = 1;
  while (t < d && t < 64) t <<= 1;
  const int rows_per_block = 256 / t;
  const int64_t grid = (n + rows_per_block - 1) / rows_per_block;
  if (grid > 0x7fffffffLL) return hipErrorInvalidConfiguration;
  dim3 g((unsigned)grid), b(256);
  switch (t) {
    case 1: hipLaunchKernelGGL(fc::std_normal_kernel<1>, g, b, 0, s, z, add, out, n, d, log_z); break;
    case 2: hipLaunchKernelGGL(fc::std_normal_kernel<2>, g, b, 0, s, z, add, out, n, d, log_z); break;
    case 4: hipLaunchKernelGGL(fc::std_normal_kernel<4>, g, b, 0, s, z, add, out, n, d, log_z); break;
    case 8: hipLaunchKernelGGL(fc::std_normal_kernel<8>, g, b, 0, s, z, add, out, n, d, log_z); break;
    case 16: hipLaunchKernelGGL(fc::std_normal_kernel<16>, g, b, 0, s, z, add, out, n, d, log_z); break;
    case 32: hipLaunchKernelGGL(fc::std_normal_kernel<32>, g, b, 0, s, z, add, out, n, d, log_z); break;
    default: hipLaunchKernelGGL(fc::std_normal_kernel<64>, g, b, 0, s, z, add, out, n, d, log_z); break;
  }
  return hipGetLastError();
}
