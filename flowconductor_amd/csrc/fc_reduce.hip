// Row reductions that follow the bijector stack.
//
// fc_standard_normal_log_prob restates StandardNormal._log_prob
// (flowcon/distributions/normal.py:23-33): -0.5 * sum_j z_j^2 - (float)log_z, and can fold in
// the `+ logabsdet` of Flow._log_prob (flowcon/flows/base.py:48) so the [N] vector is touched
// once.  HBM-bound: reads [N, D] once, writes [N].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_device.h"
#include "../../include/flowcon_hip.h"

namespace fc {

// T lanes (power of two, <= 64) cooperate on one row; a 256-thread block covers 256/T rows per pass and
// strides over the batch.  kVec: rows are read as float4 (d % 4 == 0 and 16-byte aligned rows).
template <int T, bool kVec>
__global__ __launch_bounds__(256) void std_normal_kernel(const float* __restrict__ z,
                                                         const float* __restrict__ add,
                                                         float* __restrict__ out, int64_t n, int d,
                                                         float log_z) {
  const int rows_per_block = 256 / T;
  const int lane = threadIdx.x % T;
  const int64_t stride = (int64_t)gridDim.x * rows_per_block;
  for (int64_t row0 = (int64_t)blockIdx.x * rows_per_block; row0 < n; row0 += stride) {
    const int64_t row = row0 + threadIdx.x / T;
    float acc = 0.f;
    if (row < n) {
      if constexpr (kVec) {
        const float4* r = reinterpret_cast<const float4*>(z + row * d);
        for (int j = lane; j < (d >> 2); j += T) {
          const float4 v = r[j];
          acc += v.x * v.x;
          acc += v.y * v.y;
          acc += v.z * v.z;
          acc += v.w * v.w;
        }
      } else {
        const float* r = z + row * d;
        for (int j = lane; j < d; j += T) {
          const float v = r[j];
          acc += v * v;
        }
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
}

template <bool kVec>
static void launch_std_normal(int t, dim3 g, hipStream_t s, const float* z, const float* add, float* out, int64_t n,
                              int d, float log_z) {
  dim3 b(256);
  switch (t) {
    case 1: hipLaunchKernelGGL((std_normal_kernel<1, kVec>), g, b, 0, s, z, add, out, n, d, log_z); break;
    case 2: hipLaunchKernelGGL((std_normal_kernel<2, kVec>), g, b, 0, s, z, add, out, n, d, log_z); break;
    case 4: hipLaunchKernelGGL((std_normal_kernel<4, kVec>), g, b, 0, s, z, add, out, n, d, log_z); break;
    case 8: hipLaunchKernelGGL((std_normal_kernel<8, kVec>), g, b, 0, s, z, add, out, n, d, log_z); break;
    case 16: hipLaunchKernelGGL((std_normal_kernel<16, kVec>), g, b, 0, s, z, add, out, n, d, log_z); break;
    case 32: hipLaunchKernelGGL((std_normal_kernel<32, kVec>), g, b, 0, s, z, add, out, n, d, log_z); break;
    default: hipLaunchKernelGGL((std_normal_kernel<64, kVec>), g, b, 0, s, z, add, out, n, d, log_z); break;
  }
}

}  // namespace fc

extern "C" int fc_standard_normal_log_prob(const float* z, const float* add, float* out, int64_t n,
                                           int32_t d, float log_z, void* stream) {
  if (n < 0 || d <= 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!z || !out) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (d % 4 == 0) && (((uintptr_t)z & 15u) == 0);
  const int items = vec ? d / 4 : d;   // loads per row
  int t = 1;
  while (t < items && t < 64) t <<= 1;
  const int rows_per_block = 256 / t;
  const int cus = fc::device_cu_count();
  int64_t grid = (n + rows_per_block - 1) / rows_per_block;
  if (grid > (int64_t)cus * 16) grid = (int64_t)cus * 16;   // grid-stride: 16 blocks of 256 threads per CU
  dim3 g((unsigned)grid);
  if (vec) fc::launch_std_normal<true>(t, g, s, z, add, out, n, d, log_z);
  else fc::launch_std_normal<false>(t, g, s, z, add, out, n, d, log_z);
  return hipGetLastError();
}
