// Feature permutation: bit-exact gather along one dimension.
//
// Restates Permutation._permute (flowcon/transforms/permutations.py:27-46):
// outputs = index_select(inputs, dim, perm), logabsdet = 0.  The tensor is viewed as
// [outer, d, inner]; inner == 1 is the usual [N, D] case.  Pure data movement: reads every
// input element once, writes every output element once (HBM-bound, 8 B per element).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flowcon_hip.h"

namespace fc {

// [N, D] rows: a block owns R rows; the rows are staged in LDS with 16-byte loads and read
// back through the permutation, so both HBM streams are fully coalesced.
__global__ __launch_bounds__(256) void permute_rows_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y,
                                                           const int32_t* __restrict__ perm,
                                                           int64_t n, int d, int rows) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int64_t r0 = (int64_t)blockIdx.x * rows;
  const int r_eff = (int)((n - r0) < (int64_t)rows ? (n - r0) : (int64_t)rows);
  const int count = r_eff * d;
  const float* src = x + r0 * d;
  float* dst = y + r0 * d;
  const bool vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0;
  if (vec) {
    const int nvec = count >> 2;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x)
      reinterpret_cast<float4*>(tile)[i] = reinterpret_cast<const float4*>(src)[i];
    for (int i = (nvec << 2) + threadIdx.x; i < count; i += blockDim.x) tile[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < count; i += blockDim.x) tile[i] = src[i];
  }
  __syncthreads();
  if (vec) {
    const int nvec = count >> 2;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
      float4 v;
      const int e = i << 2;
      int r = e / d, j = e - r * d;
      float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        vp[k] = tile[r * d + perm[j]];
        if (++j == d) { j = 0; ++r; }
      }
      reinterpret_cast<float4*>(dst)[i] = v;
    }
    for (int e = (nvec << 2) + threadIdx.x; e < count; e += blockDim.x) {
      const int r = e / d, j = e - r * d;
      dst[e] = tile[r * d + perm[j]];
    }
  } else {
    for (int e = threadIdx.x; e < count; e += blockDim.x) {
      const int r = e / d, j = e - r * d;
      dst[e] = tile[r * d + perm[j]];
    }
  }
}

// general [outer, d, inner] gather (inner > 1: e.g. NCHW channel permutation)
__global__ __launch_bounds__(256) void permute_strided_kernel(const float* __restrict__ x,
                                                              float* __restrict__ y,
                                                              const int32_t* __restrict__ perm,
                                                              int64_t total, int d, int64_t inner) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t i = e % inner;
    const int64_t oj = e / inner;
    const int64_t o = oj / d;
    const int j = (int)(oj - o * d);
    y[e] = x[(o * d + perm[j]) * inner + i];
  }
}

}  // namespace fc

extern "C" int fc_permute(const float* x, float* y, const int32_t* perm, int64_t outer, int32_t d,
                          int64_t inner, void* stream) {
  if (outer < 0 || d <= 0 || inner <= 0) return hipErrorInvalidValue;
  if (outer == 0) return hipSuccess;
  if (!x || !y || !perm || x == y) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (inner == 1 && d <= 8192) {
    int rows = 2048 / d;  // ~8 KiB of LDS per block
    if (rows < 1) rows = 1;
    if (rows >= 4) rows &= ~3;
    const int64_t grid = (outer + rows - 1) / rows;
    if (grid > 0x7fffffffLL) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(fc::permute_rows_kernel, dim3((unsigned)grid), dim3(256),
                       sizeof(float) * (size_t)rows * d, s, x, y, perm, outer, d, rows);
  } else {
    const int64_t total = outer * d * inner;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(fc::permute_strided_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, y,
                       perm, total, d, inner);
  }
  return hipGetLastError();
}
