// Row-per-wavefront helpers shared by fc_rowwave.hip (forward / inverse kernels) and fc_rowwave_backward.hip: one
// 64-lane wave owns one sample row, lane l holds elements l, l + 64, ... in registers (E <= 8, D <= 512).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_lane.h"

namespace fc {

constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ float wave_sum(float v) { return wave64_allsum(v, threadIdx.x & 63); }
// the ds_bpermute butterfly: slower per reduction, but the shared-parameter Sylvester kernel runs faster with it
// (3.4 vs 5.4 ms at D = 128, M = 32, N = 2^18 -- its reductions overlap the column reads of the mat-vecs)
__device__ __forceinline__ float wave_sum_lds(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int E>
struct Row {
  float v[E];
};

template <int E>
__device__ __forceinline__ void load_row(Row<E>& r, const float* __restrict__ p, int d, int lane) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = lane + 64 * e;
    r.v[e] = i < d ? p[i] : 0.f;
  }
}

template <int E>
__device__ __forceinline__ void store_row(const Row<E>& r, float* __restrict__ p, int d, int lane) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = lane + 64 * e;
    if (i < d) p[i] = r.v[e];
  }
}

template <int E, bool kLds = false>
__device__ __forceinline__ float dot_rows(const Row<E>& a, const Row<E>& b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) s += a.v[e] * b.v[e];
  return kLds ? wave_sum_lds(s) : wave_sum(s);
}

// x_j for a wave-uniform j: element j lives in register j/64 of lane j%64
template <int E>
__device__ __forceinline__ float bcast(const Row<E>& r, int j) {
  float out = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    // j is wave-uniform: v_readlane, no LDS traffic
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.v[e]), j & 63));
    if ((j >> 6) == e) out = c;
  }
  return out;
}

}  // namespace fc
