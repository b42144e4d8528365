// Shared definitions of the two fused final-Linear + RQ-spline kernels (fc_rq_fused.hip, fc_rq_fused2.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_rq_op.h"

namespace fc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kR = 32;            // rows per tile
constexpr int kH = 64;            // hidden width (GEMM K)
constexpr int kDt = 32;           // transformed dims
constexpr int kK = 8;             // spline bins
constexpr int kPP = 24;           // padded
constexpr int kHalfDims = 16;
constexpr int kHalfCols = kHalfDims * kPP;        // 384 = 12 tiles of 32
constexpr int kPRow = kHalfCols + 2 * 4 + 1;      // + skew (2 floats per group of 4 dims) + odd pad = 393
constexpr int kPBuf = (2 * kR * kPRow + 3) & ~3;   // floats, rounded so the next region is 16-byte aligned
constexpr int kHRow = kH + 1;                     // 65: conflict-free A-fragment reads
constexpr int kHBuf = (2 * kR * kHRow + 3) & ~3;
constexpr int kTilesPerWave = 3;                  // 12 tiles per half / 4 producer waves
constexpr int kSteps = kH / 2;                    // 32 MFMA k-steps

__device__ __forceinline__ int skewed(int c) { return c + 2 * (c / (4 * kPP)); }

struct FusedArgs {
  const float* x;        // [N, D]
  float* y;              // [N, D]
  const float* h;        // [N, 64] last hidden activation of the conditioner
  const float* wpad;     // [768, 64] final-layer weight, zero-padded from 23 to 24 rows per dim
  const float* bias;     // [768] padded bias
  const int32_t* cols;   // [32]
  float* logabsdet;      // [N]
  uint32_t* err;
  int64_t tiles;         // full 32-row tiles
  int D;
  int debug;             // profiling ablations (tools/ only): 1 skip MFMA, 2 skip spline arithmetic
};


hipError_t launch_fused2(const RQOp<kK>& op, const FusedArgs& a, size_t lds, unsigned grid, hipStream_t stream);
size_t fused3_lds_bytes(int d);
hipError_t launch_fused3(const RQOp<kK>& op, const FusedArgs& a, unsigned grid, hipStream_t stream);

}  // namespace fc
