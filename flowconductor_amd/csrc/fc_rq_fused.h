// Shared definitions of the fused final-Linear + RQ-spline kernel (fc_rq_fused.hip: C entry,
// fc_rq_fused3.hip: kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_rq_op.h"

namespace fc {

constexpr int kRowsMin = 32;      // the entry takes multiples of 32 rows (tiles of 64 rows + one of 32)
constexpr int kH = 64;            // hidden width (GEMM K)
constexpr int kDt = 32;           // transformed dims, at most (4 per wave; fewer dims leave waves without spline work)
constexpr int kK = 8;             // spline bins
constexpr int kPP = 24;           // parameters per dim, padded from 3K - 1 = 23

struct FusedArgs {
  const float* x;        // [N, D]
  float* y;              // [N, D]
  const float* h;        // [N, 64] last hidden activation of the conditioner
  const float* wpad;     // [ceil(dt/4)*4*24, 64] final-layer weight, zero-padded from 23 to 24 rows per dim
  const float* bias;     // padded bias, same rows
  const int32_t* cols;   // [dt]
  float* logabsdet;      // [N]
  uint32_t* err;
  int64_t tiles;         // full tiles (64 or 32 rows, see launch_fused3)
  int D;
  int accumulate;        // logabsdet[n] += instead of = (FC_RQ_ACCUMULATE_LOGABSDET)
  int dt;                // transformed dims, <= 32 (wpad / bias hold ceil(dt / 4) * 4 dims)
  int wrows;             // rows per dim in wpad / bias: 24 (zero-padded) or 23 (the nn.Linear tensors as they are, dt dims)
};

size_t fused3_lds_bytes(int d, int rows);
hipError_t launch_fused3(const RQOp<kK>& op, const FusedArgs& a, int rows, unsigned grid, hipStream_t stream);

}  // namespace fc
