// C entry of the fused final-Linear + RQ-spline kernel.  The kernel is in fc_rq_fused3.hip.
//
//   params[n, :] = h[n, :] @ W^T + b          W: [d_t*(3K-1), H]    (nn.Linear of ResidualNet,
//                                                                     flowcon/nn/nets/resnet.py:91,99)
//   y, logabsdet = rq_spline_coupling(x, params)                     (coupling.py:279-293, 549-582)
//
// Unfused, the [N, 736] f32 parameter tensor is written by the GEMM and read back by the spline
// kernel: 5.9 KB of HBM traffic per sample and layer for cfg 3, 85 % of everything the layer moves.
// Fused, it only ever exists in registers.  Specialised for the north-star layer shape: H = 64, d_t = 32,
// K = 8, linear tails (P = 23); up to 32 transformed dims (wave w owns dims 4w..4w+3; the weight image is padded
// to a multiple of 4 dims, padding lanes evaluate but do not write; with fewer than 29 dims the remaining waves
// only move tiles).
//
// History of the structure (measurements per 2^20-row launch):
//   1. f32-input MFMA, 4 producer + 4 consumer waves, parameters through LDS        1.43 ms
//   2. f32-input MFMA, 8 symmetric waves, MFMA and spline in one stream             1.27 ms
//      (v_mfma_f32_*_f32 does not overlap with VALU work: MFMA-only 0.85 + spline-only 0.72 = both 1.29)
//   3. three-piece bf16 split on the bf16 matrix cores, parameters born in registers 0.77 ms
//   4. scaled two-piece f16 split (this code)                                       0.55 ms
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_tile.h"
#include "fc_rq_op.h"
#include "fc_rq_fused.h"
#include "../../include/flowcon_hip.h"

extern "C" int fc_rq_spline_fused_linear(const float* x, float* y, const float* h, const float* w_pad,
                                         const float* bias_pad, const int32_t* cols, float* logabsdet,
                                         uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t, int32_t hidden,
                                         const fc_rq_config* cfg, void* stream) {
  if (!cfg || n < 0 || d < d_t) return hipErrorInvalidValue;
  if (hidden != fc::kH || d_t < 1 || d_t > fc::kDt || cfg->num_bins != fc::kK || cfg->tails != 1 ||
      d > 128)
    return hipErrorInvalidValue;  // only the north-star layer shape is fused; callers fall back otherwise
  if (n % fc::kRowsMin != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !h || !w_pad || !bias_pad || !cols || !logabsdet) return hipErrorInvalidValue;
  if ((((uintptr_t)h | (uintptr_t)x | (uintptr_t)y) & 15u) != 0) return hipErrorInvalidValue;

  fc::RQOp<fc::kK> op;
  fc::RQParams& q = op.q;
  q.K = cfg->num_bins; q.tails = 1; q.inverse = cfg->inverse;
  q.left = cfg->left; q.right = cfg->right; q.bottom = cfg->bottom; q.top = cfg->top;
  q.min_w = (float)cfg->min_bin_width; q.min_h = (float)cfg->min_bin_height; q.min_d = (float)cfg->min_derivative;
  q.cw = (float)(1.0 - cfg->min_bin_width * q.K);
  q.ch = (float)(1.0 - cfg->min_bin_height * q.K);
  fc::rq_finish_params(q);
  q.wh_div = cfg->wh_divisor > 0.f ? cfg->wh_divisor : 1.f;
  q.beta = cfg->softplus_beta;
  q.tail_const = cfg->tail_constant;
  op.inv_div = 1.f / q.wh_div;
  op.inv_beta = 1.f / q.beta;

  const int acc = (cfg->flags & FC_RQ_ACCUMULATE_LOGABSDET) ? 1 : 0;
  const int wrows = (cfg->flags & FC_RQ_RAW_WEIGHTS) ? fc::kPP - 1 : fc::kPP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t cus = fc::device_cu_count();   // one persistent 512-thread workgroup per CU
  // 64-row tiles (fewer barriers and tile hand-overs per row) when their LDS image fits, 32-row tiles for very
  // wide inputs and for the last 32 rows of an odd multiple of 32
  const bool wide_ok = fc::fused3_lds_bytes(d, 64) <= 160 * 1024;
  const int64_t n64 = wide_ok ? n - n % 64 : 0;
  if (n64 > 0) {
    fc::FusedArgs a{x, y, h, w_pad, bias_pad, cols, logabsdet, err_flag, n64 / 64, d, acc, d_t, wrows};
    const hipError_t e = fc::launch_fused3(op, a, 64, (unsigned)(cus < a.tiles ? cus : a.tiles), s);
    if (e != hipSuccess) return e;
  }
  if (n64 < n) {
    fc::FusedArgs a{x + n64 * d, y + n64 * d, h + n64 * fc::kH, w_pad, bias_pad, cols, logabsdet + n64, err_flag,
                    (n - n64) / 32, d, acc, d_t, wrows};
    return fc::launch_fused3(op, a, 32, (unsigned)(cus < a.tiles ? cus : a.tiles), s);
  }
  return hipSuccess;
}
