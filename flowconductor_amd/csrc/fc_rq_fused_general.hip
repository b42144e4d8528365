// C entry of the general fused final-Linear + RQ-spline kernel (fc_rq_fused_general.h).
#include "fc_rq_fused_general.h"

extern "C" int fc_rq_spline_fused_general(const float* x, float* y, const float* h, const void* w_frag,
                                          const float* w_unscale, const float* bias_pad, const int32_t* cols,
                                          float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d, int32_t d_t,
                                          int32_t hidden, const fc_rq_config* cfg, void* stream) {
  if (!cfg || n < 0 || d < d_t || d_t < 1 || d_t > 32 || d > 128) return hipErrorInvalidValue;
  if (hidden != 64 && hidden != 128 && hidden != 256) return hipErrorInvalidValue;
  if (n % fc::kGenRows != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !h || !w_frag || !w_unscale || !bias_pad || !cols || !logabsdet) return hipErrorInvalidValue;
  if ((((uintptr_t)h | (uintptr_t)x | (uintptr_t)y | (uintptr_t)w_frag) & 15u) != 0) return hipErrorInvalidValue;

  fc::RQParams q;
  q.K = cfg->num_bins; q.tails = cfg->tails ? 1 : 0; q.inverse = cfg->inverse;
  q.left = cfg->left; q.right = cfg->right; q.bottom = cfg->bottom; q.top = cfg->top;
  q.min_w = (float)cfg->min_bin_width; q.min_h = (float)cfg->min_bin_height; q.min_d = (float)cfg->min_derivative;
  q.cw = (float)(1.0 - cfg->min_bin_width * q.K);
  q.ch = (float)(1.0 - cfg->min_bin_height * q.K);
  fc::rq_finish_params(q);
  q.wh_div = cfg->wh_divisor > 0.f ? cfg->wh_divisor : 1.f;
  q.beta = cfg->softplus_beta;
  q.tail_const = cfg->tail_constant;

  fc::GenArgs a{x, y, h, static_cast<const fc::f16x8*>(w_frag), w_unscale, bias_pad, cols, logabsdet, err_flag,
                n / fc::kGenRows, d, hidden, d_t, (cfg->flags & FC_RQ_ACCUMULATE_LOGABSDET) ? 1 : 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  // bin counts with a resident-weight instance (fc_rq_fused4_body.h: hidden 64, linear tails) run there
  if (!(cfg->flags & FC_RQ_STREAMED_WEIGHTS) && fc::fused4_takes(q, a)) return fc::launch_fused4(q, a, s);
  return q.tails ? fc::launch_general_tails(q.K, q, a, s) : fc::launch_general_box(q.K, q, a, s);
}
