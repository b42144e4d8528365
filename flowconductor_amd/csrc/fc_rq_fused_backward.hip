// C entry of the fused final-Linear + RQ-spline backward kernel (fc_rq_fused_backward512.h).
#include "fc_rq_fused_backward512.h"

extern "C" int fc_rq_fused_linear_backward(int32_t role, const float* x, const float* h, const float* grad_y,
                                           const float* grad_logabsdet, const void* w_frag, const float* w_unscale,
                                           const float* bias_pad, const void* wt_frag, const int32_t* cols,
                                           float* grad_x, float* grad_h, float* grad_bias_pad, float* grad_w_pad,
                                           int64_t n, int32_t d, int32_t d_t, const fc_rq_config* cfg, void* stream) {
  if (!cfg || n < 0 || d < d_t || d_t < 1 || d_t > 32 || d > 128 || role != 3) return hipErrorInvalidValue;
  if (cfg->inverse) return hipErrorInvalidValue;      // gradients of the forward direction
  if (n % fc::kBwdR != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !h || !grad_y || !w_frag || !w_unscale || !bias_pad || !cols) return hipErrorInvalidValue;
  if (!wt_frag || !grad_x || !grad_h || !grad_bias_pad || !grad_w_pad) return hipErrorInvalidValue;
  if ((((uintptr_t)h | (uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)w_frag | (uintptr_t)wt_frag |
        (uintptr_t)grad_x | (uintptr_t)grad_h) & 15u) != 0)
    return hipErrorInvalidValue;

  fc::RQParams q;
  q.K = cfg->num_bins; q.tails = cfg->tails ? 1 : 0; q.inverse = 0;
  q.left = cfg->left; q.right = cfg->right; q.bottom = cfg->bottom; q.top = cfg->top;
  q.min_w = (float)cfg->min_bin_width; q.min_h = (float)cfg->min_bin_height; q.min_d = (float)cfg->min_derivative;
  q.cw = (float)(1.0 - cfg->min_bin_width * q.K);
  q.ch = (float)(1.0 - cfg->min_bin_height * q.K);
  fc::rq_finish_params(q);
  q.wh_div = cfg->wh_divisor > 0.f ? cfg->wh_divisor : 1.f;
  q.beta = cfg->softplus_beta;
  q.tail_const = cfg->tail_constant;

  fc::BwdArgs a{x, h, grad_y, grad_logabsdet, static_cast<const fc::f16x8*>(w_frag), w_unscale, bias_pad,
                static_cast<const fc::f16x8*>(wt_frag), cols, grad_x, grad_h, grad_bias_pad, grad_w_pad,
                n / fc::kBwdR, d, d_t};
  hipStream_t s = static_cast<hipStream_t>(stream);
  return fc::launch_backward512_any(q.K, q.tails != 0, q, a, s);
}
