// Affine / additive bijectors with per-sample parameters (coupling and masked-autoregressive).
//
// Restates (not copies):
//   flowcon/transforms/coupling.py:234-252   shift = p[:, :d_t], u = p[:, d_t:],
//                                            s = sigmoid(u + 2) + 1e-3  (default) or
//                                            clamp(softplus(u) + 1e-3, 0, 3); y = x*s + shift
//   flowcon/transforms/coupling.py:255-269   additive: s == 1, logabsdet == 0
//   flowcon/transforms/autoregressive/autoregressive.py:97-129
//                                            p.view(N, D, 2): u = p[..., 0], shift = p[..., 1],
//                                            s = softplus(u) + 1e-3
//   flowcon/transforms/autoregressive/autoregressive.py:164-196
//                                            MaskedShift: forward x + 2*tanh(p), inverse x - p
//   flowcon/transforms/conditional.py:155-272      per-sample shift (additive) / scale = softplus(p) + 1e-5
#include "fc_tile.h"
#include "fc_math.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct AffineOp {
  static constexpr bool kHasPrepare = false;
  __device__ void prepare(float*, int, int) const {}
  int act;      // FC_AFFINE_*
  int inverse;

  __device__ __forceinline__ void eval(const float* __restrict__ prow, int j, int d_t, float x,
                                       float& y, float& lad, uint32_t& err) const {
    float shift, s;
    bool unit = false;
    switch (act) {
      case FC_AFFINE_SIGMOID_PLUS2:
        shift = prow[j];
        s = sigmoidf(prow[d_t + j] + 2.f) + 1e-3f;
        break;
      case FC_AFFINE_SOFTPLUS_CLAMP3:
        shift = prow[j];
        s = fminf(fmaxf(softplus1(prow[d_t + j]) + 1e-3f, 0.f), 3.f);
        break;
      case FC_AFFINE_SCALE_GIVEN:
        shift = prow[j];
        s = prow[d_t + j];
        break;
      case FC_AFFINE_MAF_SOFTPLUS:
        s = softplus1(prow[2 * j]) + 1e-3f;
        shift = prow[2 * j + 1];
        break;
      case FC_AFFINE_SCALE_SOFTPLUS:
        shift = 0.f;
        s = softplus1(prow[j]) + 1e-5f;
        break;
      case FC_AFFINE_SHIFT_TANH2:
        shift = tanhf(prow[j]) * 2.f;
        s = 1.f;
        unit = true;
        break;
      default:  // FC_AFFINE_ADDITIVE
        shift = prow[j];
        s = 1.f;
        unit = true;
        break;
    }
    const float ls = unit ? 0.f : logf(s);
    if (!inverse) {
      y = x * s + shift;
      lad = ls;
    } else {
      y = (x - shift) / s;
      lad = -ls;
    }
  }
};

}  // namespace fc

extern "C" int fc_affine(const float* x, float* y, const float* params, const int32_t* cols,
                         float* logabsdet, int64_t n, int32_t d, int32_t d_t, int32_t activation,
                         int32_t inverse, int32_t shared_params, int32_t lad_mode, void* stream) {
  if (n < 0 || d <= 0 || d_t <= 0 || d_t > d) return hipErrorInvalidValue;
  if (activation < 0 || activation > FC_AFFINE_SCALE_SOFTPLUS) return hipErrorInvalidValue;
  if (n > 0 && (!x || !y || !params)) return hipErrorInvalidValue;
  fc::AffineOp op{activation, inverse};
  fc::TileArgs a{};
  a.x = x; a.y = y; a.params = params; a.cols = cols; a.logabsdet = logabsdet; a.err = nullptr;
  a.N = n; a.D = d; a.d_t = d_t;
  a.rowlen = (activation == FC_AFFINE_ADDITIVE || activation == FC_AFFINE_SHIFT_TANH2 ||
              activation == FC_AFFINE_SCALE_SOFTPLUS) ? d_t : 2 * d_t;
  a.shared_params = shared_params;
  a.lad_mode = lad_mode;
  return fc::launch_tile(op, a, static_cast<hipStream_t>(stream));
}
