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

// torch.clamp: a NaN stays a NaN (fminf / fmaxf would return the bound)
__device__ __forceinline__ float clamp_like_torch(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }


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
        s = clamp_like_torch(softplus1(prow[d_t + j]) + 1e-3f, 0.f, 3.f);
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

// ---- backward (forward direction, per-sample parameters) ------------------------------------------------
// y = x s + shift, logabsdet = sum_j log s_j  ->  dL/dx = gy s,  dL/dshift = gy,  dL/ds = gy x + gl / s,
// chained through the activation that maps the raw conditioner output to (shift, s).  What torch.autograd
// yields for coupling.py:234-252 / autoregressive.py:97-129; SURVEY section 8(f) #3.
namespace fc {

__global__ __launch_bounds__(256) void affine_backward_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                              const int32_t* __restrict__ cols,
                                                              const float* __restrict__ gy, const float* __restrict__ gl,
                                                              float* __restrict__ gx, float* __restrict__ gp, int64_t n,
                                                              int d, int d_t, int act, int rowlen) {
  const int64_t total = n * d_t;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / d_t;
    const int j = (int)(e - row * d_t);
    const int col = cols ? cols[j] : j;
    const float xv = x[row * d + col], g = gy[row * d + col], l = gl ? gl[row] : 0.f;
    const float* prow = params + row * rowlen;
    float* grow = gp + row * rowlen;
    float s = 1.f, ds = 0.f;          // scale and d s / d(raw scale value)
    int i_shift = -1, i_scale = -1;   // positions of the raw values in the row
    float dshift = 1.f;               // d shift / d(raw shift value)
    switch (act) {
      case FC_AFFINE_SIGMOID_PLUS2: {
        i_shift = j; i_scale = d_t + j;
        const float sg = sigmoidf(prow[i_scale] + 2.f);
        s = sg + 1e-3f; ds = sg * (1.f - sg);
        break;
      }
      case FC_AFFINE_SOFTPLUS_CLAMP3: {
        i_shift = j; i_scale = d_t + j;
        const float u = prow[i_scale], v = softplus1(u) + 1e-3f;
        s = clamp_like_torch(v, 0.f, 3.f);
        ds = (v >= 0.f && v <= 3.f) ? (u > 20.f ? 1.f : sigmoidf(u)) : 0.f;   // clamp passes the gradient inside [0, 3]
        break;
      }
      case FC_AFFINE_SCALE_GIVEN:
        i_shift = j; i_scale = d_t + j;
        s = prow[i_scale]; ds = 1.f;
        break;
      case FC_AFFINE_MAF_SOFTPLUS: {
        i_scale = 2 * j; i_shift = 2 * j + 1;
        const float u = prow[i_scale];
        s = softplus1(u) + 1e-3f; ds = u > 20.f ? 1.f : sigmoidf(u);
        break;
      }
      case FC_AFFINE_SCALE_SOFTPLUS: {
        i_scale = j;
        const float u = prow[i_scale];
        s = softplus1(u) + 1e-5f; ds = u > 20.f ? 1.f : sigmoidf(u);
        break;
      }
      case FC_AFFINE_SHIFT_TANH2: {
        i_shift = j;
        const float t = tanhf(prow[j]);
        dshift = 2.f * (1.f - t * t);
        break;
      }
      default:  // FC_AFFINE_ADDITIVE
        i_shift = j;
        break;
    }
    gx[row * d + col] = g * s;
    if (i_shift >= 0) grow[i_shift] = g * dshift;
    if (i_scale >= 0) grow[i_scale] = (g * xv + l / s) * ds;
  }
}

}  // namespace fc

extern "C" int fc_affine_backward(const float* x, const float* params, const int32_t* cols, const float* grad_y,
                                  const float* grad_logabsdet, float* grad_x, float* grad_params, int64_t n,
                                  int32_t d, int32_t d_t, int32_t activation, void* stream) {
  if (n < 0 || d <= 0 || d_t <= 0 || d_t > d) return hipErrorInvalidValue;
  if (activation < 0 || activation > FC_AFFINE_SCALE_SOFTPLUS) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !params || !grad_y || !grad_x || !grad_params) return hipErrorInvalidValue;
  const int rowlen = (activation == FC_AFFINE_ADDITIVE || activation == FC_AFFINE_SHIFT_TANH2 ||
                      activation == FC_AFFINE_SCALE_SOFTPLUS) ? d_t : 2 * d_t;
  const int64_t total = n * d_t;
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(fc::affine_backward_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     params, cols, grad_y, grad_logabsdet, grad_x, grad_params, n, d, d_t, activation, rowlen);
  return hipGetLastError();
}
