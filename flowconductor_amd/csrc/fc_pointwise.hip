// Batch-shared point-wise affine map: y = x * scale + shift, or (x - shift) / scale.
//
// Restates PointwiseAffineTransform.forward/inverse (flowcon/transforms/standard.py:54-68),
// ActNorm.forward/inverse (flowcon/transforms/normalization.py:171-204, scale = exp(log_scale))
// and the eval-mode BatchNorm map (normalization.py:98-141).  The per-call constant
// logabsdet is a host-side scalar in the reference and stays one here.
// Streaming kernel: 16-byte loads/stores, x read once, y written once (8 B per element).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flowcon_hip.h"

namespace fc {

// mode 0: y = x*s + t          1: y = (x - t)/s
// mode 2: y = w*((x - mean)/sd) + b   with a = mean, s = sd, w, t = b   (BatchNorm eval forward)
// mode 3: y = sd*((x - b)/w) + mean                                       (BatchNorm eval inverse)
template <int kMode>
__device__ __forceinline__ float apply(float x, float s, float t, float a, float w) {
  if (kMode == 0) return x * s + t;
  if (kMode == 1) return (x - t) / s;
  if (kMode == 2) return w * ((x - a) / s) + t;
  return s * ((x - t) / w) + a;
}

template <int kMode>
__global__ __launch_bounds__(256) void pointwise_kernel(const float* __restrict__ x,
                                                        float* __restrict__ y,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const float* __restrict__ aux_a,
                                                        const float* __restrict__ aux_w,
                                                        int64_t total, int64_t m, int scale_len1,
                                                        int shift_len1, int vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec && ((stride << 2) % m) == 0) {
    // the grid stride is a whole number of items: a thread always meets the same 4 positions of the item, so its
    // parameters are read once and the loop is load - 4 fma - store (no 64-bit modulo per element)
    const int64_t nvec = total >> 2;
    const int64_t j = (gid << 2) % m;
    float s4[4], t4[4], a4[4], w4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s4[k] = scale[scale_len1 ? 0 : j + k];
      t4[k] = shift[shift_len1 ? 0 : j + k];
      a4[k] = aux_a ? aux_a[j + k] : 0.f;
      w4[k] = aux_w ? aux_w[j + k] : 1.f;
    }
    for (int64_t i = gid; i < nvec; i += stride) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 o;
      o.x = apply<kMode>(v.x, s4[0], t4[0], a4[0], w4[0]);
      o.y = apply<kMode>(v.y, s4[1], t4[1], a4[1], w4[1]);
      o.z = apply<kMode>(v.z, s4[2], t4[2], a4[2], w4[2]);
      o.w = apply<kMode>(v.w, s4[3], t4[3], a4[3], w4[3]);
      reinterpret_cast<float4*>(y)[i] = o;
    }
  } else if (vec) {
    const int64_t nvec = total >> 2;
    for (int64_t i = gid; i < nvec; i += stride) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      const int64_t j = (i << 2) % m;  // m % 4 == 0 in vec mode, so the 4 lanes stay in one item
      float4 o;
      const float* vi = reinterpret_cast<const float*>(&v);
      float* vo = reinterpret_cast<float*>(&o);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float s = scale[scale_len1 ? 0 : j + k];
        const float t = shift[shift_len1 ? 0 : j + k];
        const float a = aux_a ? aux_a[j + k] : 0.f;
        const float w = aux_w ? aux_w[j + k] : 1.f;
        vo[k] = apply<kMode>(vi[k], s, t, a, w);
      }
      reinterpret_cast<float4*>(y)[i] = o;
    }
  } else {
    for (int64_t e = gid; e < total; e += stride) {
      const int64_t j = e % m;
      const float s = scale[scale_len1 ? 0 : j];
      const float t = shift[shift_len1 ? 0 : j];
      const float a = aux_a ? aux_a[j] : 0.f;
      const float w = aux_w ? aux_w[j] : 1.f;
      y[e] = apply<kMode>(x[e], s, t, a, w);
    }
  }
}

}  // namespace fc

extern "C" int fc_pointwise_affine(const float* x, float* y, const float* scale, const float* shift,
                                   const float* aux_mean, const float* aux_weight, int64_t n,
                                   int64_t m, int32_t scale_len, int32_t shift_len, int32_t mode,
                                   void* stream) {
  if (n < 0 || m <= 0 || mode < 0 || mode > 3) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !scale || !shift) return hipErrorInvalidValue;
  if ((scale_len != 1 && scale_len != m) || (shift_len != 1 && shift_len != m)) return hipErrorInvalidValue;
  if (mode >= 2 && (!aux_mean || !aux_weight || scale_len != m || shift_len != m)) return hipErrorInvalidValue;
  const int64_t total = n * m;
  const int vec = (m % 4 == 0) && ((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0;
  int64_t work = vec ? total / 4 : total;
  int64_t grid = (work + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  if (grid < 1) grid = 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  dim3 g((unsigned)grid), b(256);
  const int sl1 = scale_len == 1 && m != 1, tl1 = shift_len == 1 && m != 1;
#define FC_LAUNCH(MODE)                                                                          \
  hipLaunchKernelGGL(fc::pointwise_kernel<MODE>, g, b, 0, s, x, y, scale, shift, aux_mean,       \
                     aux_weight, total, m, sl1, tl1, vec)
  switch (mode) {
    case 0: FC_LAUNCH(0); break;
    case 1: FC_LAUNCH(1); break;
    case 2: FC_LAUNCH(2); break;
    default: FC_LAUNCH(3); break;
  }
#undef FC_LAUNCH
  return hipGetLastError();
}
