// Backward of the sum-of-sigmoids bijector (+ extended softplus), forward direction, gfx950.
//
// For y, logabsdet = sum_of_sigmoids(x, raw) (flowcon/transforms/adaptive_sigmoids.py:108-142 with
// nonlinearities.py:519-552; fc_sos.hip) and upstream gradients gy [N, D], gl [N]:
//     grad_x[n, j]          = gy dy/dx + gl dlad/dx
//     grad_raw[n, j, 3S+1]  = gy dy/draw + gl dlad/draw         (what torch.autograd yields for the reference's ops)
//
// With  s_k = 10 tanh(a_k),  alpha_k = 0.1 + 9.9 sigmoid(b_k),  n_k = (softmax(c)_k + 1e-6) / (1 + 1e-6 S),  w_k = post n_k,
// sh = softplus(e) + 0.1,  pre_k = alpha_k (x - s_k),  sg_k = sigmoid(pre_k),  sg'_k = sg_k (1 - sg_k),  W = sum_k w_k:
//     y    = sum_k w_k sg_k / W + softplus(x - sh) - softplus(-(x + sh)) - offset
//     lad  = log(D_sos + D_esp),   D_sos = sum_k w_k alpha_k sg'_k,   D_esp = sigmoid(x - sh) + sigmoid(-(x + sh))
// every derivative is closed form; the chain to the raw values goes through tanh' = 1 - tanh^2, the sigmoid's slope
// (alpha - 0.1)(10 - alpha) / 9.9, the renormalised softmax and softplus' = 1 - exp(-(sh - 0.1)).
//
// One lane per (sample, dim), four passes over the S sigmoids recomputing the derived parameters from the raw row (no
// per-thread arrays: S is a run-time value); the rows travel through LDS (coalesced in, coalesced out), see
// sos_backward_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_device.h"
#include "fc_math.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct SoSBwdArgs {
  const float* x;        // [N, D]
  const float* raw;      // [N, D, 3S+1]
  const float* gy;       // [N, D]
  const float* gl;       // [N] or null
  float* gx;             // [N, D]
  float* graw;           // [N, D, 3S+1]
  int64_t total;         // N * D
  int D, S;
  float post;            // exp(log_scale_postact)
};


// One element: r -> the P raw values, g -> where the P gradients go (may alias r: every slot is read before it is
// written, except the softmax logits, whose probabilities are kept in `sm` [S] until the last loop).  Returns grad_x.
// kSmKept: `sm` is storage of its own (kept to the end); otherwise it IS g's logit slots and the last loop recomputes.
template <bool kSmKept>
__device__ __forceinline__ float sos_backward_element(const SoSBwdArgs& a, const float* r, float* g, float* sm, float x,
                                                      float gy, float gl) {
  const int S = a.S;
  // pass 1: softmax normalisation of the weight logits
  float m = -INFINITY;
  for (int k = 0; k < S; ++k) m = fmaxf(m, r[2 * S + k]);
  float zsum = 0.f;
  for (int k = 0; k < S; ++k) {
    const float z = exp_lean(r[2 * S + k] - m);
    sm[k] = z;
    zsum += z;
  }
  const float rz = div_lean(1.f, zsum);
  const float tot = 1.f + 1e-6f * (float)S, rtot = div_lean(1.f, tot);
  // pass 2: y_sos, D_sos, W
  float ynum = 0.f, dsos = 0.f, wsum = 0.f;
  for (int k = 0; k < S; ++k) {
    const float sk = 10.f * tanh_lean(r[k]);
    const float al = 0.1f + 9.9f * sigmoid_lean(r[S + k]);
    const float w = a.post * ((sm[k] * rz + 1e-6f) * rtot);
    const float sg = sigmoid_lean(al * (x - sk));
    ynum += w * sg;
    dsos += w * al * (sg * (1.f - sg));
    wsum += w;
  }
  const float rw = div_lean(1.f, wsum);
  const float ysos = ynum * rw;
  const float sh = softplus_lean(r[3 * S], 1.f) + 0.1f;
  const float su = sigmoid_lean(x - sh), sv = sigmoid_lean(-(x + sh));
  const float desp = su + sv;
  const float gD = gl * div_lean(1.f, dsos + desp);       // d lad / d D_sos = d lad / d D_esp
  const float dsu = su * (1.f - su), dsv = sv * (1.f - sv);
  // pass 3: per-sigmoid adjoints; the gradient of the normalised weight n_k is parked in the logit slot
  float gxs = 0.f, gn_n = 0.f, gn_sm = 0.f;
  for (int k = 0; k < S; ++k) {
    const float th = tanh_lean(r[k]);
    const float sk = 10.f * th;
    const float sb = sigmoid_lean(r[S + k]);
    const float al = 0.1f + 9.9f * sb;
    const float smk = sm[k] * rz;
    const float nk = (smk + 1e-6f) * rtot;
    const float w = a.post * nk;
    const float sg = sigmoid_lean(al * (x - sk));
    const float d1 = sg * (1.f - sg);
    const float g_pre = gy * (w * d1 * rw) + gD * (w * al * d1 * (1.f - 2.f * sg));
    const float g_al = g_pre * (x - sk) + gD * (w * d1);
    const float g_w = gy * ((sg - ysos) * rw) + gD * (al * d1);
    gxs += g_pre * al;
    g[k] = (-g_pre * al) * (10.f * (1.f - th * th));           // d s / d a = 10 (1 - tanh^2)
    g[S + k] = g_al * (9.9f * sb * (1.f - sb));                 // d alpha / d b
    const float g_n = a.post * g_w;                             // w = post n
    g[2 * S + k] = g_n;
    gn_n += g_n * nk;
    gn_sm += g_n * smk;
  }
  // n_k = (sm_k + 1e-6) / tot with tot = sum_k (sm_k + 1e-6):  g_wt_j = (g_n_j - sum_k g_n_k n_k) / tot, then the softmax
  const float bsum = (gn_sm - gn_n) * rtot;                     // sum_k g_wt_k sm_k   (sum_k sm_k = 1)
  for (int k = 0; k < S; ++k) {
    const float g_wt = (g[2 * S + k] - gn_n) * rtot;
    const float z = kSmKept ? sm[k] : exp_lean(r[2 * S + k] - m);
    g[2 * S + k] = (z * rz) * (g_wt - bsum);
  }
  // extended softplus: y_esp = sp(x - sh) - sp(-(x + sh)),  D_esp = su + sv
  const float g_sh = gy * (sv - su) - gD * (dsu + dsv);
  g[3 * S] = g_sh * (1.f - exp_lean(-(sh - 0.1f)));              // d sh / d e = sigmoid(e) = 1 - exp(-softplus(e))
  return gxs + gy * desp + gD * (dsu - dsv);
}

// A wave owns groups of 64 consecutive elements: their raw rows (64 x P floats, contiguous in memory) come into LDS with
// coalesced loads, every lane then works on ITS element's row in place (row stride odd: conflict-free whatever k the
// lanes are at; S extra slots per row hold the softmax numerators), and the gradient rows leave coalesced.  [One
// thread per element straight on global memory reads its 4 (3S + 1)-byte row with a stride of as many bytes between
// lanes: 12.7 ms per 2^18 x 8 elements at S = 30 against 1.4 ms for the forward.]
__global__ __launch_bounds__(256) void sos_backward_kernel(SoSBwdArgs a, int waves_per_block, int ts) {
  extern __shared__ float sos_smem[];
  const int S = a.S, P = 3 * S + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= waves_per_block) return;
  float* tile = sos_smem + (size_t)wave * 64 * ts;
  const int64_t groups = (a.total + 63) / 64;
  for (int64_t grp = (int64_t)blockIdx.x * waves_per_block + wave; grp < groups; grp += (int64_t)gridDim.x * waves_per_block) {
    const int64_t e0 = grp * 64;
    const int cnt = a.total - e0 < 64 ? (int)(a.total - e0) : 64;
    for (int e = 0; e < cnt; ++e) {
      const float* src = a.raw + (e0 + e) * P;
      for (int k = lane; k < P; k += 64) tile[e * ts + k] = src[k];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < cnt) {
      const int64_t el = e0 + lane;
      float* row = tile + lane * ts;
      a.gx[el] = sos_backward_element<true>(a, row, row, row + P, a.x[el], a.gy[el], a.gl ? a.gl[el / a.D] : 0.f);
    }
    __builtin_amdgcn_wave_barrier();
    for (int e = 0; e < cnt; ++e) {
      float* dst = a.graw + (e0 + e) * P;
      for (int k = lane; k < P; k += 64) dst[k] = tile[e * ts + k];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// rows too long for the LDS tiles: one thread per element on global memory, the softmax numerators recomputed
__global__ __launch_bounds__(256) void sos_backward_direct_kernel(SoSBwdArgs a) {
  const int S = a.S, P = 3 * S + 1;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.total; e += (int64_t)gridDim.x * blockDim.x) {
    const float* r = a.raw + e * P;
    float* g = a.graw + e * P;
    // (g is a different array here: its logit slots can hold the numerators until pass 3 overwrites them one by one)
    a.gx[e] = sos_backward_element<false>(a, r, g, g + 2 * S, a.x[e], a.gy[e], a.gl ? a.gl[e / a.D] : 0.f);
  }
}

}  // namespace fc

extern "C" int fc_sum_of_sigmoids_backward(const float* x, const float* params, const float* grad_y,
                                           const float* grad_logabsdet, float* grad_x, float* grad_params, int64_t n,
                                           int32_t d, int32_t n_sigmoids, float log_scale_postact, void* stream) {
  if (n < 0 || d <= 0 || n_sigmoids <= 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !params || !grad_y || !grad_x || !grad_params) return hipErrorInvalidValue;
  fc::SoSBwdArgs a{x, params, grad_y, grad_logabsdet, grad_x, grad_params, n * (int64_t)d, d, n_sigmoids,
                   expf(log_scale_postact)};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int P = 3 * n_sigmoids + 1;
  int ts = P + n_sigmoids;           // row + softmax numerators, odd stride (conflict-free per-lane rows)
  if ((ts & 1) == 0) ++ts;
  const size_t wave_bytes = (size_t)64 * ts * sizeof(float);
  const int per_cu = (int)((size_t)(160 * 1024) / wave_bytes);     // single-wave workgroups: as many as the LDS of a CU holds
  if (per_cu >= 1) {
    static fc::PerDeviceOnce attr;
    const hipError_t ea = fc::ensure_max_dynamic_lds(attr, reinterpret_cast<const void*>(&fc::sos_backward_kernel), 160 * 1024);
    if (ea != hipSuccess) return ea;
    int64_t grid = (a.total + 63) / 64;
    const int64_t cap = (int64_t)fc::device_cu_count() * per_cu;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(fc::sos_backward_kernel, dim3((unsigned)grid), dim3(64), wave_bytes, s, a, 1, ts);
    return hipGetLastError();
  }
  int64_t grid = (a.total + 255) / 256;
  const int64_t cap = (int64_t)fc::device_cu_count() * 16;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(fc::sos_backward_direct_kernel, dim3((unsigned)grid), dim3(256), 0, s, a);
  return hipGetLastError();
}
