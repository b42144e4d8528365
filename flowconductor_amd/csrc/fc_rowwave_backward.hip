// Backward of the row-per-wavefront bijectors with batch-shared parameters: planar flow and Householder sequence.
// gfx950.  (What torch.autograd yields for flowcon/transforms/no_analytic_inv/planar.py:30-49 and
// orthogonal.py:144-194; forward kernels in fc_rowwave.hip.)
//
// One wave owns one sample row (fc_row.h).  Input gradients leave per row; parameter gradients are sums over the
// batch: every wave accumulates its rows' contributions in registers and adds them to the global result once, at the
// end, with one atomic per element and wave.
//
// Planar:  a = x.w + b, t = tanh(a), y = x + u t, lad = log(1e-7 + |psi|), psi = 1 + (1 - t^2)(u.w)
//   g_t = gy.u - gl sign(psi) / (1e-7 + |psi|) 2 t (u.w);   g_a = g_t (1 - t^2);   gx = gy + g_a w
//   gw += g_a x + c u,  gu += t gy + c w  with  c = gl sign(psi) (1 - t^2) / (1e-7 + |psi|);   gb += g_a
//
// Householder:  y = H_K ... H_1 x,  H_i v = v - (2 v.q_i / q_i.q_i) q_i  (an involution).  Walking back from the saved
// output: y_prev = H_i y restores the input of reflection i, g <- H_i g is the gradient flowing on, and with
// alpha = 2 / q.q, a = y_prev.q, c = g.q:   gq_i += -alpha (c y_prev + a g) + alpha^2 a c q_i.   gx = H_1 ... H_K gy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_math.h"
#include "fc_lane.h"
#include "fc_row.h"
#include "../../include/flowcon_hip.h"

namespace fc {

template <int E>
__global__ __launch_bounds__(256) void planar_backward_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                              const float* __restrict__ gl, const float* __restrict__ w,
                                                              const float* __restrict__ u_hat, const float* __restrict__ b_ptr,
                                                              float* __restrict__ gx, float* __restrict__ gw,
                                                              float* __restrict__ gu, float* __restrict__ gb, int64_t n, int d) {
  const float b = b_ptr[0];
  const int lane = threadIdx.x & 63;
  Row<E> wv, uv, gwv, guv;
  load_row<E>(wv, w, d, lane);
  load_row<E>(uv, u_hat, d, lane);
#pragma unroll
  for (int e = 0; e < E; ++e) gwv.v[e] = guv.v[e] = 0.f;
  float gbv = 0.f;
  const float uw = dot_rows<E>(uv, wv);
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n; row += stride) {
    Row<E> r, g;
    load_row<E>(r, x + row * d, d, lane);
    load_row<E>(g, gy + row * d, d, lane);
    const float glr = gl ? gl[row] : 0.f;
    const float a = dot_rows<E>(r, wv) + b;
    const float t = tanhf(a);
    const float dt = 1.f - t * t;
    const float psi = 1.f + dt * uw;
    const float dl = glr * (psi >= 0.f ? 1.f : -1.f) / (1e-7f + fabsf(psi));      // d lad / d psi times gl
    const float g_t = dot_rows<E>(g, uv) - dl * (2.f * t * uw);
    const float g_a = g_t * dt;
    const float c = dl * dt;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      gwv.v[e] += g_a * r.v[e] + c * uv.v[e];
      guv.v[e] += t * g.v[e] + c * wv.v[e];
      g.v[e] = g.v[e] + g_a * wv.v[e];
    }
    gbv += g_a;
    store_row<E>(g, gx + row * d, d, lane);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = lane + 64 * e;
    if (i < d) {
      atomicAdd(gw + i, gwv.v[e]);
      atomicAdd(gu + i, guv.v[e]);
    }
  }
  if (lane == 0) atomicAdd(gb, gbv);
}

// KQ: reflections whose gradient a wave keeps in registers at a time (KQ * E registers); the sequence is walked in
// chunks of KQ from the output side.
template <int E, int KQ>
__global__ __launch_bounds__(256) void householder_backward_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                                   const float* __restrict__ q, float* __restrict__ gx,
                                                                   float* __restrict__ gq, int64_t n, int d, int k_count,
                                                                   int reverse) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t row0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  // position p in application order (0 = first reflection applied) -> index into q
  auto qidx = [&](int p) { return reverse ? k_count - 1 - p : p; };
  for (int hi = k_count; hi > 0; hi -= KQ) {        // reflections [lo, hi) in application order, last chunk first
    const int lo = hi - KQ > 0 ? hi - KQ : 0;
    Row<E> acc[KQ];
#pragma unroll
    for (int k = 0; k < KQ; ++k)
#pragma unroll
      for (int e = 0; e < E; ++e) acc[k].v[e] = 0.f;
    for (int64_t row = row0; row < n; row += stride) {
      Row<E> v, g;
      load_row<E>(v, y + row * d, d, lane);
      load_row<E>(g, gy + row * d, d, lane);
      // undo the reflections after this chunk (positions k_count - 1 .. hi): values and gradients
      for (int p = k_count - 1; p >= hi; --p) {
        Row<E> qv;
        load_row<E>(qv, q + (int64_t)qidx(p) * d, d, lane);
        const float alpha = 2.f / dot_rows<E>(qv, qv);
        const float a = dot_rows<E>(v, qv), c = dot_rows<E>(g, qv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          v.v[e] -= a * (alpha * qv.v[e]);
          g.v[e] -= c * (alpha * qv.v[e]);
        }
      }
#pragma unroll
      for (int k = KQ - 1; k >= 0; --k) {
        const int p = lo + k;
        if (p < hi) {
          Row<E> qv;
          load_row<E>(qv, q + (int64_t)qidx(p) * d, d, lane);
          const float alpha = 2.f / dot_rows<E>(qv, qv);
          const float a_out = dot_rows<E>(v, qv), c = dot_rows<E>(g, qv);
          // y_prev = H v: y_prev.q = a_out - alpha a_out (q.q) = -a_out
          const float a = -a_out;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            v.v[e] -= a_out * (alpha * qv.v[e]);        // v = y_prev
            acc[k].v[e] += -alpha * (c * v.v[e] + a * g.v[e]) + (alpha * alpha) * (a * c) * qv.v[e];
            g.v[e] -= c * (alpha * qv.v[e]);            // gradient flowing to the previous reflection
          }
        }
      }
      if (lo == 0) store_row<E>(g, gx + row * d, d, lane);
    }
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
      const int p = lo + k;
      if (p < hi) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = lane + 64 * e;
          if (i < d) atomicAdd(gq + (int64_t)qidx(p) * d + i, acc[k].v[e]);
        }
      }
    }
  }
}

// Middle of the Sylvester backward (no_analytic_inv/planar.py:144-166): with pre = R1 Q^T z + b, act = tanh(pre),
// diag = 1 + (1 - act^2) rd (rd = diag R1 diag R2) and lad = sum_j log diag_j, turn the gradient wrt act (from the
// R2 / Q product above it) into the gradient wrt pre, adding the log-determinant's share, and reduce the two
// batch sums the parameter gradients need:
//   g_act += gl rd (-2 act) / diag;   g_pre = g_act (1 - act^2)   (in place);   g_bias += g_pre;   g_rd += gl (1 - act^2) / diag
template <int E>
__global__ __launch_bounds__(256) void sylvester_mid_backward_kernel(const float* __restrict__ pre, float* __restrict__ g,
                                                                     const float* __restrict__ gl,
                                                                     const float* __restrict__ rd, float* __restrict__ g_bias,
                                                                     float* __restrict__ g_rd, int64_t n, int d) {
  const int lane = threadIdx.x & 63;
  Row<E> rdv, sb, sr;
  load_row<E>(rdv, rd, d, lane);
#pragma unroll
  for (int e = 0; e < E; ++e) sb.v[e] = sr.v[e] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n; row += stride) {
    Row<E> p, gv;
    load_row<E>(p, pre + row * d, d, lane);
    load_row<E>(gv, g + row * d, d, lane);
    const float glr = gl ? gl[row] : 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float act = tanhf(p.v[e]);
      const float dt = 1.f - act * act;
      const float gd = glr / (1.f + dt * rdv.v[e]);
      const float gp = (gv.v[e] + gd * rdv.v[e] * (-2.f * act)) * dt;
      gv.v[e] = gp;
      sb.v[e] += gp;
      sr.v[e] += gd * dt;
    }
    store_row<E>(gv, g + row * d, d, lane);
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = lane + 64 * e;
    if (i < d) {
      atomicAdd(g_bias + i, sb.v[e]);
      atomicAdd(g_rd + i, sr.v[e]);
    }
  }
}

inline unsigned bwd_row_grid(int64_t n) {
  int64_t g = (n + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t cap = 256 * 4;
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace fc

#define FC_ROW_DISPATCH_B(D, CALL)                \
  switch ((D) <= 64 ? 1 : (D) <= 128 ? 2 : (D) <= 256 ? 4 : 8) { \
    case 1: { constexpr int E = 1; CALL; break; } \
    case 2: { constexpr int E = 2; CALL; break; } \
    case 4: { constexpr int E = 4; CALL; break; } \
    default: { constexpr int E = 8; CALL; break; } \
  }

extern "C" int fc_planar_backward(const float* x, const float* grad_y, const float* grad_logabsdet, const float* w,
                                  const float* u_hat, const float* b, float* grad_x, float* grad_w, float* grad_u_hat,
                                  float* grad_b, int64_t n, int32_t d, void* stream) {
  if (n < 0 || d <= 0 || d > 512) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !grad_y || !w || !u_hat || !b || !grad_x || !grad_w || !grad_u_hat || !grad_b) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  FC_ROW_DISPATCH_B(d, hipLaunchKernelGGL(fc::planar_backward_kernel<E>, dim3(fc::bwd_row_grid(n)), dim3(256), 0, s, x,
                                          grad_y, grad_logabsdet, w, u_hat, b, grad_x, grad_w, grad_u_hat, grad_b, n, d));
  return hipGetLastError();
}

extern "C" int fc_householder_backward(const float* y, const float* grad_y, const float* q, float* grad_x, float* grad_q,
                                       int64_t n, int32_t d, int32_t num_transforms, int32_t reverse, void* stream) {
  if (n < 0 || d <= 0 || d > 512 || num_transforms < 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!y || !grad_y || !grad_x || (num_transforms > 0 && (!q || !grad_q))) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (num_transforms == 0) return hipMemcpyAsync(grad_x, grad_y, sizeof(float) * n * d, hipMemcpyDeviceToDevice, s);
  FC_ROW_DISPATCH_B(d, hipLaunchKernelGGL((fc::householder_backward_kernel<E, 8>), dim3(fc::bwd_row_grid(n)), dim3(256), 0,
                                          s, y, grad_y, q, grad_x, grad_q, n, d, num_transforms, reverse));
  return hipGetLastError();
}

extern "C" int fc_sylvester_mid_backward(const float* pre, float* grad_act_inout, const float* grad_logabsdet,
                                         const float* r_diag_prod, float* grad_bias, float* grad_r_diag_prod, int64_t n,
                                         int32_t d, void* stream) {
  if (n < 0 || d <= 0 || d > 512) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!pre || !grad_act_inout || !r_diag_prod || !grad_bias || !grad_r_diag_prod) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  FC_ROW_DISPATCH_B(d, hipLaunchKernelGGL(fc::sylvester_mid_backward_kernel<E>, dim3(fc::bwd_row_grid(n)), dim3(256), 0, s,
                                          pre, grad_act_inout, grad_logabsdet, r_diag_prod, grad_bias, grad_r_diag_prod, n, d));
  return hipGetLastError();
}
