// Row-per-wavefront bijectors with batch-shared (or per-sample) dense parameters, gfx950.
//
// One 64-lane wave owns one sample row; lane l holds elements l, l+64, l+128, ... of the row in
// registers (E <= 8 registers, D <= 512).  Dot products are 64-lane butterflies (__shfl_xor),
// mat-vecs broadcast x_j with v_readlane and FMA a contiguous weight column per lane, so the row
// never leaves registers between the fused stages.  HBM traffic is the minimum: the row in, the
// row out, one logabsdet word -- parameters are a few KB and live in L1/L2.
//
// Restates (not copies):
//   flowcon/transforms/orthogonal.py:144-194       K Householder reflections
//   flowcon/transforms/no_analytic_inv/planar.py:30-69     planar flow + constrained u
//   flowcon/transforms/no_analytic_inv/planar.py:144-166   Sylvester flow (Householder Q, tanh)
//   flowcon/transforms/lu.py:56-91                 LU linear forward / triangular-solve inverse
//   flowcon/transforms/linear.py:45-76             cached dense weight / inverse path
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_math.h"
#include "fc_lane.h"
#include "fc_row.h"
#include "../../include/flowcon_hip.h"

namespace fc {

// out -= (out . q) * (2 / |q|^2) * q, K times (orthogonal.py:144-171).
// q: [K, d] shared, or this row's [K, d] block when per-sample.  order: 0..K-1 or reversed.
template <int E, bool kLds = false>
__device__ __forceinline__ void householder(Row<E>& x, const float* __restrict__ q, int k_count, int d,
                                            int lane, bool reverse) {
  for (int t = 0; t < k_count; ++t) {
    const int k = reverse ? k_count - 1 - t : t;
    Row<E> qv;
    load_row<E>(qv, q + (int64_t)k * d, d, lane);
    const float sq = dot_rows<E, kLds>(qv, qv);
    const float ip = dot_rows<E, kLds>(x, qv);
    const float c = 2.f / sq;
#pragma unroll
    for (int e = 0; e < E; ++e) x.v[e] = x.v[e] - ip * (c * qv.v[e]);
  }
}

// y_i = sum_j W[i][j] x_j with Wt = W^T stored [d_in][d_out] (lane i reads a contiguous column)
template <int E>
__device__ __forceinline__ void matvec(Row<E>& y, const Row<E>& x, const float* __restrict__ wt, int d,
                                       int lane, int j_lo_is_i /*1: upper-triangular W (skip j<i)*/) {
#pragma unroll
  for (int e = 0; e < E; ++e) y.v[e] = 0.f;
  for (int j = 0; j < d; ++j) {
    const float xj = bcast<E>(x, j);
    const float* col = wt + (int64_t)j * d;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = lane + 64 * e;
      if (i < d) y.v[e] += col[i] * xj;
    }
  }
}

// y_i = sum_{j >= i} R[i][j] x_j for a per-sample upper-triangular R stored row-major as the hyper-network emits it:
// row i is one coalesced read of its columns j >= i only (the strictly lower part is never fetched), one multiply
// per lane and a wave reduction.
template <int E>
__device__ __forceinline__ void matvec_rows_upper(Row<E>& y, const Row<E>& x, const float* __restrict__ r, int d,
                                                  int lane) {
#pragma unroll
  for (int e = 0; e < E; ++e) y.v[e] = 0.f;
  // Eight rows per step, the NEXT eight requested before the current eight are reduced (one step at a time the wave
  // waits out a full HBM latency per step: 16 per mat-vec at D = 128), and their eight sums taken together
  // (wave64_sum8: a third of the cross-lane steps of eight separate reductions).
  constexpr int kRows = 8;
  auto request = [&](int i0, float (&w)[kRows][E]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
      const int i = i0 + k;
      const float* row = r + (int64_t)(i < d ? i : d - 1) * d;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = lane + 64 * e;
        w[k][e] = (j >= i && j < d && i < d) ? row[j] : 0.f;
      }
    }
  };
  auto reduce = [&](int i0, const float (&w)[kRows][E]) __attribute__((always_inline)) {
    float s[kRows], tot[kRows];
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
      s[k] = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s[k] += w[k][e] * x.v[e];
    }
    wave64_sum8(s, tot);
#pragma unroll
    for (int k = 0; k < kRows; ++k)
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (lane + 64 * e == i0 + k) y.v[e] = tot[k];
  };
  float wa[kRows][E], wb[kRows][E];
  request(0, wa);
  for (int i0 = 0; i0 < d; i0 += 2 * kRows) {
    request(i0 + kRows, wb);
    __builtin_amdgcn_sched_barrier(0);
    reduce(i0, wa);
    request(i0 + 2 * kRows, wa);
    __builtin_amdgcn_sched_barrier(0);
    reduce(i0 + kRows, wb);
  }
}

// The K reflections of a PER-SAMPLE q [K, d] (this row's block, from HBM): as householder() above, with the q rows
// requested eight at a time and one batch ahead, and the eight |q|^2 of a batch in one batched reduction; only the
// (x . q_k) chain is sequential.
template <int E>
__device__ __forceinline__ void householder_rows(Row<E>& x, const float* __restrict__ q, int k_count, int d, int lane,
                                                 bool reverse) {
  constexpr int kB = 8;
  auto request = [&](int t0, Row<E> (&qv)[kB]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < kB; ++j) {
      const int t = t0 + j < k_count ? t0 + j : k_count - 1;
      load_row<E>(qv[j], q + (int64_t)(reverse ? k_count - 1 - t : t) * d, d, lane);
    }
  };
  auto apply = [&](int t0, const Row<E> (&qv)[kB]) __attribute__((always_inline)) {
    float s[kB], sq[kB];
#pragma unroll
    for (int j = 0; j < kB; ++j) {
      s[j] = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) s[j] += qv[j].v[e] * qv[j].v[e];
    }
    wave64_sum8(s, sq);
#pragma unroll
    for (int j = 0; j < kB; ++j)
      if (t0 + j < k_count) {
        const float ip = dot_rows<E>(x, qv[j]);
        const float c = div_lean(2.f, sq[j]);      // (v_rcp + one correction, <= 1 ulp: the IEEE division costs ten instructions)
#pragma unroll
        for (int e = 0; e < E; ++e) x.v[e] = x.v[e] - ip * (c * qv[j].v[e]);
      }
  };
  if (k_count <= 0) return;      // no reflections: the first request below would index row -1 (q may be NULL then)
  Row<E> qa[kB], qb[kB];
  request(0, qa);
  for (int t0 = 0; t0 < k_count; t0 += 2 * kB) {
    request(t0 + kB, qb);
    __builtin_amdgcn_sched_barrier(0);
    apply(t0, qa);
    request(t0 + 2 * kB, qa);
    __builtin_amdgcn_sched_barrier(0);
    apply(t0 + kB, qb);
  }
}

// ---- kernels ----------------------------------------------------------------------------

template <int E>
__global__ __launch_bounds__(256) void householder_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ q, int64_t n, int d,
                                                          int k_count, int per_sample, int reverse) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n; row += stride) {
    Row<E> r;
    load_row<E>(r, x + row * d, d, lane);
    const float* qr = per_sample ? q + row * (int64_t)k_count * d : q;
    if (per_sample) householder_rows<E>(r, qr, k_count, d, lane, reverse != 0);     // q from HBM: batched requests
    else householder<E>(r, qr, k_count, d, lane, reverse != 0);
    store_row<E>(r, y + row * d, d, lane);
  }
}

// Rows of <= 16 features (the low-dimensional conditional flows): four samples per wave, one per 16-lane DPP row, so
// all 64 lanes work and a dot product is four DPP row rotations (no cross-row step at all).
__global__ __launch_bounds__(256) void householder_narrow_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 const float* __restrict__ q, int64_t n, int d,
                                                                 int k_count, int per_sample, int reverse) {
  const int lane = threadIdx.x & 63, j = lane & 15;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * 4;
  for (int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * 4; base < n; base += stride) {
    const int64_t row = base + (lane >> 4);
    const bool live = row < n && j < d;
    const int64_t rr = row < n ? row : n - 1;
    float v = live ? x[rr * d + j] : 0.f;
    const float* qr = per_sample ? q + rr * (int64_t)k_count * d : q;
    // the q rows of a sample do not depend on v: four in flight at a time
    for (int t0 = 0; t0 < k_count; t0 += 4) {
      float qv[4], c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u < k_count ? t0 + u : k_count - 1;
        qv[u] = j < d ? qr[(int64_t)(reverse ? k_count - 1 - t : t) * d + j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = t0 + u < k_count ? 2.f / row16_allsum(qv[u] * qv[u]) : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) v -= row16_allsum(v * qv[u]) * (c[u] * qv[u]);
    }
    if (live) y[row * d + j] = v;
  }
}

template <int E>
__global__ __launch_bounds__(256) void planar_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     float* __restrict__ lad, const float* __restrict__ w,
                                                     const float* __restrict__ u_hat, const float* __restrict__ b_ptr,
                                                     int64_t n, int d, int per_sample) {
  float b = b_ptr[0];
  const int lane = threadIdx.x & 63;
  Row<E> wv, uv;
  load_row<E>(wv, w, d, lane);
  load_row<E>(uv, u_hat, d, lane);
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n; row += stride) {
    Row<E> r;
    load_row<E>(r, x + row * d, d, lane);
    if (per_sample) {
      load_row<E>(wv, w + row * d, d, lane);
      load_row<E>(uv, u_hat + row * d, d, lane);
      b = b_ptr[row];
    }
    const float a = dot_rows<E>(r, wv) + b;   // mm(inputs, w.T) + b
    const float t = tanhf(a);
    const float dt = 1.f - t * t;
    // abs_det = |1 + sum_j u_j * ((1 - tanh^2 a) * w_j)|  (planar.py:43-48)
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      s += uv.v[e] * (dt * wv.v[e]);
      r.v[e] = r.v[e] + uv.v[e] * t;
    }
    s = wave_sum(s);
    store_row<E>(r, y + row * d, d, lane);
    if (lane == 0 && lad) lad[row] = logf(1e-7f + fabsf(1.f + s));
  }
}

// The same map for rows of 4 L' <= 4 L floats (16-byte aligned): L lanes carry a row as float4 pieces, a wave 64 / L rows
// (D = 64: four rows per wave, two 4-step butterflies instead of two 6-step ones per ROW, 16-byte requests), the next group's
// rows requested before this group's arithmetic (round 4: planar_kernel<1> ran at 0.39 of the HBM peak, 68 % of its wave
// cycles waiting on memory with one 256-byte request in flight per wave).
template <int L>
__global__ __launch_bounds__(256) void planar_rows4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           float* __restrict__ lad, const float* __restrict__ w,
                                                           const float* __restrict__ u_hat, const float* __restrict__ b_ptr,
                                                           int64_t n, int d, int per_sample) {
  constexpr int kRows = 64 / L;
  const int lane = threadIdx.x & 63, sub = lane % L, rw = lane / L;
  const int d4 = d >> 2;
  const bool live = sub < d4;
  const float4 zero4 = float4{0.f, 0.f, 0.f, 0.f};
  float b = b_ptr[0];
  float4 wv = (live && !per_sample) ? reinterpret_cast<const float4*>(w)[sub] : zero4;
  float4 uv = (live && !per_sample) ? reinterpret_cast<const float4*>(u_hat)[sub] : zero4;
  const int64_t groups = (n + kRows - 1) / kRows;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  int64_t grp = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  auto fetch = [&](int64_t g, const float* src) {
    const int64_t row = g * kRows + rw;
    return (live && row < n) ? reinterpret_cast<const float4*>(src + row * d)[sub] : zero4;
  };
  float4 rnext = grp < groups ? fetch(grp, x) : zero4;
  for (; grp < groups; grp += stride) {
    const int64_t row = grp * kRows + rw;
    const float4 r = rnext;
    if (grp + stride < groups) rnext = fetch(grp + stride, x);
    if (per_sample) {
      wv = fetch(grp, w);
      uv = fetch(grp, u_hat);
      b = b_ptr[row < n ? row : n - 1];
    }
    float a = (r.x * wv.x + r.y * wv.y) + (r.z * wv.z + r.w * wv.w);
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, L);
    a += b;                                      // mm(inputs, w.T) + b
    const float t = tanhf(a);
    const float dt = 1.f - t * t;
    // abs_det = |1 + sum_j u_j * ((1 - tanh^2 a) * w_j)|  (planar.py:43-48)
    float s = (uv.x * (dt * wv.x) + uv.y * (dt * wv.y)) + (uv.z * (dt * wv.z) + uv.w * (dt * wv.w));
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, L);
    if (live && row < n)
      reinterpret_cast<float4*>(y + row * d)[sub] = float4{r.x + uv.x * t, r.y + uv.y * t, r.z + uv.z * t, r.w + uv.w * t};
    if (sub == 0 && row < n && lad) lad[row] = logf(1e-7f + fabsf(1.f + s));
  }
}

// mode 0: y = W x + bias (Wt given)               -- linear.py:45-52 cached path, lu.py:56-68
// mode 1: y = L (U x) + bias (Ut, Lt given)        -- lu.py:56-68 (two F.linear)
// mode 2: y = U^-1 L^-1 (x - bias), L unit-lower   -- lu.py:70-91 (two solve_triangular)
template <int E>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     const float* __restrict__ at, const float* __restrict__ bt,
                                                     const float* __restrict__ bias, int64_t n, int d, int mode) {
  const int lane = threadIdx.x & 63;
  Row<E> bv;
  if (bias) load_row<E>(bv, bias, d, lane);
  else {
#pragma unroll
    for (int e = 0; e < E; ++e) bv.v[e] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n; row += stride) {
    Row<E> r, t;
    load_row<E>(r, x + row * d, d, lane);
    if (mode == 0) {
      matvec<E>(t, r, at, d, lane, 0);
#pragma unroll
      for (int e = 0; e < E; ++e) r.v[e] = t.v[e] + bv.v[e];
    } else if (mode == 1) {
      matvec<E>(t, r, at, d, lane, 1);   // U x
      matvec<E>(r, t, bt, d, lane, 0);   // L (U x)
#pragma unroll
      for (int e = 0; e < E; ++e) r.v[e] = r.v[e] + bv.v[e];
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) r.v[e] = r.v[e] - bv.v[e];
      // forward substitution with unit-lower L: bt = L^T, i.e. bt[j*d + i] = L[i][j]
      for (int j = 0; j < d; ++j) {
        const float xj = bcast<E>(r, j);
        const float* col = bt + (int64_t)j * d;  // column j of L
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = lane + 64 * e;
          if (i > j && i < d) r.v[e] -= col[i] * xj;
        }
      }
      // back substitution with U: column j of U
      for (int j = d - 1; j >= 0; --j) {
        const float* col = at + (int64_t)j * d;
        const float ujj = col[j];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = lane + 64 * e;
          if (i == j) r.v[e] = r.v[e] / ujj;
        }
        const float xj = bcast<E>(r, j);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = lane + 64 * e;
          if (i < j) r.v[e] -= col[i] * xj;
        }
      }
    }
    store_row<E>(r, y + row * d, d, lane);
  }
}

// Sylvester flow, fused (planar.py:144-166):
//   Qtz = Householder^-1(z); pre = R1 Qtz + b; act = tanh(pre); out = z + Householder(R2 act)
//   logdet = sum_j log(1 + (1 - act_j^2) * diag(R1)_j * diag(R2)_j)
// kPS: per-sample parameters (row-major upper-triangular R from HBM, VALU cross-lane sums); the shared-parameter
// kernel keeps the transposed-R column reads and the ds_bpermute sums it is fastest with
template <int E, bool kPS>
__global__ __launch_bounds__(256) void sylvester_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ lad, const float* __restrict__ q,
                                                        const float* __restrict__ r1t, const float* __restrict__ r2t,
                                                        const float* __restrict__ bias, const float* __restrict__ rdiag,
                                                        int64_t n, int d, int m) {
  constexpr bool per_sample = kPS;
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n; row += stride) {
    Row<E> z, t, a;
    load_row<E>(z, x + row * d, d, lane);
    const float* qr = per_sample ? q + row * (int64_t)m * d : q;
    const float* r1 = per_sample ? r1t + row * (int64_t)d * d : r1t;
    const float* r2 = per_sample ? r2t + row * (int64_t)d * d : r2t;
    const float* bb = per_sample ? bias + row * (int64_t)d : bias;
    const float* rd = per_sample ? rdiag + row * (int64_t)d : rdiag;
    t = z;
    if constexpr (kPS) householder_rows<E>(t, qr, m, d, lane, true);      // Q^T z
    else householder<E, true>(t, qr, m, d, lane, true);
    if constexpr (kPS) matvec_rows_upper<E>(a, t, r1, d, lane);   // R1 Q^T z (row-major per-sample R, upper part only)
    else matvec<E>(a, t, r1, d, lane, 1);
    float ld = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = lane + 64 * e;
      if (i < d) {
        const float act = tanhf(a.v[e] + bb[i]);
        a.v[e] = act;
        ld += logf(1.f + (1.f - act * act) * rd[i]);
      } else {
        a.v[e] = 0.f;
      }
    }
    ld = per_sample ? wave_sum(ld) : wave_sum_lds(ld);
    if constexpr (kPS) matvec_rows_upper<E>(t, a, r2, d, lane);   // R2 act
    else matvec<E>(t, a, r2, d, lane, 1);
    if constexpr (kPS) householder_rows<E>(t, qr, m, d, lane, false);     // Q R2 act
    else householder<E, true>(t, qr, m, d, lane, false);
#pragma unroll
    for (int e = 0; e < E; ++e) z.v[e] = z.v[e] + t.v[e];
    store_row<E>(z, y + row * d, d, lane);
    if (lane == 0 && lad) lad[row] = ld;
  }
}

// Per-sample [d, d] matrices, row-major as emitted by a hyper-network.  Lane j holds x_j; output i is
// a coalesced read of matrix row i, one multiply per lane and a 64-lane butterfly.  Modes: see header.
template <int E>
__global__ __launch_bounds__(256) void linear_per_sample_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                float* __restrict__ lad, const float* __restrict__ m,
                                                                int64_t n, int d, int mode, float sp, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n; row += stride) {
    const float* mr = m + row * (int64_t)d * d;
    Row<E> v, out;
    load_row<E>(v, x + row * d, d, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) out.v[e] = 0.f;
    float ld = 0.f;
    // Rows are taken kRows at a time: their loads are issued together (row by row the loop waits a full memory
    // latency per row) and every matrix element is read exactly once in all four modes.
    constexpr int kRows = 8;
    auto load_rows = [&](int i0, Row<E> (&mi)[kRows]) {
#pragma unroll
      for (int k = 0; k < kRows; ++k) load_row<E>(mi[k], mr + (int64_t)(i0 + k < d ? i0 + k : d - 1) * d, d, lane);
    };
    // softplus(diag M) + eps of the LU forms: lane j loads M[j][j] once per sample (one strided load per register)
    Row<E> diag;
    if (mode >= 2) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = lane + 64 * e;
        diag.v[e] = j < d ? softplus1(mr[(int64_t)j * d + j]) + eps : 1.f;
        ld += j < d ? logf(diag.v[e]) : 0.f;
      }
      ld = wave_sum(ld);
      if (mode == 3) ld = -ld;
    }
    if (mode == 0) {  // y_i = sum_j M_ij x_j: eight rows' sums in one batched reduction (wave64_sum8)
      for (int i0 = 0; i0 < d; i0 += kRows) {
        Row<E> mi[kRows];
        load_rows(i0, mi);
        float part[kRows], tot[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
          part[k] = 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) part[k] += mi[k].v[e] * v.v[e];
        }
        wave64_sum8(part, tot);
#pragma unroll
        for (int k = 0; k < kRows; ++k)
#pragma unroll
          for (int e = 0; e < E; ++e) if (lane + 64 * e == i0 + k) out.v[e] = tot[k];
      }
    } else if (mode == 1) {  // y_j = sum_i M_ij x_i: row i scaled by the broadcast x_i
      for (int i0 = 0; i0 < d; i0 += kRows) {
        Row<E> mi[kRows];
        load_rows(i0, mi);
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
          const float xi = i0 + k < d ? bcast<E>(v, i0 + k) : 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) out.v[e] += mi[k].v[e] * xi;
        }
      }
    } else if (mode == 2) {  // t = U x, y = L t in ONE pass over the rows: y_i = sum_{j<i} L_ij t_j + t_i only needs
                             // the t_j of earlier rows
      Row<E> t;
#pragma unroll
      for (int e = 0; e < E; ++e) t.v[e] = 0.f;
      for (int i0 = 0; i0 < d; i0 += kRows) {
        Row<E> mi[kRows];
        load_rows(i0, mi);
        // (both sets of eight sums are independent once the chunk's t_i are in place: two batched reductions per chunk
        //  instead of sixteen sequential ones)
        float part[kRows], tot[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {       // t_i = dg_i x_i + sp sum_{j>i} M_ij x_j
          const int i = i0 + k;
          part[k] = 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int j = lane + 64 * e;
            part[k] += (j > i && j < d) ? sp * mi[k].v[e] * v.v[e] : (j == i ? diag.v[e] * v.v[e] : 0.f);
          }
        }
        wave64_sum8(part, tot);
#pragma unroll
        for (int k = 0; k < kRows; ++k)
#pragma unroll
          for (int e = 0; e < E; ++e) if (lane + 64 * e == i0 + k) t.v[e] = tot[k];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {       // y_i = t_i + sp sum_{j<i} M_ij t_j
          const int i = i0 + k;
          part[k] = 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int j = lane + 64 * e;
            part[k] += j < i ? sp * mi[k].v[e] * t.v[e] : (j == i ? t.v[e] : 0.f);
          }
        }
        wave64_sum8(part, tot);
#pragma unroll
        for (int k = 0; k < kRows; ++k)
#pragma unroll
          for (int e = 0; e < E; ++e) if (lane + 64 * e == i0 + k) out.v[e] = tot[k];
      }
    } else {  // forward substitution with unit-lower L, then back substitution with U (no pivoting)
      // (measured and dropped: per chunk of eight rows one batched reduction over the columns outside the chunk + the 8 x 8
      //  triangle worked off with v_readlane broadcasts: 3.0 -> 4.4 ms at D = 128 -- the 72 broadcasts per chunk cost more
      //  than the seven dependent reductions they replace)
      out = v;
      for (int i0 = 0; i0 < d; i0 += kRows) {
        Row<E> mi[kRows];
        load_rows(i0, mi);
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
          const int i = i0 + k;
          float part = 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) if (lane + 64 * e < i) part += sp * mi[k].v[e] * out.v[e];
          part = wave_sum(part);
#pragma unroll
          for (int e = 0; e < E; ++e) if (lane + 64 * e == i) out.v[e] -= part;
        }
      }
      for (int i0 = ((d - 1) / kRows) * kRows; i0 >= 0; i0 -= kRows) {
        Row<E> mi[kRows];
        load_rows(i0, mi);
#pragma unroll
        for (int k = kRows - 1; k >= 0; --k) {
          const int i = i0 + k;
          if (i >= d) continue;
          float part = 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int j = lane + 64 * e;
            if (j > i && j < d) part += sp * mi[k].v[e] * out.v[e];
          }
          part = wave_sum(part);
#pragma unroll
          for (int e = 0; e < E; ++e) if (lane + 64 * e == i) out.v[e] = (out.v[e] - part) / diag.v[e];
        }
      }
    }
    store_row<E>(out, y + row * d, d, lane);
    if (lad && lane == 0) lad[row] = ld;
  }
}

// The same four modes for matrices of <= 16 x 16 (low-dimensional conditional flows): four samples per wave, one per
// 16-lane DPP row; lane j of a row holds x_j and column j of every matrix row, a dot product is four DPP rotations.
__global__ __launch_bounds__(256) void linear_per_sample_narrow_kernel(const float* __restrict__ x,
                                                                       float* __restrict__ y, float* __restrict__ lad,
                                                                       const float* __restrict__ m, int64_t n, int d,
                                                                       int mode, float sp, float eps) {
  const int lane = threadIdx.x & 63, j = lane & 15, rowbase = lane & 48;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * 4;
  for (int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * 4; base < n; base += stride) {
    const int64_t row = base + (lane >> 4);
    const int64_t rr = row < n ? row : n - 1;
    const bool live = row < n && j < d;
    const float* mr = m + rr * (int64_t)d * d;
    float mi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mi[i] = (i < d && j < d) ? mr[i * d + j] : 0.f;
    const float xj = j < d ? x[rr * d + j] : 0.f;
    float out = 0.f, ld = 0.f, dg = 1.f;
    if (mode >= 2) {
      dg = j < d ? softplus1(mr[j * d + j]) + eps : 1.f;
      ld = row16_allsum(j < d ? logf(dg) : 0.f);
      if (mode == 3) ld = -ld;
    }
    auto pick = [&](float v, int i) { return __shfl(v, rowbase | i, 64); };   // lane i of this sample's row
    if (mode == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float sres = row16_allsum(mi[i] * xj);
        if (j == i) out = sres;
      }
    } else if (mode == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) out += mi[i] * pick(xj, i);
    } else if (mode == 2) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float part = row16_allsum(j > i ? sp * mi[i] * xj : (j == i ? dg * xj : 0.f));
        if (j == i) t = part;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float part = row16_allsum(j < i ? sp * mi[i] * t : (j == i ? t : 0.f));
        if (j == i) out = part;
      }
    } else {
      out = xj;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float part = row16_allsum(j < i ? sp * mi[i] * out : 0.f);
        if (j == i) out -= part;
      }
#pragma unroll
      for (int i = 15; i >= 0; --i) {
        const float part = row16_allsum(j > i ? sp * mi[i] * out : 0.f);
        if (j == i) out = (out - part) / dg;
      }
    }
    if (live) y[row * d + j] = out;
    if (lad && row < n && j == 0) lad[row] = ld;
  }
}

inline unsigned row_grid(int64_t n) {
  int64_t g = (n + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t cap = 256 * 8;
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

inline int elems_for(int d) { return d <= 64 ? 1 : d <= 128 ? 2 : d <= 256 ? 4 : 8; }

}  // namespace fc

#define FC_ROW_DISPATCH(D, CALL)                  \
  switch (fc::elems_for(D)) {                     \
    case 1: { constexpr int E = 1; CALL; break; } \
    case 2: { constexpr int E = 2; CALL; break; } \
    case 4: { constexpr int E = 4; CALL; break; } \
    default: { constexpr int E = 8; CALL; break; } \
  }

extern "C" int fc_householder(const float* x, float* y, const float* q, int64_t n, int32_t d,
                              int32_t num_transforms, int32_t per_sample, int32_t reverse, void* stream) {
  if (n < 0 || d <= 0 || d > 512 || num_transforms < 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || (!q && num_transforms > 0)) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d <= 16) {
    hipLaunchKernelGGL(fc::householder_narrow_kernel, dim3(fc::row_grid((n + 3) / 4)), dim3(256), 0, s, x, y, q, n, d,
                       num_transforms, per_sample, reverse);
    return hipGetLastError();
  }
  FC_ROW_DISPATCH(d, hipLaunchKernelGGL(fc::householder_kernel<E>, dim3(fc::row_grid(n)), dim3(256), 0, s, x, y,
                                        q, n, d, num_transforms, per_sample, reverse));
  return hipGetLastError();
}

extern "C" int fc_planar(const float* x, float* y, float* logabsdet, const float* w, const float* u_hat,
                         const float* b, int64_t n, int32_t d, int32_t per_sample, void* stream) {
  if (n < 0 || d <= 0 || d > 512) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !w || !u_hat || !b) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d % 4 == 0 && d <= 256 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)u_hat) & 15u) == 0) {
    int lanes = 1;
    while (lanes < d / 4) lanes <<= 1;
    const int rows = 64 / lanes;
    const unsigned grid = fc::row_grid((n + rows - 1) / rows);
#define FC_PLANAR4(LV)                                                                                          \
  case LV:                                                                                                      \
    hipLaunchKernelGGL(fc::planar_rows4_kernel<LV>, dim3(grid), dim3(256), 0, s, x, y, logabsdet, w, u_hat, b, n, d, \
                       per_sample);                                                                             \
    break;
    switch (lanes) {
      FC_PLANAR4(1) FC_PLANAR4(2) FC_PLANAR4(4) FC_PLANAR4(8) FC_PLANAR4(16) FC_PLANAR4(32) FC_PLANAR4(64)
    }
#undef FC_PLANAR4
    return hipGetLastError();
  }
  FC_ROW_DISPATCH(d, hipLaunchKernelGGL(fc::planar_kernel<E>, dim3(fc::row_grid(n)), dim3(256), 0, s, x, y,
                                        logabsdet, w, u_hat, b, n, d, per_sample));
  return hipGetLastError();
}

extern "C" int fc_linear(const float* x, float* y, const float* a_t, const float* b_t, const float* bias,
                         int64_t n, int32_t d, int32_t mode, void* stream) {
  if (n < 0 || d <= 0 || d > 512 || mode < 0 || mode > 2) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !a_t || (mode != 0 && !b_t)) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  FC_ROW_DISPATCH(d, hipLaunchKernelGGL(fc::linear_kernel<E>, dim3(fc::row_grid(n)), dim3(256), 0, s, x, y, a_t,
                                        b_t, bias, n, d, mode));
  return hipGetLastError();
}

extern "C" int fc_linear_per_sample(const float* x, float* y, float* logabsdet, const float* m, int64_t n,
                                    int32_t d, int32_t mode, float offdiag_scale, float eps, void* stream) {
  if (n < 0 || d <= 0 || d > 512 || mode < 0 || mode > 3) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !m) return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d <= 16) {
    hipLaunchKernelGGL(fc::linear_per_sample_narrow_kernel, dim3(fc::row_grid((n + 3) / 4)), dim3(256), 0, s, x, y,
                       logabsdet, m, n, d, mode, offdiag_scale, eps);
    return hipGetLastError();
  }
  FC_ROW_DISPATCH(d, hipLaunchKernelGGL(fc::linear_per_sample_kernel<E>, dim3(fc::row_grid(n)), dim3(256), 0, s, x,
                                        y, logabsdet, m, n, d, mode, offdiag_scale, eps));
  return hipGetLastError();
}

extern "C" int fc_sylvester(const float* x, float* y, float* logabsdet, const float* q, const float* r1_t,
                            const float* r2_t, const float* bias, const float* r_diag_prod, int64_t n,
                            int32_t d, int32_t num_householder, int32_t per_sample, void* stream) {
  if (n < 0 || d <= 0 || d > 512 || num_householder < 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !y || !r1_t || !r2_t || !bias || !r_diag_prod || (!q && num_householder > 0))
    return hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (per_sample) {
    FC_ROW_DISPATCH(d, hipLaunchKernelGGL((fc::sylvester_kernel<E, true>), dim3(fc::row_grid(n)), dim3(256), 0, s, x,
                                          y, logabsdet, q, r1_t, r2_t, bias, r_diag_prod, n, d, num_householder));
  } else {
    FC_ROW_DISPATCH(d, hipLaunchKernelGGL((fc::sylvester_kernel<E, false>), dim3(fc::row_grid(n)), dim3(256), 0, s, x,
                                          y, logabsdet, q, r1_t, r2_t, bias, r_diag_prod, n, d, num_householder));
  }
  return hipGetLastError();
}
