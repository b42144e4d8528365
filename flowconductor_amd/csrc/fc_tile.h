// Tile machinery shared by every per-sample-parameter bijector kernel (gfx950).
//
// One workgroup owns a tile of S consecutive samples.  The tile's per-sample
// parameter rows ([S, rowlen] f32, contiguous in HBM) and its input rows
// ([S, D] f32, contiguous) are copied to LDS with 16-byte coalesced loads, one
// thread then evaluates one (sample, transformed-dim) element out of LDS, the
// transformed value is written back into the LDS image of the row (so identity
// columns pass through untouched), the per-sample log|det J| is reduced across
// the d_t lanes of the row, and the whole [S, D] tile leaves with 16-byte
// coalesced stores.  HBM sees exactly: params once, x once, y once, logabsdet
// once -- the algorithmic bytes of DESIGN.md.
//
// Replaces the gather / bool-mask / scatter op sequences of
// flowcon/transforms/coupling.py:82-98 and the reshape + sum_except_batch of
// coupling.py:279-293 (reference citations, not copied code).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_device.h"

namespace fc {

constexpr int kMaxBlock = 256;

// error bits OR-ed into the caller's device flag word
constexpr uint32_t kErrOutsideDomain = 1u;   // transforms/base.py:16 InputOutsideDomain
constexpr uint32_t kErrDiscriminant = 2u;    // splines/rational_quadratic.py:142 assert
constexpr uint32_t kErrNonFinite = 4u;

struct TileArgs {
  const float* x;        // [N, D]
  float* y;              // [N, D] (may alias x)
  const float* params;   // [N, rowlen] or [rowlen] when shared
  const int32_t* cols;   // [d_t] transformed column of x for dim j, or nullptr (j -> j)
  float* logabsdet;      // [N] or nullptr
  uint32_t* err;         // device flag word or nullptr
  int64_t N;
  int D;
  int d_t;
  int rowlen;            // parameter floats per sample
  int S;                 // samples per tile
  int shared_params;     // 1: one parameter row for the whole batch (nonlinearities.py:246)
  int lad_mode;          // 0 store, 1 accumulate (+=), 2 store negated, 3 accumulate negated
  int vec_ok;            // 1: every tile start is 16-byte aligned for x, y and params
};

// ---- cooperative contiguous copies -------------------------------------------------

template <bool kVec>
__device__ __forceinline__ void copy_in(float* __restrict__ dst, const float* __restrict__ src,
                                        int count) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (kVec) {
    const int nvec = count >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = tid; i < nvec; i += nthr) d4[i] = s4[i];
    for (int i = (nvec << 2) + tid; i < count; i += nthr) dst[i] = src[i];
  } else {
    for (int i = tid; i < count; i += nthr) dst[i] = src[i];
  }
}

template <bool kVec>
__device__ __forceinline__ void copy_out(float* __restrict__ dst, const float* __restrict__ src,
                                         int count) {
  copy_in<kVec>(dst, src, count);
}

// sum over groups of W consecutive lanes (W power of two, <= 64)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, W);
  return v;
}

__device__ __forceinline__ float group_sum_rt(float v, int W) {
  for (int o = W >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void emit_lad(float* p, float v, int mode) {
  if (mode & 2) v = -v;
  if (mode & 1) *p += v; else *p = v;
}

// ---- the tile kernel ------------------------------------------------------------------
//
// Op interface:
//   static constexpr bool kHasPrepare;
//   __device__ void prepare(float* prow, int j, int d_t) const;      (only if kHasPrepare)
//   __device__ void eval(const float* prow, int j, int d_t, float x,
//                        float& y, float& lad, uint32_t& err) const;
// prow = this sample's parameter row in LDS; the op knows its own layout.
//
// LDS layout (dynamic): [ params: S*rowlen (or rowlen if shared) | x: S*D | lad: S*d_t ]
// Every region start is rounded up to 4 floats so float4 LDS accesses stay aligned.

__host__ __device__ inline int round4(int v) { return (v + 3) & ~3; }

template <class Op, bool kVec>
__global__ __launch_bounds__(kMaxBlock) void tile_kernel(Op op, TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = a.S, D = a.D, d_t = a.d_t, rowlen = a.rowlen;
  const int64_t n0 = (int64_t)blockIdx.x * S;
  const int s_eff = (int)((a.N - n0) < (int64_t)S ? (a.N - n0) : (int64_t)S);

  float* ps = smem;
  float* xs = ps + round4(a.shared_params ? rowlen : S * rowlen);
  float* ls = xs + round4(S * D);

  if (a.shared_params) {
    copy_in<kVec>(ps, a.params, rowlen);
  } else {
    copy_in<kVec>(ps, a.params + n0 * rowlen, s_eff * rowlen);
  }
  copy_in<kVec>(xs, a.x + n0 * D, s_eff * D);
  __syncthreads();

  if (Op::kHasPrepare) {
    // once per staged (row, dim): turn raw parameters into derived ones in place (LDS)
    const int rows = a.shared_params ? 1 : s_eff;
    for (int e = threadIdx.x; e < rows * d_t; e += blockDim.x) {
      const int s = e / d_t, j = e - s * d_t;
      op.prepare(ps + s * rowlen, j, d_t);
    }
    __syncthreads();
  }

  const int total = s_eff * d_t;
  const bool pow2 = (d_t & (d_t - 1)) == 0 && d_t <= 64;
  uint32_t err = 0;

  // Padded trip count: every lane of every wave runs the same number of iterations so the
  // cross-lane row reduction never sees a retired lane.
  const int padded = ((S * d_t + 63) / 64) * 64;
  for (int e = threadIdx.x; e < padded; e += blockDim.x) {
    int s, j;
    if (pow2) {
      const int sh = __builtin_ctz(d_t);
      s = e >> sh;
      j = e & (d_t - 1);
    } else {
      s = e / d_t;
      j = e - s * d_t;
    }
    float lad = 0.f;
    const bool live = e < total;
    if (live) {
      const int col = a.cols ? a.cols[j] : j;
      const float* prow = a.shared_params ? ps : ps + s * rowlen;
      const float xv = xs[s * D + col];
      float yv;
      op.eval(prow, j, d_t, xv, yv, lad, err);
      xs[s * D + col] = yv;
    }
    if (a.logabsdet) {
      if (pow2) {
        const float tot = group_sum_rt(lad, d_t);
        if (live && j == 0) emit_lad(a.logabsdet + n0 + s, tot, a.lad_mode);
      } else if (live) {
        ls[e] = lad;
      }
    }
  }
  __syncthreads();

  if (a.logabsdet && !pow2) {
    // one thread per sample walks its row in dim order (deterministic)
    for (int s = threadIdx.x; s < s_eff; s += blockDim.x) {
      float tot = 0.f;
      const float* r = ls + s * d_t;
      for (int j = 0; j < d_t; ++j) tot += r[j];
      emit_lad(a.logabsdet + n0 + s, tot, a.lad_mode);
    }
  }
  copy_out<kVec>(a.y + n0 * D, xs, s_eff * D);

  if (err && a.err) atomicOr(a.err, err);
}

// ---- persistent, register-prefetching variant ---------------------------------------------------
//
// Same tile algorithm, restructured for memory-level parallelism: a resident workgroup walks tiles
// b, b + G, b + 2G, ... and issues the 16-byte global loads of its NEXT tile into registers right
// after it has parked the current tile in LDS, so the HBM latency of tile t+1 is covered by the
// evaluation and the store of tile t.  (With one tile per workgroup every wave spent ~75 % of its
// life in s_waitcnt for its own loads: rocprofv3 SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/.)
// No global load may sit inside the evaluation phase -- vmcnt is an in-order counter, waiting for
// any later load would also drain the prefetch -- so `cols` is staged in LDS once.
// Handles full tiles only; the host runs the < S leftover samples through tile_kernel.
template <class Op, int NVP, int NVX>
__global__ __launch_bounds__(kMaxBlock) void tile_kernel_pf(Op op, TileArgs a, int64_t num_tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = a.S, D = a.D, d_t = a.d_t, rowlen = a.rowlen;
  float* ps = smem;
  float* xs = ps + round4(S * rowlen);
  float* ls = xs + round4(S * D);
  int* cs = reinterpret_cast<int*>(ls + round4(S * d_t));
  const int tid = threadIdx.x;
  const int pvec = (S * rowlen) >> 2, xvec = (S * D) >> 2;

  for (int j = tid; j < d_t; j += kMaxBlock) cs[j] = a.cols ? a.cols[j] : j;

  float4 pr[NVP], xr[NVX];
  int64_t tile = blockIdx.x;
  if (tile < num_tiles) {
    const float4* pg = reinterpret_cast<const float4*>(a.params + tile * S * rowlen);
    const float4* xg = reinterpret_cast<const float4*>(a.x + tile * S * D);
    // unconditional loads with a clamped index keep pr/xr in registers (a predicated load leaves the
    // array "maybe uninitialised" and hipcc demotes it to scratch)
#pragma unroll
    for (int k = 0; k < NVP; ++k) pr[k] = pg[min(tid + k * kMaxBlock, pvec - 1)];
#pragma unroll
    for (int k = 0; k < NVX; ++k) xr[k] = xg[min(tid + k * kMaxBlock, xvec - 1)];
  } else {
#pragma unroll
    for (int k = 0; k < NVP; ++k) pr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < NVX; ++k) xr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  const bool pow2 = (d_t & (d_t - 1)) == 0 && d_t <= 64;
  const int total = S * d_t;
  const int padded = ((total + 63) / 64) * 64;
  uint32_t err = 0;

  for (; tile < num_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * S;
#pragma unroll
    for (int k = 0; k < NVP; ++k)
      if (tid + k * kMaxBlock < pvec) reinterpret_cast<float4*>(ps)[tid + k * kMaxBlock] = pr[k];
#pragma unroll
    for (int k = 0; k < NVX; ++k)
      if (tid + k * kMaxBlock < xvec) reinterpret_cast<float4*>(xs)[tid + k * kMaxBlock] = xr[k];
    __syncthreads();

    const int64_t next = tile + gridDim.x;
    if (next < num_tiles) {
      const float4* pg = reinterpret_cast<const float4*>(a.params + next * S * rowlen);
      const float4* xg = reinterpret_cast<const float4*>(a.x + next * S * D);
#pragma unroll
      for (int k = 0; k < NVP; ++k) pr[k] = pg[min(tid + k * kMaxBlock, pvec - 1)];
#pragma unroll
      for (int k = 0; k < NVX; ++k) xr[k] = xg[min(tid + k * kMaxBlock, xvec - 1)];
    }

    if (Op::kHasPrepare) {
      for (int e = tid; e < total; e += kMaxBlock) {
        const int s = e / d_t, j = e - s * d_t;
        op.prepare(ps + s * rowlen, j, d_t);
      }
      __syncthreads();
    }

    for (int e = tid; e < padded; e += kMaxBlock) {
      int s, j;
      if (pow2) {
        const int sh = __builtin_ctz(d_t);
        s = e >> sh;
        j = e & (d_t - 1);
      } else {
        s = e / d_t;
        j = e - s * d_t;
      }
      float lad = 0.f;
      const bool live = e < total;
      if (live) {
        const int col = cs[j];
        const float xv = xs[s * D + col];
        float yv;
        op.eval(ps + s * rowlen, j, d_t, xv, yv, lad, err);
        xs[s * D + col] = yv;
      }
      if (a.logabsdet) {
        if (pow2) {
          const float tot = group_sum_rt(lad, d_t);
          if (live && j == 0) ls[s] = tot;
        } else if (live) {
          ls[e] = lad;
        }
      }
    }
    __syncthreads();

    if (a.logabsdet) {
      // one coalesced store of the tile's S logabsdet values
      for (int s = tid; s < S; s += kMaxBlock) {
        float tot;
        if (pow2) {
          tot = ls[s];
        } else {
          tot = 0.f;
          const float* r = ls + s * d_t;
          for (int j = 0; j < d_t; ++j) tot += r[j];
        }
        emit_lad(a.logabsdet + n0 + s, tot, a.lad_mode);
      }
    }
    float4* yg = reinterpret_cast<float4*>(a.y + n0 * D);
    for (int i = tid; i < xvec; i += kMaxBlock) yg[i] = reinterpret_cast<const float4*>(xs)[i];
    __syncthreads();
  }
  if (err && a.err) atomicOr(a.err, err);
}

// ---- host-side launch ---------------------------------------------------------------

struct TilePlan {
  int S;
  int block;
  size_t lds_bytes;
  int vec_ok;
  int64_t grid;
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Pick samples-per-tile so that a tile is ~one element per thread of a 256-thread block and
// fits comfortably in LDS (several tiles resident per CU: 160 KiB per CU on gfx950).
#ifndef FC_TILE_TARGET_KB
#define FC_TILE_TARGET_KB 24      // (probe builds sweep it)
#endif
inline bool plan_tile(const TileArgs& a, TilePlan* plan) {
  const size_t kLdsSoft = 40 * 1024, kLdsHard = 150 * 1024;
  int S = kMaxBlock / (a.d_t > 0 ? a.d_t : 1);
  if (S < 1) S = 1;
  // multiples of 4 samples keep every tile start 16-byte aligned whatever D / rowlen are
  if (S >= 4) S &= ~3;
  auto bytes = [&](int s) {
    return sizeof(float) * (size_t)(round4(a.shared_params ? a.rowlen : s * a.rowlen) +
                                    round4(s * a.D) + round4(s * a.d_t));
  };
  // cheap per-sample rows (affine, small D): grow the tile towards ~24 KiB so that the per-tile
  // barriers and the load latency are amortised over more bytes (threads then loop over elements)
  if (!a.shared_params) {
    const size_t kTarget = FC_TILE_TARGET_KB * 1024;
    while (S >= 4 && bytes(S + 4) <= kTarget && (int64_t)(S + 4) * a.D <= 2 * 4 * kMaxBlock &&
           (int64_t)(S + 4) * a.rowlen <= 8 * 4 * kMaxBlock && (int64_t)(S + 4) * 64 <= a.N)
      S += 4;
  }
  while (S > 4 && bytes(S) > kLdsSoft) S -= 4;
  while (S > 1 && bytes(S) > kLdsSoft) S -= 1;
  if (bytes(S) > kLdsHard) return false;
  if ((int64_t)S > a.N) S = (int)(a.N > 0 ? a.N : 1);
  plan->S = S;
  int block = ((S * a.d_t + 63) / 64) * 64;
  plan->block = block > kMaxBlock ? kMaxBlock : block;
  plan->lds_bytes = bytes(S);
  const bool strides_ok = ((int64_t)S * a.D) % 4 == 0 &&
                          (a.shared_params || ((int64_t)S * a.rowlen) % 4 == 0);
  plan->vec_ok = strides_ok && aligned16(a.x) && aligned16(a.y) && aligned16(a.params);
  plan->grid = (a.N + S - 1) / S;
  return true;
}

template <class Op, int NVP, int NVX>
inline hipError_t launch_tile_pf(const Op& op, const TileArgs& a, const TilePlan& plan, int64_t full_tiles,
                                 hipStream_t stream) {
  const size_t lds = plan.lds_bytes + sizeof(int) * (size_t)round4(a.d_t);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel_pf<Op, NVP, NVX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int64_t grid = (int64_t)device_cu_count() * per_cu;
  if (grid > full_tiles) grid = full_tiles;
  hipLaunchKernelGGL((tile_kernel_pf<Op, NVP, NVX>), dim3((unsigned)grid), dim3(kMaxBlock), lds, stream, op, a,
                     full_tiles);
  return hipGetLastError();
}

template <class Op>
inline hipError_t launch_tile(const Op& op, TileArgs a, hipStream_t stream) {
  if (a.N <= 0) return hipSuccess;
  TilePlan plan;
  if (!plan_tile(a, &plan)) return hipErrorInvalidConfiguration;
  a.S = plan.S;
  a.vec_ok = plan.vec_ok;

  // Large per-sample-parameter batches: persistent prefetching kernel on the full tiles, then the
  // < S leftover samples through the one-tile-per-workgroup kernel below.
  const int64_t full_tiles = a.N / plan.S;
  const int64_t pvec = ((int64_t)plan.S * a.rowlen) / 4, xvec = ((int64_t)plan.S * a.D) / 4;
  if (plan.vec_ok && !a.shared_params && plan.block == kMaxBlock &&
      full_tiles >= 64 && pvec <= 8 * kMaxBlock && xvec <= 2 * kMaxBlock) {
    TileArgs body = a;
    body.N = full_tiles * plan.S;
    hipError_t e = (pvec <= 6 * kMaxBlock && xvec <= kMaxBlock)
                       ? launch_tile_pf<Op, 6, 1>(op, body, plan, full_tiles, stream)
                       : launch_tile_pf<Op, 8, 2>(op, body, plan, full_tiles, stream);
    if (e != hipSuccess) return e;
    const int64_t done = body.N;
    if (done == a.N) return hipSuccess;
    a.x += done * a.D;
    a.y += done * a.D;
    a.params += done * a.rowlen;
    if (a.logabsdet) a.logabsdet += done;
    a.N -= done;
    if (!plan_tile(a, &plan)) return hipErrorInvalidConfiguration;
    a.S = plan.S;
    a.vec_ok = plan.vec_ok;
  }
  if (plan.grid > 0x7fffffffLL) return hipErrorInvalidConfiguration;
  dim3 grid((unsigned)plan.grid), block((unsigned)plan.block);
  if (plan.lds_bytes > 64 * 1024) {
    hipError_t e;
    if (plan.vec_ok)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel<Op, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
    else
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_kernel<Op, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
    if (e != hipSuccess) return e;
  }
  if (plan.vec_ok)
    hipLaunchKernelGGL((tile_kernel<Op, true>), grid, block, plan.lds_bytes, stream, op, a);
  else
    hipLaunchKernelGGL((tile_kernel<Op, false>), grid, block, plan.lds_bytes, stream, op, a);
  return hipGetLastError();
}

}  // namespace fc
