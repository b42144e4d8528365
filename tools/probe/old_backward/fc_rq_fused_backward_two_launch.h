// Backward of the fused final-Linear + RQ-spline coupling layer (training through the HIP path), gfx950.
//
//   forward:   params = W h + b;   y, logabsdet = rq_spline(x, params)            (fc_rq_fused_general.h)
//   backward:  given gy = dL/dy [N, D], gl = dL/dlogabsdet [N]
//       G[n, f]  = dL/dparams                      (closed-form spline backward, fc_rq_backward.hip)
//       gx       = gy on the identity columns, gy dy/dx + gl dlad/dx on the transformed ones
//       gh[n, :] = W^T G[n, :]                     (gradient into the conditioner's hidden stack)
//       gW       = sum_n G[n, :] (x) h[n, :],   gb = sum_n G[n, :]
//
// What torch.autograd does for the reference (examples/toy_2d.py:57-68) materialises params [N, d_t P] in the forward
// and G [N, d_t P] in the backward and runs three library GEMMs over them: ~9 GB of HBM traffic per layer at N = 2^19.
// Here neither tensor ever exists: the parameters are RECOMPUTED on the matrix cores from the saved h (the forward
// kernel's product, fragment for fragment), the spline backward runs on them in registers, and G goes from the lane's
// registers straight into the two products that consume it.  Two launches share this code (`kRole`), each recomputing
// G, because the second product's accumulators (the wave's [4 dims x P, 64] slice of gW) fill the register file:
//
//   kRole 0 ("dx"): gx, gh, gb.   gh^T = W^T G: the lane's own 3K-/+1 gradients ARE its B operand (accumulator order =
//           k order, the trick of fc_resnet_hidden.hip); per (sample, wave) power-of-two scale; the 8 waves' partial
//           gh tiles are summed in a fixed order through LDS (deterministic).
//   kRole 1 ("dw"): gW.   gW = G^T h contracts over SAMPLES, which live on lanes: G passes through a wave-private LDS
//           strip, one 16-feature tile at a time, to become an A operand (features on rows, samples on k); h^T comes
//           from a transposed copy of the h tile.  h keeps its per-row scale 2^T_s (shared with the recompute product),
//           so G is pre-multiplied by 2^-T_s (exact); one running power-of-two scale per wave for G: when a tile
//           needs a smaller one the accumulators are rescaled (exact), so the sum over all tiles stays in one scale.
//
// hidden == 64 (the conditioner width fc_resnet_hidden serves), K = 4..16 where the accumulators fit (T <= 8).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_device.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "fc_rq_op.h"
#include "fc_rq_backward_op.h"
#include "fc_split.h"
#include "fc_tile.h"
#include "fc_rq_fused_general.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct BwdArgs {
  const float* x;          // [N, D]  layer input (saved by the forward)
  const float* h;          // [N, 64] conditioner's last hidden activation (saved by the forward)
  const float* gy;         // [N, D]
  const float* gl;         // [N] or null (zeros)
  const f16x8* wfrag;      // forward fragments  [groups][2][T][2][64]        (ops.pack_final_layer_general, hidden 64)
  const float* wun;        // [groups]
  const float* bias;       // [groups][4][PP]
  const f16x8* wtfrag;     // role 0: W^T fragments [groups][4 hidden tiles][KK][2][64]   (ops.pack_final_layer_transposed)
  const int32_t* cols;     // [dt]
  float* gx;               // role 0: [N, D]
  float* gh;               // role 0: [N, 64]
  float* gb;               // role 0: [groups][4][PP], accumulated with atomics (zeroed by the caller)
  float* gw;               // role 1: [groups][4][PP][64], accumulated with atomics (zeroed by the caller)
  int64_t tiles;           // 32-row tiles
  int D, dt;
};

// a fragment pointer that went through an opaque asm (to keep it a scalar base) must say it is global memory again: the
// compiler otherwise emits flat loads, which also count on lgkmcnt and are waited for before every LDS read
typedef const __attribute__((address_space(1))) f16x8* GlobalFrags;

// LDS strides (tools/lds_conflicts.py; ds_read_b128 is served in four non-contiguous 16-lane groups: conflict-free rows need a
// stride of 32 mod 64 bytes): h rows 160 B; the transposed h image [hidden][32 samples] 96 B (TSh: its 48 fragment reads per tile
// conflict-free, its 8 transposing 16-bit stores 32-way; at 80 B the reads are 2-way, which costs more in all); the wave-private
// G^T strips 80 B (TS: their 96 16-bit stores per tile conflict-free, their 12 fragment reads 2-way; at 96 B the stores are 2-way).
// Tried and dropped: 64-byte rows with the 16-byte chunks XOR-swizzled by the row (conflict-free reads, 8-way stores) -- the
// address arithmetic costs the registers role 1 does not have (K = 10: +52 B of spills, 2.64 -> 2.85 ms).
constexpr int kBwdH = 64, kBwdR = kGenRows, kBwdTS = 32 + 8, kBwdTSh = 32 + 16;
// f16 per h row: 80 (160 B, conflict-free); role 0 at D > 124 keeps the 144 B of rounds 1-2 (2-way on the fragment reads):
// its 68 KB of partial gh tiles leave no room for the wider rows next to a 128-column x / gy tile pair
__host__ __device__ inline int bwd_hb(int d, int role) { return (role == 0 && d > 124) ? kBwdH + 8 : kBwdH + 16; }

__host__ __device__ inline size_t bwd_lds_bytes(int d, int role) {
  size_t b = (size_t)2 * 2 * kBwdR * bwd_hb(d, role) * 2;   // hbuf [buf][piece][row][80 or 72]
  b += (size_t)2 * 2 * kBwdR * (d + 4) * 4;                 // xbuf + gbuf, [buf][row][D + 4]
  b += 2 * kBwdR * 4 * 2;                                   // hscale, gl  [buf][row]
  b += 32 * 4 + (size_t)32 * 52 * 4;                        // cols, bias image
  if (role == 0) b += (size_t)8 * kBwdR * (kBwdH + 4) * 4;  // partial gh tiles of the 8 waves
  if (role == 2) b += (size_t)2 * kBwdR * kBwdH * 4;        // merged: gh tiles summed by LDS atomics, ring of two
  if (role != 0) b += (size_t)2 * 2 * kBwdH * kBwdTSh * 2 + (size_t)8 * 2 * 16 * kBwdTS * 2;   // h^T [buf][piece][64][40], G^T strips
  return b;
}


template <int K, bool kTails, int kRole>
__global__ __launch_bounds__(kGenThreads) void rq_fused_backward_kernel(RQParams q, float inv_div, BwdArgs a) {
  using S = GenShape<K, kTails>;
  constexpr int P = S::P, PP = S::PP, T = S::T;
  constexpr int PP8 = (PP + 7) / 8 * 8, KK = PP8 / 8;       // role 0: k-steps of the W^T product (8 parameters per lane)
  constexpr int R = kBwdR, H = kBwdH, KS = 2, TS = kBwdTS, TSh = kBwdTSh;
  constexpr bool kDx = kRole != 1, kDw = kRole != 0, kMerged = kRole == 2;    // role 2: both products from one G
  constexpr bool kBlockwise = kRole == 1 && T > 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  const int D = a.D;
  const int HB = bwd_hb(D, kRole);
  const bool pad_x = (D & 3) == 0;
  const int XS = pad_x ? D + 4 : D;
  _Float16* hbuf = reinterpret_cast<_Float16*>(bsm);                             // [2][2][R][HB]
  float* xbuf = reinterpret_cast<float*>(bsm + (size_t)2 * 2 * R * HB * 2);       // [2][R][D + 4]
  float* gbuf = xbuf + 2 * R * (D + 4);                                           // [2][R][D + 4]  gy in, gx out
  float* hscale = gbuf + 2 * R * (D + 4);                                         // [2][R]
  float* glb = hscale + 2 * R;                                                    // [2][R]
  int* cs = reinterpret_cast<int*>(glb + 2 * R);                                  // [32]
  float* bias_lds = reinterpret_cast<float*>(cs + 32);                            // [8][4][PP] (<= 32 * 52)
  float* part = bias_lds + 32 * 52;                                               // role 0: [8][R][H + 4]; role 2: [2][R][H]
  _Float16* htbuf = reinterpret_cast<_Float16*>(bias_lds + 32 * 52 + (kMerged ? 2 * R * H : 0));   // roles 1, 2: [2][2][H][TS]
  _Float16* strips = htbuf + (size_t)2 * 2 * H * TSh;                             // roles 1, 2: [8 waves][2][16][TS]

  // (the wave index as a scalar: fragment, bias and strip addresses then are SGPR bases + one lane offset instead of two dozen
  //  64-bit VGPR pointers -- those were spilled, and every fragment load of the recompute waited for its address reload)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int s16 = lane & 15, g = lane >> 4;
  const int64_t stride = gridDim.x, tile0 = blockIdx.x;
  if (tile0 >= a.tiles) return;
#ifdef FC_BWD_POISON   // tools/probe only (-DFC_BWD_POISON=0x7fc07fc0u): every LDS word starts as that pattern, so a read of a word nobody wrote shows
  {
    const int words = (int)(bwd_lds_bytes(D, kRole) / 4);
    for (int i = threadIdx.x; i < words; i += kGenThreads) reinterpret_cast<uint32_t*>(bsm)[i] = FC_BWD_POISON;
    __syncthreads();
  }
#endif
  if (tid < 32) cs[tid] = tid < a.dt ? a.cols[tid] : 0;
  const int WD = (a.dt + 3) >> 2;
  const bool active = wave < WD;
  const bool dim_ok = 4 * wave + g < a.dt;
  const int grp = active ? wave : 0;
  for (int i = tid; i < WD * 4 * PP; i += kGenThreads) bias_lds[i] = a.bias[i];
  const f32x4* bw = reinterpret_cast<const f32x4*>(bias_lds + (grp * 4 + g) * PP);
  const float w_un = a.wun[grp];
  const f16x8* const wgrp = a.wfrag + (size_t)grp * KS * T * 2 * 64;      // wave-uniform

  const int xvec = R * D / 4;
  float4 hv, xv0, gv0;
  float glv = 0.f;
  hv = xv0 = gv0 = float4{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int64_t t) __attribute__((always_inline)) {
    hv = reinterpret_cast<const float4*>(a.h + t * R * H)[tid];
    const float4* xg = reinterpret_cast<const float4*>(a.x + t * R * D);
    const float4* gg = reinterpret_cast<const float4*>(a.gy + t * R * D);
    // (one 16-byte piece of x and gy per thread rides in registers across the tile; a layer wider than 64 features has a second
    //  piece per thread, which park() reads when it needs it: eight registers fewer for every other layer)
    const int i0 = tid < xvec ? tid : 0;
    xv0 = xg[i0];
    gv0 = gg[i0];
    if (tid < R) glv = a.gl ? a.gl[t * R + tid] : 0.f;
  };
  auto slot = [&](float* base, int buf, int i) __attribute__((always_inline)) {
    if (!pad_x) return reinterpret_cast<float4*>(base + buf * R * (D + 4) + 4 * i);
    const int e = i * 4, r = e / D, c = e - r * D;
    return reinterpret_cast<float4*>(base + buf * R * (D + 4) + r * XS + c);
  };
  auto park = [&](int buf, int64_t t) __attribute__((always_inline)) {
    {   // thread tid holds h[row tid / 16][4 (tid % 16) ..]: the 16 threads of a row are one DPP row
      const int r = tid >> 4, c = (tid & 15) * 4;
      const float v[4] = {hv.x, hv.y, hv.z, hv.w};
      const float m = row16_allmax(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      float sc, un;
      pow2_scale(m, sc, un);
      f16x4 p0, p1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        _Float16 ph, pl;
        split2(v[j] * sc, ph, pl);
        p0[j] = ph;
        p1[j] = pl;
      }
      _Float16* dst = hbuf + ((size_t)(buf * 2) * R + r) * HB + c;
      *reinterpret_cast<f16x4*>(dst) = p0;
      *reinterpret_cast<f16x4*>(dst + (size_t)R * HB) = p1;
      if constexpr (kDw) {   // the same scaled pieces, transposed: [hidden][sample]
        _Float16* dt = htbuf + ((size_t)(buf * 2) * H + c) * TSh + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dt[(size_t)j * TSh] = p0[j];
          dt[((size_t)H + j) * TSh] = p1[j];
        }
      }
      if ((tid & 15) == 0) hscale[buf * R + r] = un;
    }
    if (tid < xvec) {
      *slot(xbuf, buf, tid) = xv0;
      *slot(gbuf, buf, tid) = gv0;
    }
    if (tid + kGenThreads < xvec) {
      *slot(xbuf, buf, tid + kGenThreads) = reinterpret_cast<const float4*>(a.x + t * R * D)[tid + kGenThreads];
      *slot(gbuf, buf, tid + kGenThreads) = reinterpret_cast<const float4*>(a.gy + t * R * D)[tid + kGenThreads];
    }
    if (tid < R) glb[buf * R + tid] = glv;
  };
  auto hfrag = [&](int buf, int blk, int piece, int ks) __attribute__((always_inline)) {
    return *reinterpret_cast<const f16x8*>(hbuf + ((size_t)(buf * 2 + piece) * R + 16 * blk + s16) * HB + 32 * ks + 8 * g);
  };

  // role 0: sums of G over this lane's samples (the bias gradient); role 1: the wave's slice of gW
  float gbacc[kDx ? PP : 1];
  f32x4 dw[kDw ? T : 1][4];
  // per feature, the gW accumulators hold sum G' 2^fshift: fshift + 128 in one byte (255 = nothing yet), four features per
  // register (24 separate registers were a quarter of the role's spills)
  uint32_t fsh[kDw ? (PP + 3) / 4 : 1];
#pragma unroll
  for (int i = 0; i < (kDw ? (PP + 3) / 4 : 1); ++i) fsh[i] = 0xffffffffu;
  if constexpr (kDx) {
#pragma unroll
    for (int i = 0; i < PP; ++i) gbacc[i] = 0.f;
  }
  if constexpr (kDw) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) dw[t][ht] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (kMerged) {
    for (int i = tid; i < 2 * R * H; i += kGenThreads) part[i] = 0.f;
  }

  fetch(tile0);
  park(0, tile0);
  __syncthreads();
  int buf = 0;
  for (int64_t tile = tile0; tile < a.tiles; tile += stride) {
    const bool has_next = tile + stride < a.tiles;
    // vmcnt retires in order: rows requested ahead of the fragment loads make the first product wait out an HBM round trip
    // instead of an L2 one, so the roles that have the registers ask for them after their recompute (role 0; role 1 up to K = 8)
    constexpr bool kFetchLate = kRole == 0 || (kRole == 1 && T <= 6) || kBlockwise;   // (blockwise: between its two blocks)
    if (!kFetchLate && has_next) fetch(tile + stride);
    if (active) {
      float gp[kBlockwise ? 1 : 2][PP8];
      auto spline_block = [&](int b, const f32x4 (&accb)[T], float (&gpb)[PP8]) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);     // the two blocks' register-hungry spline code must not interleave
        const int row = 16 * b + s16;
        const int col = cs[(4 * wave + g) & 31];
        const float xin = xbuf[buf * R * (D + 4) + row * XS + col];
        float* gslot = gbuf + buf * R * (D + 4) + row * XS + col;
        const float gyv = *gslot, glr = glb[buf * R + row];
        const float c = hscale[buf * R + row] * w_un;
        float p[PP];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const f32x4 bt = bw[t];
#pragma unroll
          for (int r = 0; r < 4; ++r) p[4 * t + r] = __builtin_fmaf(accb[t][r], c, bt[r]);
        }
        float gxv, gpe[3 * K + 1];
        rq_backward_element_fast<K, kTails>(q, inv_div, p, xin, gyv, glr, gxv, gpe);
#pragma unroll
        for (int i = 0; i < PP8; ++i) gpb[i] = (i < P && dim_ok) ? gpe[i < P ? i : 0] : 0.f;
        if constexpr (kDx) {
          if (dim_ok) *gslot = gxv;
#pragma unroll
          for (int i = 0; i < PP; ++i) gbacc[i] += gpb[i];
        }
      };
      if constexpr (kBlockwise) {
        // Role 1 at T > 6 (K >= 9): next to the wave's gW accumulators (16 T registers) the two blocks' parameter
        // accumulators and gradients no longer fit (K = 10: 296 B of spills, 3.15 ms per 2^20 rows against role 0's 1.63).
        // Here the blocks take turns: product of block b (its own pass over the fragments: they come from L2), its spline
        // backward, then the next block -- one set of parameter accumulators alive at a time.
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          f32x4 accb[T];
#pragma unroll
          for (int t = 0; t < T; ++t) accb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const f16x8 bh = hfrag(buf, b, 0, ks), bl = hfrag(buf, b, 1, ks);
            const f16x8* wk_ = wgrp + (size_t)ks * T * 2 * 64;
            asm volatile("" : "+s"(wk_));
            const GlobalFrags wk = (GlobalFrags)wk_;
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const f16x8 ah = wk[(t * 2 + 0) * 64 + lane], al = wk[(t * 2 + 1) * 64 + lane];
              accb[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accb[t], 0, 0, 0);
              accb[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accb[t], 0, 0, 0);
              accb[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, accb[t], 0, 0, 0);
            }
          }
          spline_block(b, accb, gp[0]);
          // ---- this block's share of the gW slice: contraction over ITS 16 samples (v_mfma_f32_16x16x16_f16: lane holds
          // row / column l & 15, k = 4 (l >> 4) + j) -- same strip and h^T images as the two-block form below
          const float un_s = hscale[buf * R + 16 * b + s16];
#pragma unroll
          for (int i = 0; i < PP; ++i) {
            gp[0][i] *= un_s;
            const float m = row16_allmax(fabsf(gp[0][i]));
            const uint32_t e = (__float_as_uint(m) >> 23) & 255u;
            const int sh8 = 8 * (i & 3);
            uint32_t cur = (fsh[i >> 2] >> sh8) & 255u;
            const uint32_t want = (e >= 11u && e < 255u) ? 265u - e : cur;
            if (want < cur) {
              if (cur != 255u) {
                const int dlt = (int)want - (int)cur;
                const float resc = dlt < -126 ? 0.f : __uint_as_float((uint32_t)(127 + dlt) << 23);
#pragma unroll
                for (int ht = 0; ht < 4; ++ht) dw[i >> 2][ht][i & 3] *= resc;
              }
              fsh[i >> 2] = (fsh[i >> 2] & ~(255u << sh8)) | (want << sh8);
              cur = want;
            }
            gp[0][i] *= cur == 255u ? 1.f : __uint_as_float((cur - 1u) << 23);
          }
          _Float16* strip = strips + (size_t)wave * 2 * 16 * TS;
#pragma unroll
          for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              _Float16 ph, pl;
              split2(gp[0][4 * t + r], ph, pl);
              strip[(size_t)(4 * g + r) * TS + 16 * b + s16] = ph;
              strip[(size_t)(16 + 4 * g + r) * TS + 16 * b + s16] = pl;
            }
            const f16x4 ah = *reinterpret_cast<const f16x4*>(strip + (size_t)s16 * TS + 16 * b + 4 * g);
            const f16x4 al = *reinterpret_cast<const f16x4*>(strip + (size_t)(16 + s16) * TS + 16 * b + 4 * g);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
              const _Float16* hb = htbuf + ((size_t)(buf * 2) * H + 16 * ht + s16) * TSh + 16 * b + 4 * g;
              const f16x4 bh = *reinterpret_cast<const f16x4*>(hb);
              const f16x4 bl = *reinterpret_cast<const f16x4*>(hb + (size_t)H * TSh);
              dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, dw[t][ht], 0, 0, 0);
              dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, dw[t][ht], 0, 0, 0);
              dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, dw[t][ht], 0, 0, 0);
            }
          }
          if (b == 0) {      // the next tile's rows: requested half-way, so that they are not carried through block 0
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) fetch(tile + stride);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      f32x4 acc[kBlockwise ? 1 : 2][T];
      if constexpr (!kBlockwise) {
      // ---- recompute the parameters of both blocks against each weight fragment (the forward kernel's product: one
      // pass over the wave's 24 KB of fragments per tile -- they stream from L2, whose bandwidth bounds this kernel when
      // every block fetches them again) -----------------------------------------------------------------------------
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (kRole == 0) {
        // Role 0 has registers to spare (183 of 256): the fragment pairs run through a ring of kRing loads in flight, so the
        // L2 round trip of a fragment overlaps the 6 x kRing MFMAs before it instead of being waited out pair by pair.
        constexpr int NF = KS * T, kRing = NF < 6 ? NF : 6;
        const f16x8* wk_ = wgrp;
        asm volatile("" : "+s"(wk_));
        const GlobalFrags wk = (GlobalFrags)wk_;
        f16x8 rh[kRing], rl[kRing];
#pragma unroll
        for (int i = 0; i < kRing; ++i) {
          rh[i] = wk[(i * 2 + 0) * 64 + lane];
          rl[i] = wk[(i * 2 + 1) * 64 + lane];
        }
        f16x8 bh0, bl0, bh1, bl1;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          const int ks = i / T, t = i - ks * T;
          if (t == 0) {
            bh0 = hfrag(buf, 0, 0, ks); bl0 = hfrag(buf, 0, 1, ks);
            bh1 = hfrag(buf, 1, 0, ks); bl1 = hfrag(buf, 1, 1, ks);
          }
          __builtin_amdgcn_sched_barrier(0);
          const f16x8 ah = rh[i % kRing], al = rl[i % kRing];
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh1, acc[1][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl1, acc[1][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, acc[1][t], 0, 0, 0);
          if (i + kRing < NF) {
            rh[i % kRing] = wk[((i + kRing) * 2 + 0) * 64 + lane];
            rl[i % kRing] = wk[((i + kRing) * 2 + 1) * 64 + lane];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x8 bh0 = hfrag(buf, 0, 0, ks), bl0 = hfrag(buf, 0, 1, ks);
        const f16x8 bh1 = hfrag(buf, 1, 0, ks), bl1 = hfrag(buf, 1, 1, ks);
        // scalar base, re-made per k-step behind an opaque asm: otherwise the two dozen fragment addresses are hoisted out of
        // the tile loop as 64-bit VGPR pairs (and spilled)
        const f16x8* wk_ = wgrp + (size_t)ks * T * 2 * 64;
        asm volatile("" : "+s"(wk_));
        const GlobalFrags wk = (GlobalFrags)wk_;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const f16x8 ah = wk[(t * 2 + 0) * 64 + lane], al = wk[(t * 2 + 1) * 64 + lane];
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh1, acc[1][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl1, acc[1][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, acc[1][t], 0, 0, 0);
        }
      }
      }   // !kBlockwise
      if constexpr (kFetchLate && !kBlockwise) {
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) fetch(tile + stride);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- spline backward of this lane's two elements -> G in registers ------------------------------------------
      if constexpr (!kBlockwise) {
#pragma unroll
        for (int b = 0; b < 2; ++b) spline_block(b, acc[b], gp[b]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kDx) {
        // ---- gh^T partial of this wave: W^T (this wave's rows) x G, the lane's gradients as its own B operand; both
        // blocks against each W^T fragment ---------------------------------------------------------------------------
        f16x8 bh[2][KK], bl[2][KK];
        float cc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float m = 0.f;
#pragma unroll
          for (int i = 0; i < PP; ++i) m = fmaxf(m, fabsf(gp[b][i]));
          m = rows4_allmax(m, lane);
          float sc, un;
          pow2_scale(m, sc, un);
          cc[b] = un * w_un;
#pragma unroll
          for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              _Float16 ph, pl;
              split2(gp[b][8 * kk + j] * sc, ph, pl);
              bh[b][kk][j] = ph;
              bl[b][kk][j] = pl;
            }
        }
        const f16x8* wt_ = a.wtfrag + (size_t)grp * 4 * KK * 2 * 64;
        asm volatile("" : "+s"(wt_));
        const GlobalFrags wt = (GlobalFrags)wt_;
        // (role 0: the W^T fragment pairs through the same kind of ring as the forward fragments above)
        constexpr int NW = 4 * KK, kWRing = kRole == 0 ? (NW < 6 ? NW : 6) : 1;
        f16x8 wrh[kWRing], wrl[kWRing];
        if constexpr (kRole == 0) {
#pragma unroll
          for (int i = 0; i < kWRing; ++i) {
            wrh[i] = wt[((size_t)i * 2 + 0) * 64 + lane];
            wrl[i] = wt[((size_t)i * 2 + 1) * 64 + lane];
          }
        }
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
          f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
            f16x8 ah, al;
            if constexpr (kRole == 0) {
              __builtin_amdgcn_sched_barrier(0);
              ah = wrh[(ht * KK + kk) % kWRing];
              al = wrl[(ht * KK + kk) % kWRing];
            } else {
              ah = wt[((size_t)(ht * KK + kk) * 2 + 0) * 64 + lane];
              al = wt[((size_t)(ht * KK + kk) * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              o[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[b][kk], o[b], 0, 0, 0);
              o[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[b][kk], o[b], 0, 0, 0);
              o[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[b][kk], o[b], 0, 0, 0);
            }
            if constexpr (kRole == 0) {
              if (ht * KK + kk + kWRing < NW) {
                wrh[(ht * KK + kk) % kWRing] = wt[((size_t)(ht * KK + kk + kWRing) * 2 + 0) * 64 + lane];
                wrl[(ht * KK + kk) % kWRing] = wt[((size_t)(ht * KK + kk + kWRing) * 2 + 1) * 64 + lane];
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            // lane (sample s16, hidden 16 ht + 4 g + r)
            if constexpr (kMerged) {     // the 8 waves add into one tile (order of the additions not fixed)
              float* dst = part + ((size_t)buf * R + 16 * b + s16) * H + 16 * ht + 4 * g;
#pragma unroll
              for (int r = 0; r < 4; ++r)
                __hip_atomic_fetch_add(dst + r, o[b][r] * cc[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              *reinterpret_cast<float4*>(part + ((size_t)wave * R + 16 * b + s16) * (H + 4) + 16 * ht + 4 * g) =
                  float4{o[b][0] * cc[b], o[b][1] * cc[b], o[b][2] * cc[b], o[b][3] * cc[b]};
            }
          }
        }
      }
      if constexpr (kDw && !kBlockwise) {
        // ---- gW slice of this wave: (G 2^-T_s)^T x (h 2^T_s), contraction over the tile's 32 samples ---------------
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float un_s = hscale[buf * R + 16 * b + s16];
#pragma unroll
          for (int i = 0; i < PP; ++i) gp[b][i] *= un_s;
        }
        // One power-of-two scale PER FEATURE (a row of the A operand; constant along the contraction over samples):
        // the ideal shift of this tile's 32 values lifts their maximum into [2^10, 2^11).  The accumulators of a
        // feature hold sum G' 2^fshift; when a tile needs a smaller shift they are rescaled (exact), tiny tiles join in.
#pragma unroll
        for (int i = 0; i < PP; ++i) {
          const float m = row16_allmax(fmaxf(fabsf(gp[0][i]), fabsf(gp[1][i])));
          const uint32_t e = (__float_as_uint(m) >> 23) & 255u;
          const int sh8 = 8 * (i & 3);
          uint32_t cur = (fsh[i >> 2] >> sh8) & 255u;                       // fshift + 128, 255 = nothing yet
          const uint32_t want = (e >= 11u && e < 255u) ? 265u - e : cur;     // (137 - e) + 128, in [11, 254]
          if (want < cur) {
            if (cur != 255u) {
              const int dlt = (int)want - (int)cur;
              const float resc = dlt < -126 ? 0.f : __uint_as_float((uint32_t)(127 + dlt) << 23);
#pragma unroll
              for (int ht = 0; ht < 4; ++ht) dw[i >> 2][ht][i & 3] *= resc;
            }
            fsh[i >> 2] = (fsh[i >> 2] & ~(255u << sh8)) | (want << sh8);
            cur = want;
          }
          const float sc = cur == 255u ? 1.f : __uint_as_float((cur - 1u) << 23);          // 2^(cur - 128)
          gp[0][i] *= sc;
          gp[1][i] *= sc;
        }
        _Float16* strip = strips + (size_t)wave * 2 * 16 * TS;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          // G^T tile t -> strip[piece][rho = 4 g + r][sample]: A operand rows are (dim g, param 4 t + r)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              _Float16 ph, pl;
              split2(gp[b][4 * t + r], ph, pl);
              strip[(size_t)(4 * g + r) * TS + 16 * b + s16] = ph;
              strip[(size_t)(16 + 4 * g + r) * TS + 16 * b + s16] = pl;
            }
          const f16x8 ah = *reinterpret_cast<const f16x8*>(strip + (size_t)s16 * TS + 8 * g);
          const f16x8 al = *reinterpret_cast<const f16x8*>(strip + (size_t)(16 + s16) * TS + 8 * g);
#pragma unroll
          for (int ht = 0; ht < 4; ++ht) {
            const _Float16* hb = htbuf + ((size_t)(buf * 2) * H + 16 * ht + s16) * TSh + 8 * g;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(hb);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(hb + (size_t)H * TSh);
            dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, dw[t][ht], 0, 0, 0);
            dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, dw[t][ht], 0, 0, 0);
            dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, dw[t][ht], 0, 0, 0);
          }
        }
      }
    } else if constexpr (kFetchLate) {
      if (has_next) fetch(tile + stride);       // (waves without spline work still carry their share of the next tile)
    }
    if (has_next) park(buf ^ 1, tile + stride);
    __syncthreads();
    if constexpr (kMerged) {
      float4* og = reinterpret_cast<float4*>(a.gx + tile * R * D);
      if (tid < xvec) og[tid] = *slot(gbuf, buf, tid);
      if (tid + kGenThreads < xvec) og[tid + kGenThreads] = *slot(gbuf, buf, tid + kGenThreads);
      // gh tile: complete since every wave passed the barrier; read it, clear it for the tile after next
      float4* pt = reinterpret_cast<float4*>(part + (size_t)buf * R * H) + tid;
      reinterpret_cast<float4*>(a.gh + tile * R * H)[tid] = *pt;
      *pt = float4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (kRole == 0) {
      // gx tile (gy with the transformed columns overwritten) and gh tile (the waves' partials in wave order)
      float4* og = reinterpret_cast<float4*>(a.gx + tile * R * D);
      if (tid < xvec) og[tid] = *slot(gbuf, buf, tid);
      if (tid + kGenThreads < xvec) og[tid + kGenThreads] = *slot(gbuf, buf, tid + kGenThreads);
      const int r = tid >> 4, c = (tid & 15) * 4;
      float4 s = *reinterpret_cast<const float4*>(part + (size_t)r * (H + 4) + c);
      for (int w = 1; w < WD; ++w) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)w * R + r) * (H + 4) + c);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      reinterpret_cast<float4*>(a.gh + tile * R * H)[tid] = s;
      __syncthreads();      // `part` is rewritten by the next tile
    }
    buf ^= 1;
  }
  if (!active) return;
  if constexpr (kDx) {
    // gb: sum over the 16 sample lanes of a row, one atomic per (dim, parameter) and workgroup
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const float v = row16_allsum(gbacc[i]);
      if (s16 == 0 && i < P && dim_ok) atomicAdd(a.gb + (size_t)(grp * 4 + g) * PP + i, v);
    }
  }
  if constexpr (kDw) {
    // lane (hidden 16 ht + s16, feature rho = 4 g + r of tile t) = gW[(dim 4 grp + g), param 4 t + r][hidden]
    if (dim_ok) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * t + r < P && ((fsh[t] >> (8 * r)) & 255u) != 255u) {
            const float un = __uint_as_float((255u - ((fsh[t] >> (8 * r)) & 255u)) << 23);      // 2^-(cur - 128)
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
              atomicAdd(a.gw + ((size_t)(grp * 4 + g) * PP + 4 * t + r) * H + 16 * ht + s16, dw[t][ht][r] * un);
#ifdef FC_BWD_PARTIALS   // tools/probe only (role 1 does not write gx; one tile per workgroup): this workgroup's share
              if (kRole == 1 && g == 3 && t == 0) a.gx[tile0 * R * D + (wave * 4 + r) * H + 16 * ht + s16] = dw[t][ht][r] * un;
#endif
            }
          }
    }
  }
}

hipError_t launch_backward_tails(int K, int role, const RQParams& q, const BwdArgs& a, hipStream_t stream);
hipError_t launch_backward_box(int K, int role, const RQParams& q, const BwdArgs& a, hipStream_t stream);

template <int K, bool kTails, int kRole>
hipError_t launch_backward_role(const RQParams& q, const BwdArgs& a, hipStream_t stream) {
  const size_t lds = bwd_lds_bytes(a.D, kRole);
  if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
  static PerDeviceOnce attr;
  const hipError_t ea = ensure_max_dynamic_lds(
      attr, reinterpret_cast<const void*>(&rq_fused_backward_kernel<K, kTails, kRole>), 160 * 1024);
  if (ea != hipSuccess) return ea;
  const int64_t cus = device_cu_count();
  const unsigned grid = (unsigned)(cus < a.tiles ? cus : a.tiles);
  hipLaunchKernelGGL((rq_fused_backward_kernel<K, kTails, kRole>), dim3(grid), dim3(kGenThreads), lds, stream, q,
                     1.f / q.wh_div, a);
  return hipGetLastError();
}

template <int K, bool kTails>
hipError_t launch_backward(int role, const RQParams& q, const BwdArgs& a, hipStream_t stream) {
  if constexpr (GenShape<K, kTails>::T > 8) return hipErrorInvalidValue;
  else
    return role == 0   ? launch_backward_role<K, kTails, 0>(q, a, stream)
           : role == 1 ? launch_backward_role<K, kTails, 1>(q, a, stream)
                       : launch_backward_role<K, kTails, 2>(q, a, stream);
}

}  // namespace fc
