// Backward of the fused final-Linear + RQ-spline coupling layer, ONE launch, one wave per SIMD (round 4).
//
// The two-launch form (fc_rq_fused_backward.h, roles 0 and 1) evaluates the parameter recompute and the closed-form
// spline backward TWICE per element because, at two waves per SIMD (256 registers per wave), the wave's gW slice
// (16 T accumulator registers) leaves no room for the gh product next to it; its merged role 2 spills.  gfx950's register
// file is one 512-entry file per lane and SIMD: a 256-thread workgroup (one wave per SIMD) owns all of it.  A single
// wave issues a vector instruction every 4 cycles against ~3.6 for two waves (tools/probe/valu_costs.hip), so giving
// up the second wave costs ~10 % of issue rate and buys the registers that let ONE evaluation feed both products.
//
//   A workgroup walks its 32-row tiles once per SWEEP of four dim groups (d_t <= 16: one sweep; 32 dims: two).  In
//   sweep p wave w owns dim group 4 p + w (dims 16 p + 4 w .. + 3): its [4 dims x PP, 64] slice of gW lives in 16 T
//   accumulator registers for the whole sweep; per tile: recompute of the parameters (split-f16 products against the
//   streamed weight fragments), spline backward of the lane's two elements, gh partial (W^T G, the lane's gradients as
//   its own B operand), gW slice (G^T through the wave-private LDS strip against the transposed h image).  The four
//   waves' gh partials meet in LDS in wave order; the second sweep reads the first sweep's gx tile as its upstream
//   gradient (the first sweep left the second's columns untouched) and adds its gh partial onto the first's: gx and
//   gh are deterministic, as in roles 0 / 1.  The second sweep re-reads x, h and gy (1.3 KB per sample).
//
// The G^T strip holds f32 and the READER scales and splits: lane (feature row s16, sample octet g) reads its 8 samples
// of one feature (two 16-byte reads), so the per-feature maximum is lane-local maxima + one 4-lane exchange instead
// of a 16-lane DPP reduction per feature on the writer's side (24 of them per tile: ~430 vector instructions in roles
// 1 / 2), the running power-of-two shift of a feature is ONE register of its reader lane, the bias gradient is the
// reader's sum of its 8 samples (one register per 16-feature tile instead of one per parameter), and the split
// produces the A operand in place (no second LDS trip).  A shift change (rare) is handed to the accumulator lanes
// through 16 LDS words under a wave-uniform branch.
//
// No scratch at any instantiated K (tools/kernel_stats.py): nothing is spilled, so no spilled address is ever reloaded
// around the strip phase (the property every failing variant of the round-2 gW fault shared).
#pragma once
#include "fc_rq_fused_backward.h"

namespace fc {

#ifndef FC_B5_WRES
#define FC_B5_WRES 1     // 1: the group's forward fragments resident in registers for the whole sweep (8 T registers); 0: streamed per tile
#endif
#ifndef FC_B5_MIX_BLOCKS
#define FC_B5_MIX_BLOCKS 0   // probe: 1 lets the scheduler interleave the two blocks' spline backward, 2 also makes each one basic block
#endif
#ifndef FC_B5_PIPE
#define FC_B5_PIPE 0     // 1: block 1's recompute products inside block 0's (branch-free) spline region with a sched_group_barrier
                         // pipeline.  Measured SLOWER (1.338 vs 1.296 ms per 2^19 rows): hipcc 7.2 still clumps the 36 products at the
                         // head of the region, and the branch-free element evaluates both sides of its selects (+600 cycles per block)
#endif
#ifndef FC_B5_WPRE
#define FC_B5_WPRE 0     // streamed forward fragments: the first ring of the NEXT tile is requested at the end of the current one
#endif
#ifndef FC_B5_WT_EARLY
#define FC_B5_WT_EARLY 0 // W^T fragment pairs of the gh product requested AHEAD of the spline backward (the rest behind it)
#endif
#ifdef FC_B5_STAMP        // tools/probe/b5_clock.py: per-phase cycle totals of every wave land in the first rows of gh (garbage output)
#define FC_B5_MARK(i) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); phase_cyc[i] += now_ - phase_t; phase_t = now_; } while (0)
#else
#define FC_B5_MARK(i) do { } while (0)
#endif
#ifndef FC_B5_ABL
#define FC_B5_ABL 0      // tools/probe/build_b5_variants.sh: 1 no recompute products, 2 stand-in spline backward, 4 no gh product, 8 no gW product
#endif
constexpr int kB5Threads = 256, kB5Waves = 4;
constexpr int kB5SS = 40;      // f32 per strip row: 160 B (ds_read_b128 conflict-free iff the stride is 32 mod 64 bytes)

__host__ __device__ inline size_t bwd512_lds_bytes(int d) {
  constexpr int HB = kBwdH + 16;
  size_t b = (size_t)2 * 2 * kBwdR * HB * 2;                  // hbuf [buf][piece][row][80]
  b += (size_t)2 * 2 * kBwdR * (d + 4) * 4;                   // xbuf + gbuf, [buf][row][D + 4]
  b += 2 * kBwdR * 4 * 2;                                     // hscale, gl  [buf][row]
  b += 32 * 4 + (size_t)32 * 32 * 4;                          // cols, bias image (PP <= 32)
  b += (size_t)kB5Waves * kBwdR * (kBwdH + 4) * 4;            // gh partials of the 4 waves
  b += (size_t)kB5Waves * (16 * kB5SS + 16) * 4;              // G^T strips [wave][16 features][40] f32 + 16 exchange words
  return b;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter (release of global
// writes at workgroup scope): here that would wait out the fragment and row requests deliberately left in flight across it.
// Global data of one tile is written and later re-read by the SAME thread (gx / gh of the previous sweep), so no global
// ordering between threads is needed inside the loop.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int K, bool kTails>
__global__ __launch_bounds__(kB5Threads) void rq_fused_backward512_kernel(RQParams q, float inv_div, BwdArgs a) {
  using S = GenShape<K, kTails>;
  constexpr int P = S::P, PP = S::PP, T = S::T;
  constexpr int PP8 = (PP + 7) / 8 * 8, KK = PP8 / 8;
  constexpr int R = kBwdR, H = kBwdH, KS = 2, SS = kB5SS, HB = kBwdH + 16;
  constexpr int NT = kB5Threads;
  constexpr bool kWres = FC_B5_WRES != 0 && T <= 6;
  constexpr bool kPipe = FC_B5_PIPE != 0 && kWres;      // wider parameter rows: the resident fragments no longer fit next to 16 T accumulators
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  const int D = a.D;
  const bool pad_x = (D & 3) == 0;
  const int XS = pad_x ? D + 4 : D;
  _Float16* hbuf = reinterpret_cast<_Float16*>(bsm);                              // [2][2][R][HB]
  float* xbuf = reinterpret_cast<float*>(bsm + (size_t)2 * 2 * R * HB * 2);        // [2][R][D + 4]
  float* gbuf = xbuf + 2 * R * (D + 4);                                            // [2][R][D + 4]  gy in, gx out
  float* hscale = gbuf + 2 * R * (D + 4);                                          // [2][R]
  float* glb = hscale + 2 * R;                                                     // [2][R]
  int* cs = reinterpret_cast<int*>(glb + 2 * R);                                   // [32]
  float* bias_lds = reinterpret_cast<float*>(cs + 32);                             // [8][4][PP]
  float* part = bias_lds + 32 * 32;                                                // [4][R][H + 4]
  float* strips = part + (size_t)kB5Waves * R * (H + 4);                            // [4 waves][16 * SS + 16]

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int s16_ = lane & 15, g_ = lane >> 4;
  // Inside the tile loop every phase takes FRESH copies of the lane coordinates behind an opaque asm: LDS addresses derived from
  // them are then recomputed per phase (a few vector instructions) instead of being hoisted out of the loop as dozens of
  // loop-invariant address registers that live across all phases.
#define FC_B5_FRESH_LANE() int s16 = s16_, g = g_; asm volatile("" : "+v"(s16), "+v"(g))
  const int64_t stride = gridDim.x, tile0 = blockIdx.x;
  if (tile0 >= a.tiles) return;
  if (tid < 32) cs[tid] = tid < a.dt ? a.cols[tid] : 0;
  const int WD = (a.dt + 3) >> 2;
  for (int i = tid; i < WD * 4 * PP; i += NT) bias_lds[i] = a.bias[i];
  float* strip = strips + (size_t)wave * (16 * SS + 16);
  float* exch = strip + 16 * SS;

  const int xvec = R * D / 4;        // float4 pieces of an x / gy tile (<= 1024 at D = 128)
  const float* gsrc = a.gy;          // upstream gradient of this sweep: gy, then the previous sweep's gx
  // (named registers, not arrays: arrays captured by the lambdas below stay in scratch memory)
  float4 hv0, hv1, xv0, xv1, gv0, gv1;
  float glv = 0.f;
  auto fetch = [&](int64_t t) __attribute__((always_inline)) {
    const float4* hg = reinterpret_cast<const float4*>(a.h + t * R * H);
    const float4* xg = reinterpret_cast<const float4*>(a.x + t * R * D);
    const float4* gg = reinterpret_cast<const float4*>(gsrc + t * R * D);
    const int i0 = tid < xvec ? tid : 0, i1 = tid + NT < xvec ? tid + NT : 0;
    hv0 = hg[tid];
    hv1 = hg[tid + NT];
    xv0 = xg[i0];
    xv1 = xg[i1];
    gv0 = gg[i0];
    gv1 = gg[i1];
    if (tid < R) glv = a.gl ? a.gl[t * R + tid] : 0.f;
  };
  // float offsets of this thread's (up to four) 16-byte pieces of an x / gy tile inside a [R][D + 4] LDS buffer: formed once
  // (the padded form needs an integer division by the run-time row length)
  int so0, so1, so2, so3;
  {
    auto slot_off = [&](int i) {
      if (!pad_x) return 4 * i;
      const int e = i * 4, r = e / D;
      return r * XS + (e - r * D);
    };
    so0 = slot_off(tid); so1 = slot_off(tid + NT); so2 = slot_off(tid + 2 * NT); so3 = slot_off(tid + 3 * NT);
  }
  auto park_h = [&](int buf, int j, const float4& hvj) __attribute__((always_inline)) {
    // thread holds h[row tid / 16 + 16 j][4 (tid % 16) ..]: the 16 threads of a row are one DPP row
    const int r = (tid >> 4) + 16 * j, c = (tid & 15) * 4;
    const float m = row16_allmax(fmaxf(fmaxf(fabsf(hvj.x), fabsf(hvj.y)), fmaxf(fabsf(hvj.z), fabsf(hvj.w))));
    float sc, un;
    pow2_scale(m, sc, un);
    uint32_t h01, l01, h23, l23;
    split2_pair(hvj.x, hvj.y, sc, h01, l01);
    split2_pair(hvj.z, hvj.w, sc, h23, l23);
    _Float16* dst = hbuf + ((size_t)(buf * 2) * R + r) * HB + c;
    *reinterpret_cast<u32x2*>(dst) = u32x2{h01, h23};
    *reinterpret_cast<u32x2*>(dst + (size_t)R * HB) = u32x2{l01, l23};
    if ((tid & 15) == 0) hscale[buf * R + r] = un;
  };
  auto park = [&](int buf, int64_t t) __attribute__((always_inline)) {
    park_h(buf, 0, hv0);
    park_h(buf, 1, hv1);
    float* xb = xbuf + buf * R * (D + 4);
    float* gb_ = gbuf + buf * R * (D + 4);
    if (tid < xvec) {
      *reinterpret_cast<float4*>(xb + so0) = xv0;
      *reinterpret_cast<float4*>(gb_ + so0) = gv0;
    }
    if (tid + NT < xvec) {
      *reinterpret_cast<float4*>(xb + so1) = xv1;
      *reinterpret_cast<float4*>(gb_ + so1) = gv1;
    }
    if (tid + 2 * NT < xvec) {      // layers wider than 64 features
      *reinterpret_cast<float4*>(xb + so2) = reinterpret_cast<const float4*>(a.x + t * R * D)[tid + 2 * NT];
      *reinterpret_cast<float4*>(gb_ + so2) = reinterpret_cast<const float4*>(gsrc + t * R * D)[tid + 2 * NT];
    }
    if (tid + 3 * NT < xvec) {
      *reinterpret_cast<float4*>(xb + so3) = reinterpret_cast<const float4*>(a.x + t * R * D)[tid + 3 * NT];
      *reinterpret_cast<float4*>(gb_ + so3) = reinterpret_cast<const float4*>(gsrc + t * R * D)[tid + 3 * NT];
    }
    if (tid < R) glb[buf * R + tid] = glv;
  };
  auto hfrag = [&](int buf, int blk, int piece, int ks, int s16, int g) __attribute__((always_inline)) {
    return *reinterpret_cast<const f16x8*>(hbuf + ((size_t)(buf * 2 + piece) * R + 16 * blk + s16) * HB + 32 * ks + 8 * g);
  };

#ifdef FC_B5_STAMP
  uint64_t phase_cyc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, phase_t = __builtin_amdgcn_s_memtime();
  const uint64_t stamp_c0 = phase_t, stamp_r0 = __builtin_amdgcn_s_memrealtime();
  uint64_t arrive10 = 0;
#endif
  const int sweeps = (WD + kB5Waves - 1) / kB5Waves;
  for (int sweep = 0; sweep < sweeps; ++sweep) {
    const int grp = wave + kB5Waves * sweep;           // wave-uniform
    const bool active = grp < WD;
    const bool dim_ok = 4 * grp + g_ < a.dt;
    const float w_un = a.wun[active ? grp : 0];
    // this wave's [4 dims x PP, 64] slice of gW (accumulator layout: lane holds rows 4 g + r, column s16) and, on the READER
    // lanes of the G^T strip (feature row s16 of tile t), that feature's running power-of-two shift (+ 128; 255 = nothing
    // yet) and the sum of G over the lane's samples (bias gradient)
    f32x4 dw[T][4];
    uint32_t cur[T];
    float gbr[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      cur[t] = 255u;
      gbr[t] = 0.f;
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) dw[t][ht] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (sweep > 0) {
      // the previous sweep's gx / gh tiles (plain stores of this workgroup) are this sweep's inputs
      __threadfence();
      __syncthreads();
      gsrc = a.gx;
    }
    // (kWres) the group's forward fragments (both pieces of both k-steps) stay in registers for the whole sweep: 8 T of the 512
    constexpr int NF = KS * T;
    f16x8 wres[kWres ? NF : 1][2];
    if constexpr (kWres) {
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        wres[i][0] = a.wfrag[((size_t)(active ? grp : 0) * NF + i) * 2 * 64 + lane];
        wres[i][1] = a.wfrag[((size_t)(active ? grp : 0) * NF + i) * 2 * 64 + 64 + lane];
      }
    }
    // streamed forward fragments: a ring of fragment pairs in flight; its first filling is requested a tile ahead (FC_B5_WPRE)
    constexpr int kRing = NF < 8 ? NF : 8;
    f16x8 rh[kWres ? 1 : kRing], rl[kWres ? 1 : kRing];
    auto wring_request = [&]() __attribute__((always_inline)) {
      if constexpr (!kWres && FC_B5_WPRE != 0) {
        const f16x8* wk_ = a.wfrag + (size_t)(active ? grp : 0) * NF * 2 * 64;
        asm volatile("" : "+s"(wk_));
        const GlobalFrags wk = (GlobalFrags)wk_;
#pragma unroll
        for (int i = 0; i < kRing; ++i) {
          rh[i] = wk[(i * 2 + 0) * 64 + lane];
          rl[i] = wk[(i * 2 + 1) * 64 + lane];
        }
      }
    };
    if (!active)      // a wave without a dim group in this sweep: its gh partial is zero
      for (int i = lane; i < R * (H + 4); i += 64) part[(size_t)wave * R * (H + 4) + i] = 0.f;

    fetch(tile0);
    if (active) wring_request();
    park(0, tile0);
    __syncthreads();
    int buf = 0;
    for (int64_t tile = tile0; tile < a.tiles; tile += stride) {
      const bool has_next = tile + stride < a.tiles;
      FC_B5_MARK(0);      // loop overhead
      float4 ghp0 = float4{0.f, 0.f, 0.f, 0.f}, ghp1 = ghp0;      // the previous sweep's gh of this tile, consumed at the write-out
      if (sweep > 0) {
        ghp0 = reinterpret_cast<const float4*>(a.gh + tile * R * H)[tid];
        ghp1 = reinterpret_cast<const float4*>(a.gh + tile * R * H)[tid + NT];
      }
      if (active) {
        // ---- recompute the parameters of both blocks against the weight fragments of this wave's group
        f32x4 acc[2][T];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int t = 0; t < T; ++t) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (kPipe) one block against the resident fragments: 6 T products
        auto recompute_block = [&](int b) __attribute__((always_inline)) {
          FC_B5_FRESH_LANE();
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const f16x8 bh = hfrag(buf, b, 0, ks, s16, g), bl = hfrag(buf, b, 1, ks, s16, g);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wres[kWres ? ks * T + t : 0][1], bh, acc[b][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wres[kWres ? ks * T + t : 0][0], bl, acc[b][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wres[kWres ? ks * T + t : 0][0], bh, acc[b][t], 0, 0, 0);
          }
        };
        if constexpr (kPipe) {
          recompute_block(0);
        } else {
          FC_B5_FRESH_LANE();
          const f16x8* wk_ = a.wfrag + (size_t)grp * NF * 2 * 64;
          asm volatile("" : "+s"(wk_));
          const GlobalFrags wk = (GlobalFrags)wk_;
          if constexpr (!kWres && FC_B5_WPRE == 0) {
#pragma unroll
            for (int i = 0; i < kRing; ++i) {
              rh[i] = wk[(i * 2 + 0) * 64 + lane];
              rl[i] = wk[(i * 2 + 1) * 64 + lane];
            }
          }
          f16x8 bh0, bl0, bh1, bl1;
#pragma unroll
          for (int i = 0; i < ((FC_B5_ABL & 1) ? 1 : NF); ++i) {
            const int ks = i / T, t = i - ks * T;
            if (t == 0) {
              bh0 = hfrag(buf, 0, 0, ks, s16, g); bl0 = hfrag(buf, 0, 1, ks, s16, g);
              bh1 = hfrag(buf, 1, 0, ks, s16, g); bl1 = hfrag(buf, 1, 1, ks, s16, g);
            }
            const f16x8 ah = kWres ? wres[kWres ? i : 0][0] : rh[kWres ? 0 : i % kRing];
            const f16x8 al = kWres ? wres[kWres ? i : 0][1] : rl[kWres ? 0 : i % kRing];
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh0, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh1, acc[1][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl0, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl1, acc[1][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, acc[1][t], 0, 0, 0);
            if constexpr (!kWres) {
              if (i + kRing < NF) {
                rh[i % kRing] = wk[((i + kRing) * 2 + 0) * 64 + lane];
                rl[i % kRing] = wk[((i + kRing) * 2 + 1) * 64 + lane];
              }
            }
          }
        }
        FC_B5_MARK(1);      // recompute
        // the next tile's rows
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) fetch(tile + stride);
        __builtin_amdgcn_sched_barrier(0);
        // the W^T fragments of the gh product: all requested here, so that their L2 round trip passes behind the spline
        // backward (one wave per SIMD: nobody else hides it)
        constexpr int NW = 4 * KK;
        f16x8 wth[NW], wtl[NW];
        if (!(FC_B5_ABL & 4)) {
          const f16x8* wt_ = a.wtfrag + (size_t)grp * 4 * KK * 2 * 64;
          asm volatile("" : "+s"(wt_));
          const GlobalFrags wt = (GlobalFrags)wt_;
#pragma unroll
          for (int i = 0; i < (FC_B5_WT_EARLY < NW ? FC_B5_WT_EARLY : NW); ++i) {
            wth[i] = wt[((size_t)i * 2 + 0) * 64 + lane];
            wtl[i] = wt[((size_t)i * 2 + 1) * 64 + lane];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        FC_B5_MARK(2);      // fetch + W^T requests issued
        // ---- spline backward of this lane's two elements -> G in registers
        float gp[2][PP8];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#if !FC_B5_MIX_BLOCKS
          __builtin_amdgcn_sched_barrier(0);     // the two blocks' register-hungry spline code must not interleave
#endif
          if constexpr (kPipe) {
            if (b == 0) recompute_block(1);      // its 6 T products are spread over this block's vector stream below
          }
          FC_B5_FRESH_LANE();
          const f32x4* bw = reinterpret_cast<const f32x4*>(bias_lds + (grp * 4 + g) * PP);
          const int row = 16 * b + s16;
          const int col = cs[(4 * grp + g) & 31];
          const float xin = xbuf[buf * R * (D + 4) + row * XS + col];
          float* gslot = gbuf + buf * R * (D + 4) + row * XS + col;
          // (a dim beyond d_t: zero upstream gradients make every parameter gradient of the element zero)
          const float gyv = dim_ok ? *gslot : 0.f, glr = dim_ok ? glb[buf * R + row] : 0.f;
          const float c = hscale[buf * R + row] * w_un;
          float p[PP];
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const f32x4 bt = bw[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[4 * t + r] = __builtin_fmaf(acc[b][t][r], c, bt[r]);
          }
          float gxv, gpe[3 * K + 1];
#if FC_B5_ABL & 2
          gxv = xin * gyv + glr;
#pragma unroll
          for (int i = 0; i < 3 * K + 1; ++i) gpe[i] = p[i % PP] * gyv;
#else
          if constexpr (kPipe || FC_B5_MIX_BLOCKS == 2) rq_backward_element_flat<K, kTails>(q, inv_div, p, xin, gyv, glr, gxv, gpe);
          else rq_backward_element_fast<K, kTails>(q, inv_div, p, xin, gyv, glr, gxv, gpe);
#endif
#pragma unroll
          for (int i = 0; i < PP8; ++i) gp[b][i] = i < P ? gpe[i < P ? i : 0] : 0.f;
          if (dim_ok) *gslot = gxv;
          if constexpr (kPipe) {
            if (b == 0) {      // one matrix-core instruction per 14 vector instructions of this region
#pragma unroll
              for (int i = 0; i < 6 * T; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
              }
            }
          }
          if (b == 0) FC_B5_MARK(3); else FC_B5_MARK(4);      // spline backward, block 0 / 1
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- gh^T partial: W^T (this group's rows) x G, the lane's gradients as its own B operand
        if (!(FC_B5_ABL & 4)) {
          FC_B5_FRESH_LANE();
          {
            const f16x8* wt_ = a.wtfrag + (size_t)grp * 4 * KK * 2 * 64;
            asm volatile("" : "+s"(wt_));
            const GlobalFrags wt = (GlobalFrags)wt_;
#pragma unroll
            for (int i = FC_B5_WT_EARLY; i < NW; ++i) {
              wth[i] = wt[((size_t)i * 2 + 0) * 64 + lane];
              wtl[i] = wt[((size_t)i * 2 + 1) * 64 + lane];
            }
          }
          f16x8 bh[2][KK], bl[2][KK];
          float cc[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < PP; i += 2) m = fmaxf(m, fmaxf(fabsf(gp[b][i]), fabsf(gp[b][i + 1])));
            m = rows4_allmax(m, lane);
            float sc, un;
            pow2_scale(m, sc, un);
            cc[b] = un * w_un;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
              u32x4 ph, pl;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                uint32_t h01, l01;
                split2_pair(gp[b][8 * kk + 2 * j], gp[b][8 * kk + 2 * j + 1], sc, h01, l01);
                ph[j] = h01;
                pl[j] = l01;
              }
              bh[b][kk] = __builtin_bit_cast(f16x8, ph);
              bl[b][kk] = __builtin_bit_cast(f16x8, pl);
            }
          }
          // eight independent accumulators (hidden tile x block): consecutive products never wait for each other
          f32x4 o[4][2];
#pragma unroll
          for (int ht = 0; ht < 4; ++ht) o[ht][0] = o[ht][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int ht = 0; ht < 4; ++ht)
#pragma unroll
              for (int b = 0; b < 2; ++b) o[ht][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wtl[ht * KK + kk], bh[b][kk], o[ht][b], 0, 0, 0);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht)
#pragma unroll
              for (int b = 0; b < 2; ++b) o[ht][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wth[ht * KK + kk], bl[b][kk], o[ht][b], 0, 0, 0);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht)
#pragma unroll
              for (int b = 0; b < 2; ++b) o[ht][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wth[ht * KK + kk], bh[b][kk], o[ht][b], 0, 0, 0);
          }
#pragma unroll
          for (int ht = 0; ht < 4; ++ht)
#pragma unroll
            for (int b = 0; b < 2; ++b)      // (sample s16, hidden 16 ht + 4 g + r)
              *reinterpret_cast<float4*>(part + ((size_t)wave * R + 16 * b + s16) * (H + 4) + 16 * ht + 4 * g) =
                  float4{o[ht][b][0] * cc[b], o[ht][b][1] * cc[b], o[ht][b][2] * cc[b], o[ht][b][3] * cc[b]};
        }
        __builtin_amdgcn_sched_barrier(0);
        FC_B5_MARK(5);      // gh product
        // ---- gW slice of this group: (G 2^-T_s)^T x (h 2^T_s), contraction over the tile's 32 samples; bias gradient.
        // Three passes over the T feature tiles so that no LDS round trip is waited for T times by this one wave:
        //   1. all strip writes and read-backs back to back (DS operations of a wave execute in order, so tile t + 1 may
        //      overwrite the strip as soon as tile t's reads are ISSUED);  2. maxima, shifts, one rare branch for all shift
        //      changes;  3. split + products against the h^T fragments, which are the same for every t (read once).
        if (!(FC_B5_ABL & 8)) {
          FC_B5_FRESH_LANE();
          f32x4 va[T], vb[T];
#pragma unroll
          for (int t = 0; t < T; ++t) {
            // G^T tile t -> strip[rho = 4 g + r][sample] (f32): rows are (dim g, param 4 t + r)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) strip[(4 * g + r) * SS + 16 * b + s16] = gp[b][4 * t + r];
            va[t] = *reinterpret_cast<const f32x4*>(strip + s16 * SS + 8 * g);
            vb[t] = *reinterpret_cast<const f32x4*>(strip + s16 * SS + 8 * g + 4);
          }
          // row unscale factors of the reader's 8 samples; h^T fragments (lane: hidden 16 ht + s16, samples 8 g ..)
          const f32x4 un8a = *reinterpret_cast<const f32x4*>(hscale + buf * R + 8 * g);
          const f32x4 un8b = *reinterpret_cast<const f32x4*>(hscale + buf * R + 8 * g + 4);
          // h^T fragments (lane: hidden 16 ht + s16, samples 8 g .. 8 g + 7) straight from the row-major h pieces by gfx950's
          // transposing read: per 16-lane group a block of 4 rows (samples) x 16 columns (hidden) comes back column-major;
          // lane 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3 (no transposed LDS image, no
          // 16-bit scatter stores in park)
          f16x8 hth[4], htl[4];
          {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* LdsTr;
            const int q4 = s16 >> 2, p4 = s16 & 3;
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) {
              const _Float16* hb = hbuf + ((size_t)(buf * 2) * R + 8 * g + q4) * HB + 16 * ht + 4 * p4;
              const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsTr)(hb));
              const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsTr)(hb + 4 * HB));
              const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsTr)(hb + (size_t)R * HB));
              const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsTr)(hb + (size_t)R * HB + 4 * HB));
              typedef short s16x8 __attribute__((ext_vector_type(8)));
              hth[ht] = __builtin_bit_cast(f16x8, s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
              htl[ht] = __builtin_bit_cast(f16x8, s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]});
            }
          }
          // one power-of-two shift per feature, running over the sweep: the ideal shift of this tile's 32 values lifts their
          // maximum into [2^10, 2^11); the accumulators hold sum G' 2^shift; when a tile needs a smaller shift they are
          // rescaled (exact), tiny tiles join in
          uint32_t want[T];
          bool change = false;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            gbr[t] += ((va[t][0] + va[t][1]) + (va[t][2] + va[t][3])) + ((vb[t][0] + vb[t][1]) + (vb[t][2] + vb[t][3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              va[t][j] *= un8a[j];
              vb[t][j] *= un8b[j];
            }
            float m = fmaxf(fmaxf(fabsf(va[t][0]), fabsf(va[t][1])), fmaxf(fabsf(va[t][2]), fabsf(va[t][3])));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(vb[t][0]), fabsf(vb[t][1])), fmaxf(fabsf(vb[t][2]), fabsf(vb[t][3]))));
            m = rows4_allmax(m, lane);
            const uint32_t e = (__float_as_uint(m) >> 23) & 255u;
            want[t] = (e >= 11u && e < 255u) ? 265u - e : cur[t];
            change |= want[t] < cur[t];
          }
          FC_B5_MARK(6);    // gW passes 1 + 2
          if (__builtin_amdgcn_ballot_w64(change) != 0) {      // rare: some feature of some tile outgrew its shift
#pragma unroll
            for (int t = 0; t < T; ++t) {
              float resc = 1.f;
              if (want[t] < cur[t] && cur[t] != 255u) {
                const int dlt = (int)want[t] - (int)cur[t];
                resc = dlt < -126 ? 0.f : __uint_as_float((uint32_t)(127 + dlt) << 23);
              }
              exch[s16] = resc;                                   // (the four lanes of a feature write the same value)
              const f32x4 r4 = *reinterpret_cast<const f32x4*>(exch + 4 * g);
              // The accumulators are rescaled THROUGH the (idle) strip: stored from and re-loaded into their accumulator
              // registers, the multiply on a copy in between.  A vector instruction that writes dw inside this loop makes the
              // register allocator treat all 16 T accumulators as vector-ALU registers for the whole loop (hundreds of bytes of
              // spills); DS instructions take accumulator registers as they are.
              typedef volatile __attribute__((address_space(3))) f32x4* LdsBounce;
              const LdsBounce bounce = (LdsBounce)(strip) + lane;
#pragma unroll
              for (int ht = 0; ht < 4; ++ht) {
                *bounce = dw[t][ht];
                f32x4 tmp = *bounce;
#pragma unroll
                for (int r = 0; r < 4; ++r) tmp[r] *= r4[r];
                *bounce = tmp;
                dw[t][ht] = *bounce;
              }
            }
          }
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const uint32_t c1 = want[t] < cur[t] ? want[t] : cur[t];
            cur[t] = c1;
            const float sc = c1 == 255u ? 1.f : __uint_as_float((c1 - 1u) << 23);          // 2^(c1 - 128)
            u32x4 ph, pl;
            uint32_t h01, l01;
            split2_pair(va[t][0], va[t][1], sc, h01, l01); ph[0] = h01; pl[0] = l01;
            split2_pair(va[t][2], va[t][3], sc, h01, l01); ph[1] = h01; pl[1] = l01;
            split2_pair(vb[t][0], vb[t][1], sc, h01, l01); ph[2] = h01; pl[2] = l01;
            split2_pair(vb[t][2], vb[t][3], sc, h01, l01); ph[3] = h01; pl[3] = l01;
            const f16x8 ah = __builtin_bit_cast(f16x8, ph), al = __builtin_bit_cast(f16x8, pl);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, hth[ht], dw[t][ht], 0, 0, 0);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, htl[ht], dw[t][ht], 0, 0, 0);
#pragma unroll
            for (int ht = 0; ht < 4; ++ht) dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hth[ht], dw[t][ht], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        FC_B5_MARK(7);      // gW pass 3
        if (has_next) wring_request();
        __builtin_amdgcn_sched_barrier(0);
      } else {
        if (has_next) fetch(tile + stride);       // (a wave without spline work still carries its share of the next tile)
      }
      if (has_next) park(buf ^ 1, tile + stride);
      FC_B5_MARK(8);        // park
#ifdef FC_B5_STAMP
      if (tile == tile0 + 10 * stride && sweep == 0) arrive10 = phase_t - stamp_c0;
#endif
      lds_barrier();
      FC_B5_MARK(9);        // barrier 1
      {
        // gx tile (the upstream gradient with this sweep's columns overwritten) and gh tile (the waves' partials in wave
        // order, on top of the previous sweep's)
        float4* og = reinterpret_cast<float4*>(a.gx + tile * R * D);
        const float* gb_ = gbuf + buf * R * (D + 4);
        if (tid < xvec) og[tid] = *reinterpret_cast<const float4*>(gb_ + so0);
        if (tid + NT < xvec) og[tid + NT] = *reinterpret_cast<const float4*>(gb_ + so1);
        if (tid + 2 * NT < xvec) og[tid + 2 * NT] = *reinterpret_cast<const float4*>(gb_ + so2);
        if (tid + 3 * NT < xvec) og[tid + 3 * NT] = *reinterpret_cast<const float4*>(gb_ + so3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r = (tid >> 4) + 16 * j, c = (tid & 15) * 4;
          float4* gh4 = reinterpret_cast<float4*>(a.gh + tile * R * H) + tid + NT * j;
          float4 s = *reinterpret_cast<const float4*>(part + (size_t)r * (H + 4) + c);
#pragma unroll
          for (int w = 1; w < kB5Waves; ++w) {
            const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)w * R + r) * (H + 4) + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
          }
          const float4 v = j == 0 ? ghp0 : ghp1;
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
          *gh4 = s;
        }
      }
      FC_B5_MARK(10);       // write-out
      lds_barrier();        // `part` is rewritten by the next tile
      FC_B5_MARK(11);       // barrier 2
      buf ^= 1;
    }
    if (!active) continue;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      // gb: the reader lanes' sums over their sample octets, one atomic per (dim, parameter) and workgroup
      const float v = rows4_allsum(gbr[t], lane);
      const int s16 = s16_, g = g_;
      const int dim = 4 * grp + (s16 >> 2), prm = 4 * t + (s16 & 3);
      if (g == 0 && dim < a.dt && prm < P) atomicAdd(a.gb + (size_t)dim * PP + prm, v);
      // gW: lane (hidden 16 ht + s16, feature rho = 4 g + r of tile t) = gW[(dim 4 grp + g), param 4 t + r][hidden]; the
      // features' shifts come from their reader lanes
      exch[s16] = cur[t] == 255u ? 0.f : __uint_as_float((255u - cur[t]) << 23);      // 2^-(shift)
      const f32x4 un4 = *reinterpret_cast<const f32x4*>(exch + 4 * g);
      if (dim_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * t + r < P && un4[r] != 0.f) {
#pragma unroll
            for (int ht = 0; ht < 4; ++ht)
              atomicAdd(a.gw + ((size_t)(grp * 4 + g) * PP + 4 * t + r) * H + 16 * ht + s16, dw[t][ht][r] * un4[r]);
          }
      }
    }
  }
#ifdef FC_B5_STAMP
  phase_cyc[12] = __builtin_amdgcn_s_memtime() - stamp_c0;
  phase_cyc[13] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
  __syncthreads();
  if (lane < 15) {
    float v = (float)arrive10;
#pragma unroll
    for (int i = 0; i < 14; ++i) v = lane == i ? (float)phase_cyc[i] : v;
    a.gh[(size_t)blockIdx.x * H + wave * 16 + lane] = v;
  }
#endif
}

template <int K, bool kTails>
hipError_t launch_backward512(const RQParams& q, const BwdArgs& a, hipStream_t stream) {
  if constexpr (GenShape<K, kTails>::T > 8) return hipErrorInvalidValue;
  else {
    const size_t lds = bwd512_lds_bytes(a.D);
    if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
    static PerDeviceOnce attr;
    const hipError_t ea =
        ensure_max_dynamic_lds(attr, reinterpret_cast<const void*>(&rq_fused_backward512_kernel<K, kTails>), 160 * 1024);
    if (ea != hipSuccess) return ea;
    const int64_t cus = device_cu_count();
    const unsigned grid = (unsigned)(cus < a.tiles ? cus : a.tiles);
    hipLaunchKernelGGL((rq_fused_backward512_kernel<K, kTails>), dim3(grid), dim3(kB5Threads), lds, stream, q, 1.f / q.wh_div, a);
    return hipGetLastError();
  }
}

hipError_t launch_backward512_any(int K, bool tails, const RQParams& q, const BwdArgs& a, hipStream_t stream);

}  // namespace fc
