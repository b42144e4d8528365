// f32 products on the f16 matrix cores: exact power-of-two scaling + two-piece f16 split.
//
// v_mfma_f32_*_f32 (f32 in, f32 out) runs at the f32 VALU rate and does not overlap with VALU work; the
// 16-bit matrix pipe is 16x faster.  A value x, scaled by a power of two so that the maximum of its row
// sits in [2^14, 2^15), is split as x = xh + xl + e, xh = f16(x), xl = f16(x - xh), |e| <= 2^-22 max|x|;
// a product sum a.b is taken as  al.bh + ah.bl + ah.bh  (the al.bl term, <= 2^-22 |a||b|, is dropped),
// three f16 MFMAs accumulating in f32.  Against float64 the error of a 64-term product is max 1.7e-7 /
// rms 2.2e-8 of sum|a||b|; an f32 GEMM's own rounding is 4.1e-7 / 3.4e-8 (tools/probe/split_accuracy.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Power-of-two scale that lifts m = max|x| into [2^14, 2^15) (f16 tops out at 65504), and its inverse.  Tiny or zero
// maxima are left alone (their pieces underflow to an absolute error far below f32 resolution of any O(1) result).
// Why the top of the range: the two pieces carry 22 bits of a value as long as its low piece stays above f16's subnormal
// quantum 2^-24, i.e. down to values 2^16 times smaller than the row maximum (one bit less per further factor of two).
// Dense products do not care (the large entry dominates the sum anyway), MASKED ones do: a MADE unit that only reads the
// small columns of a row must see them at full precision beside a large column it does not read (rounds 1-3 lifted the
// maximum to [2^10, 2^11): full precision only down to 2^12 below it -- tools/probe/fuzz_ar_inverse.py found rows of an
// affine autoregressive inverse, |y| up to 8e3 beside O(1) columns, 4e-3 off).
#ifndef FC_SPLIT_TOP_EXP
#define FC_SPLIT_TOP_EXP 14      // (probe builds: 10 = the scale of rounds 1-3)
#endif
constexpr uint32_t kSplitTopExp = FC_SPLIT_TOP_EXP;
__device__ __forceinline__ void pow2_scale(float m, float& scale, float& unscale) {
  const uint32_t e = (__float_as_uint(m) >> 23) & 255u;      // biased exponent, floor(log2 m) = e - 127
  const bool ok = e > kSplitTopExp && e < 255u;
  scale = ok ? __uint_as_float((254u + kSplitTopExp - e) << 23) : 1.f;       // 2^(14 - (e - 127))
  unscale = ok ? __uint_as_float((e - kSplitTopExp) << 23) : 1.f;
}

// x = h + l (+ residual <= 2^-22 |x|) with f16 pieces, round-to-nearest-even; the difference is exact in f32
__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) {
  h = (_Float16)x;
  l = (_Float16)(x - (float)h);
}

// The same split for two values x0 = v0 * sc, x1 = v1 * sc (sc a power of two: the products are exact), pieces packed
// as f16 pairs (low half = value 0).  The residual comes from the mixed-precision fma, l = f16(fma(v, sc, -h)) with h
// read as an f16 operand: no conversion of h back to f32 (five instructions per pair instead of six).  Bit-identical to
// split2 (tools/probe/split_pair_check.hip).  [v_fma_mixlo / mixhi_f16, which would also absorb the final conversion,
// issue at the transcendental rate on gfx950 (tools/probe/valu_costs.hip): four of them per pair measured 12 % slower
// in fc_resnet_hidden than the six plain instructions.]
__device__ __forceinline__ void split2_pair(float v0, float v1, float sc, uint32_t& h01, uint32_t& l01) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  const f16x2 h = {(_Float16)(v0 * sc), (_Float16)(v1 * sc)};
  const uint32_t hb = __builtin_bit_cast(uint32_t, h);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %2, %4, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %1, %3, %4, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(r0), "=&v"(r1)
      : "v"(v0), "v"(v1), "v"(sc), "v"(hb));
  const f16x2 l = {(_Float16)r0, (_Float16)r1};
  h01 = hb;
  l01 = __builtin_bit_cast(uint32_t, l);
}

}  // namespace fc
