// Cross-lane reductions on the VALU (DPP row rotations, gfx950 permlane swaps) instead of ds_bpermute round
// trips through the LDS pipe.  Semantics checked on hardware by tools/probe/dpp_permlane_check.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace fc {

// max over the 16 lanes of a DPP row (lanes 16r .. 16r+15); every lane of the row gets the result
// (v_max_f32 with a DPP operand, one instruction per step; written as asm because the builtin route costs a zeroing
// move, a DPP move, a canonicalising max and the max itself per step.  The s_nop 1 is the VALU-write -> DPP-read
// hazard the compiler cannot see inside an asm block.)
__device__ __forceinline__ float row16_allmax(float m) {
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "+v"(m));
  return m;
}

// sum over the 16 lanes of a DPP row; every lane of the row gets the result
__device__ __forceinline__ float row16_allsum(float v) {
#define FC_ROR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
  v += FC_ROR(8);
  v += FC_ROR(4);
  v += FC_ROR(2);
  v += FC_ROR(1);
#undef FC_ROR
  return v;
}

// value of lane ^ 32 / lane ^ 16
__device__ __forceinline__ float lane_xor32(float v, int lane) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (lane & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor16(float v, int lane) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (lane & 16) ? r[0] : r[1]);
}

// sum / max over the four lanes s, s + 16, s + 32, s + 48 of a wave; every lane gets the result
__device__ __forceinline__ float rows4_allsum(float v, int lane) {
  v += lane_xor32(v, lane);
  return v + lane_xor16(v, lane);
}
// sum over all 64 lanes, every lane gets it: four DPP row rotations + two permlane swaps, all on the VALU (a
// __shfl_xor butterfly is six dependent ds_bpermute round trips through the LDS pipe)
__device__ __forceinline__ float wave64_allsum(float v, int lane) { return rows4_allsum(row16_allsum(v), lane); }

__device__ __forceinline__ float rows4_allmax(float v, int lane) {
  v = fmaxf(v, lane_xor32(v, lane));
  return fmaxf(v, lane_xor16(v, lane));
}
// max over all 64 lanes, every lane gets it
__device__ __forceinline__ float wave64_allmax(float v, int lane) { return rows4_allmax(row16_allmax(v), lane); }

// Merge steps of the batched reductions below.  v_permlane32_swap / v_permlane16_swap with two DIFFERENT registers: after
// swap(a, b) one register holds a's lower half (even rows) next to b's, the other the upper halves (odd rows), so their sum is
// a's pair sums in one half of the wave (rows 0, 2) and b's in the other (rows 1, 3).
// (inline asm: with two different operands hipcc 7.2's __builtin_amdgcn_permlane32_swap hands back its first result twice --
//  tools/probe/sum8_check.hip; the s_nop 1 is the VALU-write -> cross-lane-read hazard the compiler cannot see inside an asm block)
__device__ __forceinline__ float lane_merge32(float a, float b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;     // lanes 0-31: a[l] + a[l + 32], lanes 32-63: b[l - 32] + b[l]
}
__device__ __forceinline__ float lane_merge16(float a, float b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;     // rows 0, 2: a's (row r) + (row r + 1), rows 1, 3: b's (row r - 1) + (row r)
}
// Four values per lane, each to be summed over the four lanes s, s + 16, s + 32, s + 48: three merge steps instead of four
// rows4_allsum (eight swaps and as many selects).  Lane (s, row r) gets the sum of value rows4_sum4_index(r).
__device__ __forceinline__ float rows4_sum4(float v0, float v1, float v2, float v3) {
  return lane_merge16(lane_merge32(v0, v1), lane_merge32(v2, v3));      // rows: v0, v2, v1, v3
}
__device__ __forceinline__ int rows4_sum4_index(int row) { return ((row & 1) << 1) | (row >> 1); }

// Eight wave sums at once.  v_permlane32_swap / v_permlane16_swap with two DIFFERENT registers are a merge step: after
// swap(a, b) one result holds a's lower half next to b's lower half and the other the two upper halves, so their sum is
// a's pair sums in one half of the wave and b's in the other -- two vectors become one per swap + add, no selects.  8 -> 4
// (xor 32) -> 2 (xor 16), then the two vectors' 16-lane rows each hold one sum's partials: four DPP rotations each.
// 6 swaps + 6 adds + 8 DPP adds for eight sums (eight separate wave64_allsum: 64 cross-lane steps).  The sums come back
// wave-uniform (v_readlane).
__device__ __forceinline__ void wave64_sum8(const float (&s)[8], float (&out)[8]) {
#ifdef FC_SUM8_SIMPLE   // probe builds: eight separate reductions
  for (int k = 0; k < 8; ++k) out[k] = wave64_allsum(s[k], threadIdx.x & 63);
  return;
#endif
  auto merge32 = [](float a, float b) { return lane_merge32(a, b); };
  auto merge16 = [](float a, float b) { return lane_merge16(a, b); };
  const float m01 = merge32(s[0], s[1]), m23 = merge32(s[2], s[3]), m45 = merge32(s[4], s[5]), m67 = merge32(s[6], s[7]);
  // rows of the merged vectors: (s0, s2, s1, s3) and (s4, s6, s5, s7)
  const float lo = row16_allsum(merge16(m01, m23)), hi = row16_allsum(merge16(m45, m67));
  auto lane_of = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
  out[0] = lane_of(lo, 0);  out[2] = lane_of(lo, 16); out[1] = lane_of(lo, 32); out[3] = lane_of(lo, 48);
  out[4] = lane_of(hi, 0);  out[6] = lane_of(hi, 16); out[5] = lane_of(hi, 32); out[7] = lane_of(hi, 48);
}

}  // namespace fc
