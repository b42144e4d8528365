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

}  // namespace fc
