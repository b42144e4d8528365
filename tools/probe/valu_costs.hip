// Issue cost of VALU instructions in a VALU-only stream on gfx950, at 1 and 2 waves per SIMD:
// cycles per instruction per wave = d(s_memtime) / (instructions per wave), all SIMDs of one CU busy.
// Build: hipcc -O2 --offload-arch=gfx950 tools/probe/valu_costs.hip -o tools/probe/build/valu_costs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, uint64_t* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + .5f, a5 = a0 + .25f, a6 = a0 + 4.f, a7 = a0 + 5.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  const float c = 1.0001f;
  const f2 pc = {c, c};
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    if constexpr (KIND == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));) }
    if constexpr (KIND == 2) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));) }
    if constexpr (KIND == 3) { REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));) }
    if constexpr (KIND == 4) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0));) }
    if constexpr (KIND == 5) { REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
    if constexpr (KIND == 6) { REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
    if constexpr (KIND == 7) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if constexpr (KIND == 8) { REP8(asm volatile("v_max3_f32 %0, %0, %4, %1\n v_max3_f32 %1, %1, %4, %2\n v_max3_f32 %2, %2, %4, %3\n v_max3_f32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    if constexpr (KIND == 9) { REP8(asm volatile("v_cmp_ge_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_ge_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc");) }
    if constexpr (KIND == 10) { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if constexpr (KIND == 11) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    if constexpr (KIND == 12) { REP8(asm volatile("v_log_f32 %0, %0\n v_rcp_f32 %1, %1\n v_log_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if constexpr (KIND == 13) { REP8(asm volatile("v_cvt_f32_f16 %0, %0\n v_cvt_f32_f16 %1, %1\n v_cvt_f32_f16 %2, %2\n v_cvt_f32_f16 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if constexpr (KIND == 14) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if constexpr (KIND == 15) { REP8(asm volatile("v_fma_mixlo_f16 %0, %1, %4, 0\n v_fma_mixhi_f16 %0, %2, %4, 0\n v_fma_mixlo_f16 %3, %1, %4, 0\n v_fma_mixhi_f16 %3, %2, %4, 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    if constexpr (KIND == 16) { REP8(asm volatile("v_fma_mix_f32 %0, %1, %4, %2\n v_fma_mix_f32 %1, %2, %4, %3\n v_fma_mix_f32 %2, %3, %4, %0\n v_fma_mix_f32 %3, %0, %4, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    if constexpr (KIND == 17) { REP8(asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y + (float)(d0 + d1 + d2 + d3);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, int threads) {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8192 * 8);
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  uint64_t h[8192];
  const int nw = blocks * threads / 64;
  hipMemcpy(h, cyc, nw * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < nw; ++i) s += h[i];
  const double per = s / nw / (iters * 32.0);
  printf("%-28s waves/SIMD %d: %.2f cycles per instruction per wave -> %.2f SIMD cycles per instruction\n", name, threads / 256, per, per / (threads / 256));
  hipFree(out); hipFree(cyc);
}

#define BOTH(K, name) run<K>(name, 256); run<K>(name, 512);
int main() {
  BOTH(0, "v_fma_f32") BOTH(1, "v_pk_fma_f32") BOTH(2, "v_pk_mul_f32") BOTH(3, "v_pk_add_f32") BOTH(4, "v_add_f64")
  BOTH(5, "v_cvt_f64_f32") BOTH(6, "v_cvt_f32_f64") BOTH(7, "v_exp_f32") BOTH(8, "v_max3_f32") BOTH(9, "v_cmp+v_cndmask")
  BOTH(10, "v_cvt_pk_f16_f32") BOTH(11, "v_mul_f32/v_add_f32") BOTH(12, "v_log_f32/v_rcp_f32") BOTH(13, "v_cvt_f32_f16") BOTH(14, "v_mov_b32_dpp")
  BOTH(15, "v_fma_mixlo/hi_f16") BOTH(16, "v_fma_mix_f32") BOTH(17, "v_cvt_f32_f16_sdwa")
  run<0>("v_fma_f32 (4 waves/SIMD)", 1024); run<15>("v_fma_mixlo/hi_f16 (4 waves/SIMD)", 1024); run<11>("v_mul/v_add e32 (4 waves/SIMD)", 1024); run<10>("v_cvt_pk_f16_f32 (4 waves/SIMD)", 1024);
  return 0;
}
