// Microbenchmark: cycles per {1 bf16 MFMA + k independent VALU ops} on one SIMD, for both MFMA shapes, one or
// two waves per SIMD.  Answers: how much VALU issue does an MFMA of each shape hide on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/build/mfma_valu_overlap tools/probe/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int KV, bool TRANS>
__global__ void k(float* out, uint64_t* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 3 + j); }
  f32x4 c4[4] = {};
  f32x16 c16[2] = {};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
  const float mulc = 1.0001f + threadIdx.x * 1e-9f;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (SHAPE == 16) c4[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[u], 0, 0, 0);
      else if (SHAPE == 32) c16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[u & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < KV; ++j) {
        // inline asm: keeps the ops scalar (no SLP packing) and in place
        if (TRANS && j == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(v[j & 7]) : "v"(mulc));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();   // the slowest wave of the workgroup sets the time
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int u = 0; u < 4; ++u) s += c4[u][0] + c4[u][3];
  s += c16[0][0] + c16[1][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int KV, bool TRANS>
void run(int threads, float* out, uint64_t* cyc) {
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL((k<SHAPE, KV, TRANS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<SHAPE, KV, TRANS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  uint64_t h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < blocks; ++i) m += (double)h[i];
  m /= blocks;
  printf("shape %2d  valu/mfma %d%s  waves/SIMD %d : %.1f cycles per (MFMA + VALU group) per wave, %.1f per SIMD-unit\n", SHAPE, KV,
         TRANS ? " (1 trans)" : "", threads / 256, m / (iters * 4.0), m / (iters * 4.0) / (threads / 256));
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  for (int threads : {256, 512}) {
    run<0, 4, false>(threads, out, cyc);  run<0, 8, false>(threads, out, cyc);
    run<16, 0, false>(threads, out, cyc); run<16, 1, false>(threads, out, cyc); run<16, 2, false>(threads, out, cyc);
    run<16, 3, false>(threads, out, cyc); run<16, 4, false>(threads, out, cyc); run<16, 5, false>(threads, out, cyc);
    run<16, 6, false>(threads, out, cyc); run<16, 8, false>(threads, out, cyc); run<16, 5, true>(threads, out, cyc);
    run<32, 0, false>(threads, out, cyc); run<32, 2, false>(threads, out, cyc); run<32, 4, false>(threads, out, cyc);
    run<32, 6, false>(threads, out, cyc); run<32, 8, false>(threads, out, cyc); run<32, 10, false>(threads, out, cyc);
    run<32, 12, false>(threads, out, cyc); run<32, 16, false>(threads, out, cyc); run<32, 10, true>(threads, out, cyc);
  }
  return 0;
}
