// Microbenchmark for the gW product phase of fc_rq_fused_backward512.h (one wave per SIMD): per feature tile 4 split2_pair
// (20 vector instructions) produce an A operand, then 12 f16 MFMAs (4 accumulators x 3 split terms) against fixed B operands.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/build/b5_pass3 tools/probe/b5_pass3.hip && tools/probe/build/b5_pass3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2_pair(float v0, float v1, float sc, uint32_t& h01, uint32_t& l01) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  const f16x2 h = {(_Float16)(v0 * sc), (_Float16)(v1 * sc)};
  const uint32_t hb = __builtin_bit_cast(uint32_t, h);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %2, %4, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %1, %3, %4, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(r0), "=&v"(r1) : "v"(v0), "v"(v1), "v"(sc), "v"(hb));
  const f16x2 l = {(_Float16)r0, (_Float16)r1};
  h01 = hb;
  l01 = __builtin_bit_cast(uint32_t, l);
}

// MODE 0: split + products as the kernel has them; 1: products only; 2: splits only; 3: ACC accumulators only 1 (dependent chain)
template <int MODE, int T>
__global__ __launch_bounds__(256) void k(float* out, uint64_t* cyc, int iters, float scl) {
  f16x8 hth[4], htl[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) { hth[i][j] = (_Float16)(float)((threadIdx.x + i + j) & 7); htl[i][j] = (_Float16)(0.001f * j); }
  f32x4 dw[T][4] = {};
  float v[T][8];
  for (int t = 0; t < T; ++t)
    for (int j = 0; j < 8; ++j) v[t][j] = threadIdx.x * 0.001f + j + t;
  f16x8 fix = hth[0];
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f16x8 ah = fix, al = fix;
      if (MODE != 1) {
        u32x4 ph, pl;
        uint32_t h01, l01;
        const float sc = scl + v[t][0] * 1e-30f;
        split2_pair(v[t][0], v[t][1], sc, h01, l01); ph[0] = h01; pl[0] = l01;
        split2_pair(v[t][2], v[t][3], sc, h01, l01); ph[1] = h01; pl[1] = l01;
        split2_pair(v[t][4], v[t][5], sc, h01, l01); ph[2] = h01; pl[2] = l01;
        split2_pair(v[t][6], v[t][7], sc, h01, l01); ph[3] = h01; pl[3] = l01;
        ah = __builtin_bit_cast(f16x8, ph); al = __builtin_bit_cast(f16x8, pl);
        if (MODE == 2) { v[t][0] += __builtin_bit_cast(float, ph[0] ^ pl[1] ^ ph[2] ^ pl[3] ^ ph[1] ^ pl[0] ^ pl[2] ^ ph[3]) * 1e-30f; }
      }
      if (MODE != 2) {
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, hth[ht], dw[t][ht], 0, 0, 0);
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, htl[ht], dw[t][ht], 0, 0, 0);
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) dw[t][ht] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hth[ht], dw[t][ht], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < T; ++t) {
    for (int j = 0; j < 8; ++j) s += v[t][j];
    for (int ht = 0; ht < 4; ++ht) s += dw[t][ht][0] + dw[t][ht][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int T>
void run(const char* what, float* out, uint64_t* cyc) {
  const int iters = 500, blocks = 256;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, T>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1024.f);
  hipDeviceSynchronize();
  uint64_t h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < blocks; ++i) m += (double)h[i];
  m /= blocks;
  printf("%-40s T=%d: %.0f cycles per pass (%.1f per feature tile)\n", what, T, m / iters, m / iters / T);
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0, 6>("split + 12 products per tile", out, cyc);
  run<1, 6>("12 products per tile only", out, cyc);
  run<2, 6>("splits only", out, cyc);
  run<0, 1>("split + 12 products, one tile", out, cyc);
  run<1, 1>("12 products, one tile (4 accumulators)", out, cyc);
  return 0;
}
