// split2_pair (4 instructions per value pair on the mixed-precision fma) against split2 (fc_split.h): the same bits.
// hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -Iflowconductor_amd/csrc tools/probe/split_pair_check.hip -o tools/probe/build/split_pair_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include "fc_split.h"

__global__ void k(const float* v, const float* scs, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float sc = scs[i & 15];
  _Float16 h0, l0, h1, l1;
  fc::split2(v[2 * i] * sc, h0, l0);
  fc::split2(v[2 * i + 1] * sc, h1, l1);
  uint32_t hp, lp;
  fc::split2_pair(v[2 * i], v[2 * i + 1], sc, hp, lp);
  const uint32_t href = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
  const uint32_t lref = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
  out[i] = (hp != href ? 1u : 0u) | (lp != lref ? 2u : 0u);
}

int main() {
  const int n = 1 << 22;
  float* hv = (float*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const float m = (float)rand() / RAND_MAX * 2.f - 1.f;
    const int e = rand() % 40 - 30;          // magnitudes 2^-30 .. 2^9: normal, subnormal and zero f16 pieces
    hv[i] = (i % 97 == 0) ? 0.f : ldexpf(m, e);
  }
  float hs[16];
  for (int i = 0; i < 16; ++i) hs[i] = ldexpf(1.f, i - 4);
  float *dv, *ds; uint32_t* dout;
  hipMalloc(&dv, n * 4); hipMalloc(&ds, 64); hipMalloc(&dout, n * 2);
  hipMemcpy(dv, hv, n * 4, hipMemcpyHostToDevice); hipMemcpy(ds, hs, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dv, ds, dout, n);
  uint32_t* ho = (uint32_t*)malloc(n * 2);
  hipMemcpy(ho, dout, n * 2, hipMemcpyDeviceToHost);
  long bad_h = 0, bad_l = 0;
  for (int i = 0; i < n / 2; ++i) { bad_h += ho[i] & 1; bad_l += (ho[i] >> 1) & 1; }
  printf("pairs %d: high pieces differing %ld, low pieces differing %ld\n", n / 2, bad_h, bad_l);
  return bad_h || bad_l;
}
