#include <hip/hip_runtime.h>
__global__ void k(float* out, const float* in) {
  float m = in[threadIdx.x];
  // row (16-lane) all-reduce max by rotations
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x128, 0xf, 0xf, false)));
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x124, 0xf, 0xf, false)));
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x122, 0xf, 0xf, false)));
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x121, 0xf, 0xf, false)));
  float l = in[threadIdx.x + 64];
  // sum over the 4 rows of a wave: lanes s, s+16, s+32, s+48
  auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, l), __builtin_bit_cast(unsigned, l), false, false);
  float l2 = l + __builtin_bit_cast(float, threadIdx.x & 32 ? r32[0] : r32[1]);
  auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, l2), __builtin_bit_cast(unsigned, l2), false, false);
  float l3 = l2 + __builtin_bit_cast(float, threadIdx.x & 16 ? r16[0] : r16[1]);
  out[threadIdx.x] = m;
  out[threadIdx.x + 64] = l3;
}
#include <stdio.h>
int main() {
  float hin[128], hout[128];
  for (int i = 0; i < 64; ++i) { hin[i] = (float)((i * 37) % 101); hin[64 + i] = (float)(1 << (i / 16)) * 100.f + (i % 16); }
  float *din, *dout;
  (void)hipMalloc(&din, sizeof(hin)); (void)hipMalloc(&dout, sizeof(hout));
  (void)hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, din);
  (void)hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    float m = 0; for (int j = 0; j < 16; ++j) m = fmaxf(m, hin[(i & ~15) + j]);
    float s = hin[64 + (i & 15)] + hin[64 + 16 + (i & 15)] + hin[64 + 32 + (i & 15)] + hin[64 + 48 + (i & 15)];
    if (hout[i] != m) { bad++; if (bad < 5) printf("max lane %d got %g want %g\n", i, hout[i], m); }
    if (hout[64 + i] != s) { bad++; if (bad < 9) printf("sum lane %d got %g want %g\n", i, hout[64 + i], s); }
  }
  printf("bad = %d\n", bad);
  return 0;
}
