// wave64_sum8 (fc_lane.h) against eight separate sums, with the intermediate vectors of the first merge steps.
//   hipcc --offload-arch=gfx950 -O3 -I flowconductor_amd/csrc -o /tmp/sum8 tools/probe/sum8_check.hip && /tmp/sum8
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fc_lane.h"
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  float s[8], t[8];
  for (int i = 0; i < 8; ++i) s[i] = (float)((lane * 7 + i * 13) % 11) + 0.25f * i;
  fc::wave64_sum8(s, t);
  for (int i = 0; i < 8; ++i) {
    out[i * 64 + lane] = t[i];
    out[512 + i * 64 + lane] = fc::wave64_allsum(s[i], lane);
  }
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s[0]), __builtin_bit_cast(unsigned, s[1]), false, false);
  out[1024 + lane] = s[0]; out[1088 + lane] = s[1];
  out[1152 + lane] = __builtin_bit_cast(float, r[0]); out[1216 + lane] = __builtin_bit_cast(float, r[1]);
}
int main() {
  float* d; static float h[1280];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("sum %d: batched %g (lane 37: %g)  separate %g\n", i, h[i * 64], h[i * 64 + 37], h[512 + i * 64]);
  for (int l = 0; l < 64; l += 9) printf("lane %d: s0 %g s1 %g r0 %g r1 %g\n", l, h[1024 + l], h[1088 + l], h[1152 + l], h[1216 + l]);
  return 0;
}
