// v_permlane32_swap / v_permlane16_swap with two different operands: which lanes of which result hold what.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pswap tools/probe/permlane_swap2.hip && /tmp/pswap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  const unsigned a = lane, b = 100 + lane;
  const auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r32[0]; out[64 + lane] = r32[1]; out[128 + lane] = r16[0]; out[192 + lane] = r16[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 r[0]", "swap32 r[1]", "swap16 r[0]", "swap16 r[1]"};
  for (int v = 0; v < 4; ++v) {
    printf("%s:", names[v]);
    for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * v + l]);
    printf("\n");
  }
  return 0;
}
