// Probe 2 (round 3, gW fault): the strip pattern itself -- a wave issues 16 ds_write_b16 back to back (no waits between them),
// adjacent lanes writing the two halves of one dword, rows 80 / 96 bytes apart as in fc_rq_fused_backward.h, with the data
// registers produced by v_cvt_f16_f32 / v_fma_mixlo_f16 in between and reused; then one drain and the MFMA-operand style
// ds_read_b128 of the rows.  Counts 16-bit elements that differ from what was stored, per quarter of the wave.
//   hipcc --offload-arch=gfx950 -O2 -o lds_b16_pairs tools/probe/lds_b16_pairs.hip && ./lds_b16_pairs
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TS, int MFMA>
__global__ __launch_bounds__(512) void strip_kernel(unsigned long long* out, int iters) {
  extern __shared__ _Float16 sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s16 = lane & 15, g = lane >> 4;
  _Float16* strip = sm + (size_t)wave * 2 * 16 * TS;
  unsigned long long bad[4] = {0, 0, 0, 0};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    float gp[2][4];
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 4; ++r) gp[b][r] = (float)(((it * 31 + lane * 7 + b * 4 + r) & 1023) - 512) * 0.37f;
    for (int t = 0; t < 6; ++t) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = gp[b][r] + (float)t;
          const _Float16 ph = (_Float16)x;
          const _Float16 pl = (_Float16)(x - (float)ph);
          strip[(size_t)(4 * g + r) * TS + 16 * b + s16] = ph;
          strip[(size_t)(16 + 4 * g + r) * TS + 16 * b + s16] = pl;
        }
      const f16x8 ah = *reinterpret_cast<const f16x8*>(strip + (size_t)s16 * TS + 8 * g);
      const f16x8 al = *reinterpret_cast<const f16x8*>(strip + (size_t)(16 + s16) * TS + 8 * g);
      if (MFMA) {
        for (int j = 0; j < MFMA; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ah, acc, 0, 0, 0);
      }
      // what lane (s16 = feature row rho, g = sample chunk) should see: element j = value of lane (g' = rho >> 2, s16' = (8 g + j) & 15)
      // block b' = (8 g + j) >> 4, register r' = rho & 3
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = 8 * g + j, bb = col >> 4, ss = col & 15, gg = s16 >> 2, rr = s16 & 3;
        const int src_lane = gg * 16 + ss;
        const float x = (float)(((it * 31 + src_lane * 7 + bb * 4 + rr) & 1023) - 512) * 0.37f + (float)t;
        const _Float16 ph = (_Float16)x;
        const _Float16 pl = (_Float16)(x - (float)ph);
        if (ah[j] != ph || al[j] != pl) bad[gg] += 1;      // attributed to the WRITING lane's quarter
      }
    }
  }
  if (acc[0] == 12345.f) out[15] = 1;
  for (int q = 0; q < 4; ++q)
    if (bad[q]) atomicAdd(out + q, bad[q]);
}

template <int TS, int MFMA>
static void run(unsigned long long* d_out, const char* what) {
  hipMemset(d_out, 0, 16 * sizeof(unsigned long long));
  hipLaunchKernelGGL((strip_kernel<TS, MFMA>), dim3(1024), dim3(512), 8 * 2 * 16 * TS * 2, 0, d_out, 500);
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s wrong halves by writing quarter: %llu %llu %llu %llu\n", what, h[0], h[1], h[2], h[3]);
}

int main() {
  unsigned long long* d_out;
  hipMalloc(&d_out, 16 * sizeof(unsigned long long));
  run<40, 0>(d_out, "rows 80 B apart, no matrix work");
  run<40, 12>(d_out, "rows 80 B apart, 12 MFMAs per tile");
  run<48, 0>(d_out, "rows 96 B apart, no matrix work");
  run<48, 12>(d_out, "rows 96 B apart, 12 MFMAs per tile");
  return 0;
}
