// Probe (round 3, the gW fault of fc_rq_fused_linear_backward): is the DATA register of an LDS store safe to overwrite right
// after the store has been issued?  A wave queues a backlog of wide LDS stores, then issues ONE narrow store of register v and
// overwrites v with another pattern in the very next instructions; after a full drain the stored value is read back.
//   kind 0: ds_write_b16   kind 1: ds_write_b32   kind 2: ds_write_b16, s_nop 7 between the store and the overwrite
// out[kind][quarter of the wave] counts the lanes that found the OVERWRITING pattern in LDS.
//   hipcc --offload-arch=gfx950 -O2 -o lds_b16_war tools/probe/lds_b16_war.hip && ./lds_b16_war
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND, int BACKLOG, int OVW, int MFMA>
__global__ __launch_bounds__(512) void war_kernel(unsigned long long* out, int iters) {
  extern __shared__ uint32_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // per wave: [0, 4096) bytes backlog area (64 lanes x 64 B), then 64 x 4 B slots for the narrow stores
  const uint32_t base = wave * 8192;
  const uint32_t big_addr = base + lane * 64;
  const uint32_t slot = base + 4096 + lane * 4;
  unsigned long long bad[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    uint32_t v = 0x1100u + (uint32_t)((it * 7 + lane) & 0xff);           // pattern A (low 16 bits matter)
    const uint32_t w = 0xee00u + (uint32_t)((it * 13 + lane) & 0xff);     // pattern B, what the register becomes
    const uint32_t expect = v & 0xffffu;
    const float wf = 1.5f + (float)(it & 7);           // as f16: 0x3e00 .. : never equal to pattern A
    if (MFMA) {       // matrix-core work in flight when the store issues (the kernel's gW products of the previous tile)
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
      f16x8 a8, b8;
      for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(lane + j); b8[j] = (_Float16)(it + j); }
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < MFMA; ++j) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
      asm volatile("" :: "v"(c));
    }
    uint32_t r;
    const uint32_t b0 = it, b1 = it + 1, b2 = it + 2, b3 = it + 3;
    asm volatile(
        "v_mov_b32 v40, %[b0]\n\tv_mov_b32 v41, %[b1]\n\tv_mov_b32 v42, %[b2]\n\tv_mov_b32 v43, %[b3]\n\t"
        ".rept %c[n]\n\t"
        "ds_write_b128 %[big], v[40:43]\n\t"
        "ds_write_b128 %[big], v[40:43] offset:16\n\t"
        "ds_write_b128 %[big], v[40:43] offset:32\n\t"
        "ds_write_b128 %[big], v[40:43] offset:48\n\t"
        ".endr\n\t"
        ".if %c[kind] == 1\n\t"
        "ds_write_b32 %[slot], %[v]\n\t"
        ".else\n\t"
        "ds_write_b16 %[slot], %[v]\n\t"
        ".endif\n\t"
        ".if %c[kind] == 2\n\t"
        "s_nop 7\n\ts_nop 7\n\t"
        ".endif\n\t"
        ".if %c[ovw] == 0\n\t"
        "v_mov_b32 %[v], %[w]\n\t"
        ".elseif %c[ovw] == 1\n\t"
        "v_cvt_f16_f32_e32 %[v], %[wf]\n\t"          // a d16 write of the low half, as the kernel's split2 does
        ".else\n\t"
        "v_fma_mixlo_f16 %[v], %[wf], %[wf], 0\n\t"
        ".endif\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_read_u16 %[r], %[slot]\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [v] "+v"(v), [r] "=&v"(r)
        : [big] "v"(big_addr), [slot] "v"(slot), [w] "v"(w), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3),
          [n] "i"(BACKLOG), [kind] "i"(KIND), [ovw] "i"(OVW), [wf] "v"(wf)
        : "v40", "v41", "v42", "v43", "memory");
    if ((r & 0xffffu) != expect) bad[lane >> 4] += 1;
    if (v == 0x12345678u) lds[0] = v;      // keep v alive
  }
  for (int q = 0; q < 4; ++q)
    if (bad[q]) atomicAdd(out + KIND * 4 + q, bad[q]);
}

template <int KIND, int BACKLOG, int OVW = 0, int MFMA = 0>
static void run(unsigned long long* d_out, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&war_kernel<KIND, BACKLOG, OVW, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL((war_kernel<KIND, BACKLOG, OVW, MFMA>), dim3(1024), dim3(512), 64 * 1024, 0, d_out, iters);
}

template <int OVW, int MFMA>
static void more(unsigned long long* d_out) {
  hipMemset(d_out, 0, 12 * sizeof(unsigned long long));
  run<0, 4, OVW, MFMA>(d_out, 2000);
  run<1, 4, OVW, MFMA>(d_out, 2000);
  run<2, 4, OVW, MFMA>(d_out, 2000);
  hipDeviceSynchronize();
  unsigned long long h[12];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  const char* ov[3] = {"v_mov_b32", "v_cvt_f16_f32", "v_fma_mixlo_f16"};
  for (int k = 0; k < 3; ++k)
    printf("overwrite by %-16s %2d MFMAs in flight | kind %d: stale-data lanes per quarter: %llu %llu %llu %llu\n", ov[OVW], MFMA, k,
           h[k * 4], h[k * 4 + 1], h[k * 4 + 2], h[k * 4 + 3]);
}

int main() {
  unsigned long long* d_out;
  hipMalloc(&d_out, 12 * sizeof(unsigned long long));
  for (int backlog = 0; backlog < 3; ++backlog) {
    hipMemset(d_out, 0, 12 * sizeof(unsigned long long));
    const int iters = 2000;
    if (backlog == 0) { run<0, 1>(d_out, iters); run<1, 1>(d_out, iters); run<2, 1>(d_out, iters); }
    if (backlog == 1) { run<0, 4>(d_out, iters); run<1, 4>(d_out, iters); run<2, 4>(d_out, iters); }
    if (backlog == 2) { run<0, 16>(d_out, iters); run<1, 16>(d_out, iters); run<2, 16>(d_out, iters); }
    hipDeviceSynchronize();
    unsigned long long h[12];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const double total = 1024.0 * 512 * iters / 4;
    const char* names[3] = {"ds_write_b16 then overwrite", "ds_write_b32 then overwrite", "ds_write_b16, 16 nops, overwrite"};
    for (int k = 0; k < 3; ++k)
      printf("backlog %2d x 4 wide stores | %-34s stale-data lanes per quarter: %llu %llu %llu %llu (of %.0f each)\n",
             backlog == 0 ? 1 : backlog == 1 ? 4 : 16, names[k], h[k * 4], h[k * 4 + 1], h[k * 4 + 2], h[k * 4 + 3], total);
  }
  more<1, 0>(d_out);
  more<2, 0>(d_out);
  more<0, 8>(d_out);
  more<1, 8>(d_out);
  more<2, 8>(d_out);
  return 0;
}
