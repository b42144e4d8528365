"""Applies the probe hunks (FC_BWD_PARTIALS, FC_BWD_POISON, FC_BWD_DRAIN, FC_PROBE_HSACO_SHIM) to the fc_rq_fused_backward.h of
commit 44d992e^ (tools/probe/build_old_bwd_variants.sh).  Probe tree only: reads an environment variable, which product code
never does."""
import sys

p = sys.argv[1]
s = open(p).read()


def sub(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)


sub('''            for (int ht = 0; ht < 4; ++ht)
              atomicAdd(a.gw + ((size_t)(grp * 4 + g) * PP + 4 * t + r) * H + 16 * ht + s16, dw[t][ht][r] * un);
''', '''            for (int ht = 0; ht < 4; ++ht) {
              atomicAdd(a.gw + ((size_t)(grp * 4 + g) * PP + 4 * t + r) * H + 16 * ht + s16, dw[t][ht][r] * un);
#ifdef FC_BWD_PARTIALS
              if (kRole == 1 && g == 3 && t == 0) a.gx[tile0 * R * D + (wave * 4 + r) * H + 16 * ht + s16] = dw[t][ht][r] * un;
#endif
            }
''')
sub('''  if (tile0 >= a.tiles) return;
''', '''  if (tile0 >= a.tiles) return;
#ifdef FC_BWD_POISON
  {
    const int words = (int)(bwd_lds_bytes(a.D, kRole) / 4);
    for (int i = threadIdx.x; i < words; i += kGenThreads) reinterpret_cast<uint32_t*>(bsm)[i] = FC_BWD_POISON;
    __syncthreads();
  }
#endif
''')
sub("inline size_t bwd_lds_bytes(int d, int role) {", "__host__ __device__ inline size_t bwd_lds_bytes(int d, int role) {")
sub('''      // ---- spline backward of this lane's two elements -> G in registers ------------------------------------------
''', '''#ifdef FC_BWD_DRAIN
      {
        float drain = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int t = 0; t < T; ++t) drain += acc[b][t][3];
        asm volatile("" ::"v"(drain));
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      // ---- spline backward of this lane's two elements -> G in registers ------------------------------------------
''')
sub('''  hipLaunchKernelGGL((rq_fused_backward_kernel<K, kTails, kRole>), dim3(grid), dim3(kGenThreads), lds, stream, q,
                     1.f / q.wh_div, a);
  return hipGetLastError();''', '''#if defined(FC_PROBE_HSACO_SHIM) && !defined(__HIP_DEVICE_COMPILE__)
  if constexpr (kRole == 1 && K == 8 && kTails) {
    if (const char* path = getenv("FC_PROBE_HSACO")) {
      static hipModule_t mod = nullptr;
      static hipFunction_t fn = nullptr;
      if (!fn) {
        if (hipModuleLoad(&mod, path) != hipSuccess) return hipErrorInvalidValue;
        if (hipModuleGetFunction(&fn, mod, "_ZN2fc24rq_fused_backward_kernelILi8ELb1ELi1EEEvNS_8RQParamsEfNS_7BwdArgsE") != hipSuccess)
          return hipErrorInvalidValue;
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }
      RQParams qq = q;
      float inv = 1.f / q.wh_div;
      BwdArgs aa = a;
      void* params[] = {&qq, &inv, &aa};
      return hipModuleLaunchKernel(fn, grid, 1, 1, kGenThreads, 1, 1, (unsigned)lds, stream, params, nullptr);
    }
  }
#endif
  hipLaunchKernelGGL((rq_fused_backward_kernel<K, kTails, kRole>), dim3(grid), dim3(kGenThreads), lds, stream, q,
                     1.f / q.wh_div, a);
  return hipGetLastError();''')
if "#include <stdlib.h>" not in s:
    s = s.replace("#include <stdint.h>", "#include <stdint.h>\n#include <stdlib.h>", 1)
open(p, "w").write(s)
