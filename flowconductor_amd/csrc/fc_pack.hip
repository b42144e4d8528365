// Weight packing for the matrix-core kernels, on the device: f32 nn.Linear tensors -> power-of-two scaled, two-piece f16
// split (fc_split.h), matrix-core A-fragment order.  During training the weights change every step; done with tensor ops
// on the host side this costs ~100 tiny launches per coupling layer, here it is one launch per weight set.
//
// One workgroup per SCALE GROUP (the rows that share one power-of-two scale): it takes the maximum of the group's
// source elements, then writes the group's fragments.  Every fragment element is W[row][col] (or 0 outside the matrix)
// for index functions that depend on the layout (`mode`):
//   FC_PACK_FINAL       fc_rq_spline_fused_general / fc_rq_fused_linear_backward forward fragments
//                       [group][ks][t][piece][lane][8]:  row = (4 group + (rho >> 2)) P + 4 t + (rho & 3), col = 32 ks + 8 gk + j
//   FC_PACK_FINAL_T     W^T fragments of the backward product gh = W^T G
//                       [group][ht][kk][piece][lane][8]: row = (4 group + gk) P + 8 kk + j,               col = 16 ht + rho
//   FC_PACK_HIDDEN      fc_resnet_hidden_backward forward fragments of one [64, K] layer (rows in accumulator order)
//                       [ks][t][piece][lane][8]:         row = feat(t, rho),                              col = 32 ks + 8 gk + j
//   FC_PACK_HIDDEN_T    fragments of the transposed layer (rows = in-features in accumulator order, k = out-features)
//                       [ks][t][piece][lane][8]:         row = 32 ks + 8 gk + j,                          col = feat(t, rho)
//   FC_PACK_HIDDEN_T0   W0^T (rows = identity features in natural order)
//                       [ks][t][piece][lane][8]:         row = 32 ks + 8 gk + j,                          col = 16 t + rho
// with lane = 16 gk + rho and feat(t, rho) = 32 (t >> 1) + 8 (rho >> 2) + 4 (t & 1) + (rho & 3).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_split.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct PackJob {
  const float* w;       // source matrix, row-major
  const float* b;       // source bias or null
  _Float16* frag;       // this group's fragments
  float* unscale;       // this group's 2^-S
  float* bias_out;      // packed bias or null
  int rows, cols;       // valid extent of w
  int mode;
  int p, pp;            // FINAL*: parameters per dim P, padded 4T
  int nks, nt;          // k-steps and tiles of the fragment image ([ks][t] or, FINAL_T, [ht = nt][kk = nks])
  int group;            // FINAL*: group index (dims 4 group .. 4 group + 3)
};

__device__ __forceinline__ int pack_feat(int t, int rho) { return 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3); }

// fragment element e of a job -> (row, col) of the source
__device__ __forceinline__ void pack_index(const PackJob& j, int e, int& row, int& col) {
  const int jj = e & 7, lane = (e >> 3) & 63, frag = e >> 9;        // e = (frag * 64 + lane) * 8 + j (per piece)
  const int rho = lane & 15, gk = lane >> 4;
  switch (j.mode) {
    case FC_PACK_FINAL: {
      const int t = frag % j.nt, ks = frag / j.nt;
      const int param = 4 * t + (rho & 3);
      row = param < j.p ? (4 * j.group + (rho >> 2)) * j.p + param : -1;
      col = 32 * ks + 8 * gk + jj;
      break;
    }
    case FC_PACK_FINAL_T: {
      const int kk = frag % j.nks, ht = frag / j.nks;
      const int param = 8 * kk + jj;
      row = param < j.p ? (4 * j.group + gk) * j.p + param : -1;
      col = 16 * ht + rho;
      break;
    }
    case FC_PACK_HIDDEN: {
      const int t = frag % j.nt, ks = frag / j.nt;
      row = pack_feat(t, rho);
      col = 32 * ks + 8 * gk + jj;
      break;
    }
    case FC_PACK_HIDDEN_T: {
      const int t = frag % j.nt, ks = frag / j.nt;
      row = 32 * ks + 8 * gk + jj;
      col = pack_feat(t, rho);
      break;
    }
    default: {   // FC_PACK_HIDDEN_T0
      const int t = frag % j.nt, ks = frag / j.nt;
      row = 32 * ks + 8 * gk + jj;
      col = 16 * t + rho;
      break;
    }
  }
}

__global__ __launch_bounds__(256) void pack_kernel(const PackJob* jobs) {
  const PackJob j = jobs[blockIdx.x];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int nfrag = j.nks * j.nt;
  const int total = nfrag * 64 * 8;          // elements per piece
  float m = 0.f;
  for (int e = tid; e < total; e += 256) {
    int row, col;
    pack_index(j, e, row, col);
    const float v = (row >= 0 && row < j.rows && col < j.cols) ? j.w[(size_t)row * j.cols + col] : 0.f;
    m = fmaxf(m, fabsf(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sc, un;
  pow2_scale(m, sc, un);
  if (tid == 0) *j.unscale = un;
  for (int e = tid; e < total; e += 256) {
    int row, col;
    pack_index(j, e, row, col);
    const float v = (row >= 0 && row < j.rows && col < j.cols) ? j.w[(size_t)row * j.cols + col] : 0.f;
    _Float16 ph, pl;
    split2(v * sc, ph, pl);
    const int jj = e & 7, lane = (e >> 3) & 63, frag = e >> 9;
    j.frag[((size_t)(frag * 2 + 0) * 64 + lane) * 8 + jj] = ph;
    j.frag[((size_t)(frag * 2 + 1) * 64 + lane) * 8 + jj] = pl;
  }
  if (j.bias_out) {
    if (j.mode == FC_PACK_FINAL) {          // [4][pp] of this group
      for (int i = tid; i < 4 * j.pp; i += 256) {
        const int dim = 4 * j.group + i / j.pp, param = i % j.pp;
        const int idx = dim * j.p + param;
        j.bias_out[i] = (param < j.p && idx < j.rows) ? j.b[idx] : 0.f;
      }
    } else if (j.mode == FC_PACK_HIDDEN) {  // accumulator order: [g][4 t + r] = b[feat(t, g, r)]
      for (int i = tid; i < 64; i += 256) {
        const int g = i >> 4, t = (i >> 2) & 3, r = i & 3;
        const int f = 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r;
        j.bias_out[i] = f < j.rows ? j.b[f] : 0.f;
      }
    }
  }
}

}  // namespace fc

// jobs: DEVICE array of fc::PackJob (the host fills a pinned / device copy: plain pointers and ints)
extern "C" int fc_pack_fragments(const void* jobs, int32_t num_jobs, void* stream) {
  if (num_jobs < 0 || (num_jobs > 0 && !jobs)) return hipErrorInvalidValue;
  if (num_jobs == 0) return hipSuccess;
  hipLaunchKernelGGL(fc::pack_kernel, dim3((unsigned)num_jobs), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const fc::PackJob*>(jobs));
  return hipGetLastError();
}

extern "C" int fc_pack_job_bytes(void) { return (int)sizeof(fc::PackJob); }
