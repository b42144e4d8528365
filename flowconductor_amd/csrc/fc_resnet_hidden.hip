// Hidden layers of the ResidualNet conditioner as ONE kernel on the exact-f32 matrix cores, gfx950.
//
//   h = W0 x_id + b0;   for each block:  h += W2 relu(W1 relu(h) + b1) + b2          -> h [N, 64]
//
// (flowcon/nn/nets/resnet.py:39-53, 93-99: initial_layer, then pre-activation residual blocks; no
//  context, no batch norm, dropout inactive.)  In PyTorch this is 5 small GEMMs + 9 element-wise
//  kernels + a gather per layer, each a full pass over [N, 64] in HBM; here the only HBM traffic is
//  the x rows in and the h rows out.  The conditioner stays a PyTorch nn.Module (parameters,
//  state_dict, CPU execution); this kernel is the device fast path for its inference forward when
//  the shapes match: hidden = 64, <= 2 blocks, ReLU, <= 64 (even) input features.
//
// One 256-thread workgroup owns 64 rows; wave w computes the 32x32 output tile (row block w>>1,
// column tile w&1) of every layer with v_mfma_f32_32x32x2_f32.  All layer weights live in the wave's
// registers as B fragments for the whole kernel (16 + 4*32 = 144 VGPRs at two blocks).  The residual
// stream h never leaves the accumulator registers: W2's product is accumulated straight onto it.
// Only relu(.) activations cross waves, through a double-buffered LDS tile that turns the MFMA C
// layout (column on the lane) into the A layout (row on the lane).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flowcon_hip.h"

namespace fc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHidRows = 64;
constexpr int kHid = 64;
constexpr int kHidStride = kHid + 1;

struct HiddenArgs {
  const float* x;         // [N, D]
  float* h;               // [N, 64]
  const int32_t* id_cols; // [k0] identity columns of x feeding the conditioner
  const float* w0;        // [2 col tiles][64 lanes][k0/2]   B fragments of initial_layer.weight [64, k0]
  const float* b0;        // [64]
  const float* wb;        // [blocks][2 linears][2 col tiles][64 lanes][32]
  const float* bb;        // [blocks][2][64]
  int64_t tiles;
  int D;
  int k0;                 // conditioner input features (even, <= 64)
};

__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int NB>
__global__ __launch_bounds__(256) void resnet_hidden_kernel(HiddenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, Dp = D + 1, k0 = a.k0;
  float* xbuf = smem;                                   // [64][D+1]
  float* abuf = xbuf + ((kHidRows * Dp + 3) & ~3);       // [2][64][65]
  int* ids = reinterpret_cast<int*>(abuf + 2 * kHidRows * kHidStride);  // [k0]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rb = wave >> 1, ct = wave & 1;
  const int arow = rb * 32 + (lane & 31);   // A-operand row of this lane
  const int khalf = lane >> 5;               // A/B-operand k offset of this lane within a k-step
  const int ccol = ct * 32 + (lane & 31);    // C-layout column of this lane
  for (int i = tid; i < k0; i += 256) ids[i] = a.id_cols[i];

  // ---- resident weights -------------------------------------------------------------------------
  float w0r[32];
  {
    const float* p = a.w0 + ((int64_t)ct * 64 + lane) * (k0 >> 1);
#pragma unroll
    for (int s = 0; s < 32; ++s) w0r[s] = s < (k0 >> 1) ? p[s] : 0.f;
  }
  float wr[NB > 0 ? NB : 1][2][32];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      const float4* p = reinterpret_cast<const float4*>(a.wb + ((((int64_t)b * 2 + l) * 2 + ct) * 64 + lane) * 32);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = p[q];
        wr[b][l][4 * q] = v.x; wr[b][l][4 * q + 1] = v.y; wr[b][l][4 * q + 2] = v.z; wr[b][l][4 * q + 3] = v.w;
      }
    }
  const float bias0 = a.b0[ccol];
  float biasb[NB > 0 ? NB : 1][2];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    biasb[b][0] = a.bb[(b * 2 + 0) * 64 + ccol];
    biasb[b][1] = a.bb[(b * 2 + 1) * 64 + ccol];
  }
  __syncthreads();

  const int xvec = kHidRows * D / 4;  // float4 per tile; D % 4 == 0, D <= 128 -> at most 8 per thread
  float4 xv[8];
  auto fetch_x = [&](int64_t t) {
    const float4* xg = reinterpret_cast<const float4*>(a.x + t * kHidRows * D);
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = xg[tid + k * 256 < xvec ? tid + k * 256 : 0];
  };
  auto park_x = [&]() {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + k * 256;
      if (i < xvec) {
        const int e = i * 4, r = e / D, c = e - r * D;
        float* dst = xbuf + r * Dp + c;
        dst[0] = xv[k].x; dst[1] = xv[k].y; dst[2] = xv[k].z; dst[3] = xv[k].w;
      }
    }
  };
  // relu(v + bias) of a C-layout accumulator into activation buffer `buf`
  auto put_act = [&](const f32x16& v, float bias, int buf) {
    float* dst = abuf + buf * kHidRows * kHidStride + (rb * 32) * kHidStride + ccol;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[c_row(r, lane) * kHidStride] = fmaxf(v[r] + bias, 0.f);
  };
  // acc += W . act[buf]   (32 k-steps over the 64 activations)
  auto gemm_act = [&](f32x16 acc, const float (&w)[32], int buf) {
    const float* src = abuf + buf * kHidRows * kHidStride + arow * kHidStride + khalf;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(src[2 * s], w[s], acc, 0, 0, 0);
    return acc;
  };

  const int64_t stride = gridDim.x;
  int64_t tile = blockIdx.x;
  if (tile < a.tiles) fetch_x(tile);
  for (; tile < a.tiles; tile += stride) {
    park_x();
    __syncthreads();
    if (tile + stride < a.tiles) fetch_x(tile + stride);  // travels during the five GEMM stages

    // initial layer: h = W0 x_id (+ b0 folded in below)
    f32x16 hacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
      const float* xr = xbuf + arow * Dp;
#pragma unroll
      for (int s = 0; s < 32; ++s)
        if (2 * s < k0) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[ids[2 * s + khalf]], w0r[s], hacc, 0, 0, 0);
    }
    float hbias = bias0;  // bias not yet added into hacc (kept out of the accumulator start value)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      put_act(hacc, hbias, 0);
      __syncthreads();
      f32x16 tacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      tacc = gemm_act(tacc, wr[b][0], 0);
      put_act(tacc, biasb[b][0], 1);
      __syncthreads();
      hacc = gemm_act(hacc, wr[b][1], 1);
      hbias += biasb[b][1];
    }
    // h rows out: for a fixed register the 32 lanes of a half-wave hold 32 consecutive floats of a row
    float* hg = a.h + (tile * kHidRows + rb * 32) * kHid + ccol;
#pragma unroll
    for (int r = 0; r < 16; ++r) hg[c_row(r, lane) * kHid] = hacc[r] + hbias;
    __syncthreads();  // xbuf / abuf are rewritten by the next tile
  }
}

}  // namespace fc

extern "C" int fc_resnet_hidden(const float* x, float* h, const int32_t* id_cols, const float* w0_frag,
                                const float* b0, const float* wb_frag, const float* bb, int64_t n, int32_t d,
                                int32_t in_features, int32_t hidden, int32_t num_blocks, void* stream) {
  if (n < 0 || d <= 0 || hidden != fc::kHid || num_blocks < 0 || num_blocks > 2) return hipErrorInvalidValue;
  if (in_features <= 0 || in_features > 64 || (in_features & 1) || in_features > d) return hipErrorInvalidValue;
  if (d % 4 != 0 || d > 128 || n % fc::kHidRows != 0) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  if (!x || !h || !id_cols || !w0_frag || !b0 || (num_blocks > 0 && (!wb_frag || !bb))) return hipErrorInvalidValue;
  if ((((uintptr_t)x | (uintptr_t)wb_frag) & 15u) != 0) return hipErrorInvalidValue;
  fc::HiddenArgs a{x, h, id_cols, w0_frag, b0, wb_frag, bb, n / fc::kHidRows, d, in_features};
  const size_t lds = sizeof(float) * (size_t)(((fc::kHidRows * (d + 1) + 3) & ~3) + 2 * fc::kHidRows * fc::kHidStride) +
                     sizeof(int) * 64;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    cus = prop.multiProcessorCount;
  int64_t grid = (int64_t)cus * 2;
  if (grid > a.tiles) grid = a.tiles;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (num_blocks) {
    case 0: hipLaunchKernelGGL(fc::resnet_hidden_kernel<0>, dim3((unsigned)grid), dim3(256), lds, s, a); break;
    case 1: hipLaunchKernelGGL(fc::resnet_hidden_kernel<1>, dim3((unsigned)grid), dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL(fc::resnet_hidden_kernel<2>, dim3((unsigned)grid), dim3(256), lds, s, a); break;
  }
  return hipGetLastError();
}
