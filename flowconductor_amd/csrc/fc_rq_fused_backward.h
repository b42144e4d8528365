// Backward of the fused final-Linear + RQ-spline coupling layer (training through the HIP path), gfx950: shared declarations.
//
//   forward:   params = W h + b;   y, logabsdet = rq_spline(x, params)            (fc_rq_fused_general.h)
//   backward:  given gy = dL/dy [N, D], gl = dL/dlogabsdet [N]
//       G[n, f]  = dL/dparams                      (closed-form spline backward, fc_rq_backward.hip)
//       gx       = gy on the identity columns, gy dy/dx + gl dlad/dx on the transformed ones
//       gh[n, :] = W^T G[n, :]                     (gradient into the conditioner's hidden stack)
//       gW       = sum_n G[n, :] (x) h[n, :],   gb = sum_n G[n, :]
//
// What torch.autograd does for the reference (examples/toy_2d.py:57-68) materialises params [N, d_t P] in the forward
// and G [N, d_t P] in the backward and runs three library GEMMs over them: ~9 GB of HBM traffic per layer at N = 2^19.
// Here neither tensor ever exists: the parameters are RECOMPUTED on the matrix cores from the saved h (the forward
// kernel's product, fragment for fragment), the spline backward runs on them in registers, and G goes from the lane's
// registers straight into the two products that consume it.
//
// The kernel is fc_rq_fused_backward512.h (round 4: ONE launch at one wave per SIMD).  Rounds 2-3 ran two launches at two
// waves per SIMD (roles "dx" and "dw", each recomputing G because the wave's gW slice filled its 256 registers; role 1
// spilled 20-200 B per lane) -- that source lives in tools/probe/old_backward/ for the probe builds of DESIGN.md section 4d.
//
// hidden == 64 (the conditioner width fc_resnet_hidden serves), K = 4..16 where the accumulators fit (T <= 8).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc_device.h"
#include "fc_lane.h"
#include "fc_math.h"
#include "fc_rq_op.h"
#include "fc_rq_backward_op.h"
#include "fc_split.h"
#include "fc_tile.h"
#include "fc_rq_fused_general.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct BwdArgs {
  const float* x;          // [N, D]  layer input (saved by the forward)
  const float* h;          // [N, 64] conditioner's last hidden activation (saved by the forward)
  const float* gy;         // [N, D]
  const float* gl;         // [N] or null (zeros)
  const f16x8* wfrag;      // forward fragments  [groups][2][T][2][64]        (ops.pack_final_layer_general, hidden 64)
  const float* wun;        // [groups]
  const float* bias;       // [groups][4][PP]
  const f16x8* wtfrag;     // W^T fragments [groups][4 hidden tiles][KK][2][64]   (ops.pack_final_layer_transposed)
  const int32_t* cols;     // [dt]
  float* gx;               // [N, D]
  float* gh;               // [N, 64]
  float* gb;               // [groups][4][PP], accumulated with atomics (zeroed by the caller)
  float* gw;               // [groups][4][PP][64], accumulated with atomics (zeroed by the caller)
  int64_t tiles;           // 32-row tiles
  int D, dt;
};

// a fragment pointer that went through an opaque asm (to keep it a scalar base) must say it is global memory again: the
// compiler otherwise emits flat loads, which also count on lgkmcnt and are waited for before every LDS read
typedef const __attribute__((address_space(1))) f16x8* GlobalFrags;

// f16 per h row in LDS: 80 (160 B; ds_read_b128 is served in four non-contiguous 16-lane groups: conflict-free rows need a
// stride of 32 mod 64 bytes, tools/lds_conflicts.py)
constexpr int kBwdH = 64, kBwdR = kGenRows;

}  // namespace fc
