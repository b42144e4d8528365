// Piecewise rational-quadratic spline bijector, forward + inverse + log|det J|, gfx950.
//
// Behaviour follows (restated, not copied) flowcon/transforms/splines/rational_quadratic.py:
//   :13-63   unconstrained wrapper (linear tails: identity outside [-B, B], derivative pad)
//   :66-181  knots = softmax -> min + (1 - min*K)*p -> cumsum -> affine -> pinned ends -> diffs,
//            derivatives = min_d + softplus(u, beta), compare-count bin search
//            (utils/torchutils.py:147-149), forward rational-quadratic / inverse quadratic root.
// One thread evaluates one (sample, dim); the K unnormalised widths/heights sit in LDS and
// are walked with static offsets so no runtime-indexed register array (-> scratch) exists.
#include <stdlib.h>
#include "fc_tile.h"
#include "fc_math.h"
#include "../../include/flowcon_hip.h"

namespace fc {

struct RQParams {
  int K;
  int tails;        // 0: none (domain [left,right] x [bottom,top]), 1: linear
  int inverse;
  float left, right, bottom, top;
  float min_w, min_h, min_d;
  float cw, ch;     // (float)(1 - min_w*K), (float)(1 - min_h*K), evaluated in double on the host
  float wh_div;     // unnormalised widths/heights are divided by this (coupling.py:554-559); 1 = off
  float beta;       // softplus beta: 1, or ln2/(1-min_d) with enable_identity_init
  float tail_const; // (float)log(exp(1 - min_d) - 1): padded end derivatives for linear tails
};

// Walk the K bins of one cumulative axis. u -> LDS pointer to K unnormalised values.
// search: idx = last bin whose lower knot <= v (== compare-count - 1 for monotone knots).
// select: take bin `idx`. Returns lower knot and bin size of the chosen bin.
template <int KS, bool kSearch>
__device__ __forceinline__ void walk_axis(const float* __restrict__ u, int K, float inv_scale,
                                          float minb, float c1,
                                          float lo, float hi, float v, int& idx, float& knot_lo,
                                          float& bin_size) {
  const float span = hi - lo;
  if (KS > 0) {
    float t[KS > 0 ? KS : 1];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      t[i] = u[i] * inv_scale;
      m = fmaxf(m, t[i]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      t[i] = exp_lean(t[i] - m);
      sum += t[i];
    }
    const float rs = div_lean(1.f, sum);
    double cum = 0.0;  // at::cumsum on the CPU accumulates f32 in double
    float prev = lo;
    int found = kSearch ? 0 : idx;
    float klo = lo, bsz = 0.f;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const float p = t[i] * rs;
      const float w = minb + c1 * p;
      cum += (double)w;
      const float next = (i == KS - 1) ? hi : (span * (float)cum + lo);
      const bool take = kSearch ? (v >= prev) : (i == idx);
      if (take) {
        found = i;
        klo = prev;
        bsz = next - prev;
      }
      prev = next;
    }
    idx = found;
    knot_lo = klo;
    bin_size = bsz;
  } else {
    float m = -INFINITY;
    for (int i = 0; i < K; ++i) {
      const float t = u[i] * inv_scale;
      m = fmaxf(m, t);
    }
    float sum = 0.f;
    for (int i = 0; i < K; ++i) {
      const float t = u[i] * inv_scale;
      sum += exp_lean(t - m);
    }
    const float rs = div_lean(1.f, sum);
    double cum = 0.0;
    float prev = lo;
    int found = kSearch ? 0 : idx;
    float klo = lo, bsz = 0.f;
    for (int i = 0; i < K; ++i) {
      const float t = u[i] * inv_scale;
      const float p = exp_lean(t - m) * rs;
      const float w = minb + c1 * p;
      cum += (double)w;
      const float next = (i == K - 1) ? hi : (span * (float)cum + lo);
      const bool take = kSearch ? (v >= prev) : (i == idx);
      if (take) {
        found = i;
        klo = prev;
        bsz = next - prev;
      }
      prev = next;
    }
    idx = found;
    knot_lo = klo;
    bin_size = bsz;
  }
}

typedef float f2 __attribute__((ext_vector_type(2)));

// exp(x) for x <= 0 in the softmax: 2^(x*log2e) without the product-error compensation of exp_lean.
// The relative error grows as |x| * 6e-8, i.e. only on bins whose softmax weight is already small.
__device__ __forceinline__ float exp_softmax(float x) {
  return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

// Both cumulative axes in one pass, widths in .x and heights in .y so that the mul/add chain maps to
// packed v_pk_{mul,add}_f32 (2 lanes of work per VALU slot).  The bin is searched on one axis
// (kSearchX: widths, forward; else heights, inverse); knots are monotone, so "last bin whose lower
// knot <= v" is tracked by one predicate that selects on both axes.
template <int KS, bool kSearchX>
__device__ __forceinline__ void walk_both(const float* __restrict__ uw, const float* __restrict__ uh,
                                          float inv_scale, f2 minb, f2 c1, f2 lo, f2 hi, float v, int& idx,
                                          f2& knot_lo, f2& bin_size) {
  f2 t[KS > 0 ? KS : 1];
  float mx = -INFINITY, my = -INFINITY;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    t[i] = f2{uw[i], uh[i]} * inv_scale;
    mx = fmaxf(mx, t[i].x);
    my = fmaxf(my, t[i].y);
  }
  const f2 m = {mx, my};
  f2 sum = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const f2 d = t[i] - m;
    t[i] = f2{exp_softmax(d.x), exp_softmax(d.y)};
    sum += t[i];
  }
  const f2 rs = {div_lean(1.f, sum.x), div_lean(1.f, sum.y)};
  const f2 span = hi - lo;
  double cx = 0.0, cy = 0.0;  // at::cumsum on the CPU accumulates f32 in double
  f2 prev = lo, sel_lo = lo, sel_hi = lo;
  int found = 0;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const f2 p = t[i] * rs;
    const f2 w = minb + c1 * p;
    cx += (double)w.x;
    cy += (double)w.y;
    const f2 cum = {(float)cx, (float)cy};
    const f2 next = (i == KS - 1) ? hi : (span * cum + lo);
    const bool take = v >= (kSearchX ? prev.x : prev.y);
    sel_lo.x = take ? prev.x : sel_lo.x;
    sel_lo.y = take ? prev.y : sel_lo.y;
    sel_hi.x = take ? next.x : sel_hi.x;
    sel_hi.y = take ? next.y : sel_hi.y;
    found = take ? i : found;
    prev = next;
  }
  idx = found;
  knot_lo = sel_lo;
  bin_size = sel_hi - sel_lo;
}

template <int KS>
struct RQOp {
  static constexpr bool kHasPrepare = false;
  __device__ void prepare(float*, int, int) const {}
  RQParams q;
  float inv_div;  // unnormalised widths/heights are multiplied by 1/wh_div (exact for the usual
                  // power-of-two sqrt(hidden_features); otherwise within 1 ulp of the reference's division)

  __device__ __forceinline__ void eval(const float* __restrict__ prow, int j, int d_t, float x,
                                       float& y, float& lad, uint32_t& err) const {
    const int K = KS > 0 ? KS : q.K;
    const int P = q.tails ? 3 * K - 1 : 3 * K + 1;
    eval_core<false>(prow + j * P, x, y, lad, err);
  }

  // p -> the P raw values of one (sample, dim): an LDS pointer, or (kRegs) a register array that
  // must only be indexed statically, so the two derivatives are picked with a select chain.
  template <bool kRegs>
  __device__ __forceinline__ void eval_core(const float* __restrict__ p, float x, float& y, float& lad,
                                            uint32_t& err) const {
    const int K = KS > 0 ? KS : q.K;
#ifdef FC_PROBE_SKIP_EVAL  // tools/ ablation build only: data movement without the spline arithmetic
    y = x + p[0] * 0.f;
    lad = 0.f;
    return;
#endif

    // rational_quadratic.py:26-38 / :81-82
    const bool inside = (x >= q.left) && (x <= q.right);
    if (!inside) {
      y = x;
      lad = 0.f;
      if (!q.tails) err |= kErrOutsideDomain;
      return;
    }

    int idx = 0;
    float xk, wk, yk, hk;
    if constexpr (KS > 0) {
      const f2 minb = {q.min_w, q.min_h}, c1 = {q.cw, q.ch};
      const f2 lo = {q.left, q.bottom}, hi = {q.right, q.top};
      f2 klo, bsz;
      if (!q.inverse)
        walk_both<KS, true>(p, p + K, inv_div, minb, c1, lo, hi, x, idx, klo, bsz);
      else
        walk_both<KS, false>(p, p + K, inv_div, minb, c1, lo, hi, x, idx, klo, bsz);
      xk = klo.x; yk = klo.y; wk = bsz.x; hk = bsz.y;
    } else if (!q.inverse) {
      walk_axis<KS, true>(p, K, inv_div, q.min_w, q.cw, q.left, q.right, x,
                          idx, xk, wk);
      walk_axis<KS, false>(p + K, K, inv_div, q.min_h, q.ch, q.bottom, q.top,
                           x, idx, yk, hk);
    } else {
      walk_axis<KS, true>(p + K, K, inv_div, q.min_h, q.ch, q.bottom, q.top,
                          x, idx, yk, hk);
      walk_axis<KS, false>(p, K, inv_div, q.min_w, q.cw, q.left, q.right, x,
                           idx, xk, wk);
    }

    // derivatives at the two knots of the bin (rational_quadratic.py:33-36, :100-104)
    const float* ud = p + 2 * K;
    float u0, u1;
    if (kRegs && KS > 0) {
      // padded derivative row: linear tails [c, ud_0..ud_{K-2}, c]; none [ud_0..ud_K]
      u0 = q.tails ? q.tail_const : ud[0];
      u1 = q.tails ? q.tail_const : ud[KS];
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const float lo_i = q.tails ? (i == 0 ? q.tail_const : ud[i > 0 ? i - 1 : 0]) : ud[i];
        const float hi_i = q.tails ? (i == KS - 1 ? q.tail_const : ud[i < KS - 1 ? i : 0]) : ud[i + 1];
        u0 = idx == i ? lo_i : u0;
        u1 = idx == i ? hi_i : u1;
      }
    } else if (q.tails) {
      u0 = idx == 0 ? q.tail_const : ud[idx - 1];
      u1 = idx == K - 1 ? q.tail_const : ud[idx];
    } else {
      u0 = ud[idx];
      u1 = ud[idx + 1];
    }
    const float d0 = q.min_d + softplus_lean(u0, q.beta);
    const float d1 = q.min_d + softplus_lean(u1, q.beta);
    const float delta = div_lean(hk, wk);
    const float dsum = d0 + d1 - 2.f * delta;

    float theta;
    if (!q.inverse) {
      theta = div_lean(x - xk, wk);
    } else {
      // rational_quadratic.py:133-146
      const float r = x - yk;
      const float qa = r * dsum + hk * (delta - d0);
      const float qb = hk * d0 - r * dsum;
      const float qc = -delta * r;
      const float disc = qb * qb - 4.f * qa * qc;
      if (!(disc >= 0.f)) err |= kErrDiscriminant;
      theta = div_lean(2.f * qc, -qb - sqrt_lean(disc));
    }
    const float t1mt = theta * (1.f - theta);
    const float den = delta + dsum * t1mt;
    const float omt = 1.f - theta;
    const float dnum = (delta * delta) * (d1 * (theta * theta) + 2.f * delta * t1mt + d0 * (omt * omt));
    const float l = log_lean(dnum) - 2.f * log_lean(den);
    if (!q.inverse) {
      const float num = hk * (delta * (theta * theta) + d0 * t1mt);
      y = yk + div_lean(num, den);
      lad = l;
    } else {
      y = theta * wk + xk;
      lad = -l;
    }
  }
};

// ---- wave-independent variant: parameters straight into registers ----------------------------------
//
// One wavefront owns G = 64 / d_t consecutive samples: lane (s, j) loads the P raw values of its own
// (sample, dim) -- a contiguous P*4-byte chunk, the 64 chunks of a wave forming ONE contiguous span of
// the conditioner output -- directly into registers.  No LDS image of the parameters, no workgroup
// barrier: waves drift freely, so one wave's HBM wait is another wave's arithmetic.  Only the G input
// rows (G*D floats) pass through a per-wave LDS strip so that identity columns are copied and global
// x / y traffic stays coalesced.  Requires d_t a power of two <= 64 and a compile-time K.
template <int KS, bool kTails>
__global__ __launch_bounds__(256) void rq_wave_kernel(RQOp<KS> op, TileArgs a, int64_t groups) {
  constexpr int P = KS > 0 ? (kTails ? 3 * KS - 1 : 3 * KS + 1) : 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d_t = a.d_t, D = a.D;
  const int sh = __builtin_ctz(d_t);
  const int G = 64 >> sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int* cs = reinterpret_cast<int*>(smem);
  float* xw = smem + round4(d_t) + wave * round4(G * D);
  for (int j = threadIdx.x; j < d_t; j += blockDim.x) cs[j] = a.cols ? a.cols[j] : j;
  __syncthreads();
  const int s = lane >> sh, j = lane & (d_t - 1);
  const int col = cs[j];
  const int row_floats = G * D;
  uint32_t err = 0;
  const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6);
  // 16-byte loads through a 4-byte aligned vector type (the lane's chunk is P*4 bytes from its
  // neighbour's, so only dword aligned; global memory allows it): ~P/4 load instructions, not P
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  auto load_params = [&](float (&dst)[P], int64_t g) {
    const float* __restrict__ pp = a.params + (g * 64 + lane) * P;  // (n0 * d_t + lane) * P, n0 = g * G
#pragma unroll
    for (int i = 0; i + 4 <= P; i += 4) {
      const f4u v = *reinterpret_cast<const f4u*>(pp + i);
      dst[i] = v.x; dst[i + 1] = v.y; dst[i + 2] = v.z; dst[i + 3] = v.w;
    }
#pragma unroll
    for (int i = P & ~3; i < P; ++i) dst[i] = pp[i];
  };
  // (a register double-buffer prefetching the next group's parameters was measured: +31 VGPRs, no
  //  gain -- 4 resident waves per SIMD already cover the load latency; the kernel sits at ~80 % of the
  //  per-CU load-path rate)
  for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; g < groups; g += wstride) {
    const int64_t n0 = g * G;
    float p[P];
    load_params(p, g);
    const float* __restrict__ xg = a.x + n0 * D;
    if ((row_floats & 255) == 0) {
      for (int i = lane; i < (row_floats >> 2); i += 64)
        reinterpret_cast<float4*>(xw)[i] = reinterpret_cast<const float4*>(xg)[i];
    } else if ((row_floats & 127) == 0) {
      for (int i = lane; i < (row_floats >> 1); i += 64)
        reinterpret_cast<float2*>(xw)[i] = reinterpret_cast<const float2*>(xg)[i];
    } else {
      for (int i = lane; i < row_floats; i += 64) xw[i] = xg[i];
    }
    const float xv = xw[s * D + col];
    float yv, lad;
    op.template eval_core<true>(p, xv, yv, lad, err);
    xw[s * D + col] = yv;
    if (a.logabsdet) {
      const float tot = group_sum_rt(lad, d_t);
      if (j == 0) a.logabsdet[n0 + s] = (a.lad_mode & 2) ? -tot : tot;
    }
    float* __restrict__ yg = a.y + n0 * D;
    if ((row_floats & 255) == 0) {
      for (int i = lane; i < (row_floats >> 2); i += 64)
        reinterpret_cast<float4*>(yg)[i] = reinterpret_cast<const float4*>(xw)[i];
    } else if ((row_floats & 127) == 0) {
      for (int i = lane; i < (row_floats >> 1); i += 64)
        reinterpret_cast<float2*>(yg)[i] = reinterpret_cast<const float2*>(xw)[i];
    } else {
      for (int i = lane; i < row_floats; i += 64) yg[i] = xw[i];
    }
  }
  if (err && a.err) atomicOr(a.err, err);
}

template <int KS>
static hipError_t launch_rq(const RQParams& q, TileArgs a, hipStream_t stream) {
  RQOp<KS> op;
  op.q = q;
  op.inv_div = 1.f / q.wh_div;
  const bool pow2 = (a.d_t & (a.d_t - 1)) == 0 && a.d_t <= 64;
  const char* force = getenv("FC_RQ_PATH");  // "tile" / "wave": pin one structure (A/B measurements)
  if constexpr (KS > 0 && KS <= 10)
  if (pow2 && !a.shared_params && a.N > 0 && a.lad_mode != 1 && a.lad_mode != 3 &&
      !(force && force[0] == 't')) {
    const int G = 64 / a.d_t;
    const int64_t groups = a.N / G;
    const bool al = ((G * a.D) % 4 != 0) || (aligned16(a.x) && aligned16(a.y));
    if (groups >= 256 && al) {
      const size_t lds = sizeof(float) * (size_t)(round4(a.d_t) + 4 * round4(G * a.D));
      int64_t grid = (int64_t)device_cu_count() * 6;
      if (grid > (groups + 3) / 4) grid = (groups + 3) / 4;
      TileArgs body = a;
      if (q.tails)
        hipLaunchKernelGGL((rq_wave_kernel<KS, true>), dim3((unsigned)grid), dim3(256), lds, stream, op, body, groups);
      else
        hipLaunchKernelGGL((rq_wave_kernel<KS, false>), dim3((unsigned)grid), dim3(256), lds, stream, op, body, groups);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      const int64_t done = groups * G;
      if (done == a.N) return hipSuccess;
      const int P = q.tails ? 3 * KS - 1 : 3 * KS + 1;
      a.x += done * a.D;
      a.y += done * a.D;
      a.params += done * (int64_t)a.d_t * P;
      if (a.logabsdet) a.logabsdet += done;
      a.N -= done;
    }
  }
  return launch_tile(op, a, stream);
}

}  // namespace fc

extern "C" int fc_rq_spline(const float* x, float* y, const float* params, const int32_t* cols,
                            float* logabsdet, uint32_t* err_flag, int64_t n, int32_t d,
                            int32_t d_t, int32_t shared_params, int32_t lad_mode,
                            const fc_rq_config* cfg, void* stream) {
  if (!cfg || n < 0 || d <= 0 || d_t <= 0 || d_t > d || cfg->num_bins <= 0) return hipErrorInvalidValue;
  if (n > 0 && (!x || !y || !params)) return hipErrorInvalidValue;
  fc::RQParams q;
  q.K = cfg->num_bins;
  q.tails = cfg->tails;
  q.inverse = cfg->inverse;
  q.left = cfg->left; q.right = cfg->right; q.bottom = cfg->bottom; q.top = cfg->top;
  q.min_w = (float)cfg->min_bin_width;
  q.min_h = (float)cfg->min_bin_height;
  q.min_d = (float)cfg->min_derivative;
  q.cw = (float)(1.0 - cfg->min_bin_width * q.K);
  q.ch = (float)(1.0 - cfg->min_bin_height * q.K);
  q.wh_div = cfg->wh_divisor > 0.f ? cfg->wh_divisor : 1.f;
  q.beta = cfg->softplus_beta;
  q.tail_const = cfg->tail_constant;

  fc::TileArgs a{};
  a.x = x; a.y = y; a.params = params; a.cols = cols; a.logabsdet = logabsdet; a.err = err_flag;
  a.N = n; a.D = d; a.d_t = d_t;
  a.rowlen = d_t * (q.tails ? 3 * q.K - 1 : 3 * q.K + 1);
  a.shared_params = shared_params;
  a.lad_mode = lad_mode;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (q.K) {
    case 4: return fc::launch_rq<4>(q, a, s);
    case 5: return fc::launch_rq<5>(q, a, s);
    case 8: return fc::launch_rq<8>(q, a, s);
    case 10: return fc::launch_rq<10>(q, a, s);
    case 16: return fc::launch_rq<16>(q, a, s);
    default: return fc::launch_rq<0>(q, a, s);
  }
}
