// Dispatch of the K-generic resident-weight fused final-Linear + RQ-spline kernel (fc_rq_fused4_body.h; one translation
// unit per bin count and tail mode: linear tails K = 4..7 and 9..11 -- K = 8 has its own kernel, fc_rq_fused3.hip --, no
// tails K = 4..10; beyond that the weights of 4 dims no longer fit a wave's registers).  Called from the general entry
// (fc_rq_fused_general.hip) for these shapes.
#include "fc_rq_fused_general.h"

namespace fc {

hipError_t launch_fused4_k4(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k4(int d);
hipError_t launch_fused4_k5(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k5(int d);
hipError_t launch_fused4_k6(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k6(int d);
hipError_t launch_fused4_k7(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k7(int d);
hipError_t launch_fused4_k9(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k9(int d);
hipError_t launch_fused4_k10(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k10(int d);
hipError_t launch_fused4_k11(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k11(int d);
hipError_t launch_fused4_k4_box(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k4_box(int d);
hipError_t launch_fused4_k5_box(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k5_box(int d);
hipError_t launch_fused4_k6_box(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k6_box(int d);
hipError_t launch_fused4_k7_box(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k7_box(int d);
hipError_t launch_fused4_k8_box(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k8_box(int d);
hipError_t launch_fused4_k9_box(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k9_box(int d);
hipError_t launch_fused4_k10_box(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k10_box(int d);

// hidden 64, a (bin count, tail mode) with an instance, and the LDS image of the tile fits
bool fused4_takes(const RQParams& q, const GenArgs& a) {
  if (a.H != 64) return false;
  if (q.tails) {
    switch (q.K) {
      case 4: return fused4_lds_bytes_k4(a.D) <= 160 * 1024;
      case 5: return fused4_lds_bytes_k5(a.D) <= 160 * 1024;
      case 6: return fused4_lds_bytes_k6(a.D) <= 160 * 1024;
      case 7: return fused4_lds_bytes_k7(a.D) <= 160 * 1024;
      case 9: return fused4_lds_bytes_k9(a.D) <= 160 * 1024;
      case 10: return fused4_lds_bytes_k10(a.D) <= 160 * 1024;
      case 11: return fused4_lds_bytes_k11(a.D) <= 160 * 1024;
      default: return false;
    }
  }
  switch (q.K) {
    case 4: return fused4_lds_bytes_k4_box(a.D) <= 160 * 1024;
    case 5: return fused4_lds_bytes_k5_box(a.D) <= 160 * 1024;
    case 6: return fused4_lds_bytes_k6_box(a.D) <= 160 * 1024;
    case 7: return fused4_lds_bytes_k7_box(a.D) <= 160 * 1024;
    case 8: return fused4_lds_bytes_k8_box(a.D) <= 160 * 1024;
    case 9: return fused4_lds_bytes_k9_box(a.D) <= 160 * 1024;
    case 10: return fused4_lds_bytes_k10_box(a.D) <= 160 * 1024;
    default: return false;
  }
}

hipError_t launch_fused4(const RQParams& q, const GenArgs& a, hipStream_t stream) {
  if (q.tails) {
    switch (q.K) {
      case 4: return launch_fused4_k4(q, a, stream);
      case 5: return launch_fused4_k5(q, a, stream);
      case 6: return launch_fused4_k6(q, a, stream);
      case 7: return launch_fused4_k7(q, a, stream);
      case 9: return launch_fused4_k9(q, a, stream);
      case 10: return launch_fused4_k10(q, a, stream);
      case 11: return launch_fused4_k11(q, a, stream);
      default: return hipErrorInvalidValue;
    }
  }
  switch (q.K) {
    case 4: return launch_fused4_k4_box(q, a, stream);
    case 5: return launch_fused4_k5_box(q, a, stream);
    case 6: return launch_fused4_k6_box(q, a, stream);
    case 7: return launch_fused4_k7_box(q, a, stream);
    case 8: return launch_fused4_k8_box(q, a, stream);
    case 9: return launch_fused4_k9_box(q, a, stream);
    case 10: return launch_fused4_k10_box(q, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace fc
