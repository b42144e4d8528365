// Dispatch of the K-generic resident-weight fused final-Linear + RQ-spline kernel (fc_rq_fused4_body.h; one translation
// unit per bin count).  Called from the general kernel's launcher (fc_rq_fused_general.h) for the shapes it covers.
#include "fc_rq_fused_general.h"

namespace fc {

hipError_t launch_fused4_k10(const RQParams& q, const GenArgs& a, hipStream_t stream);
size_t fused4_lds_bytes_k10(int d);

// hidden 64, linear tails, a bin count with an instance, and the LDS image of the tile fits
bool fused4_takes(const RQParams& q, const GenArgs& a) {
  if (a.H != 64 || !q.tails) return false;
  switch (q.K) {
    case 10: return fused4_lds_bytes_k10(a.D) <= 160 * 1024;
    default: return false;
  }
}

hipError_t launch_fused4(const RQParams& q, const GenArgs& a, hipStream_t stream) {
  switch (q.K) {
    case 10: return launch_fused4_k10(q, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace fc
