// Instantiations of the one-launch, one-wave-per-SIMD backward of the fused final-Linear + RQ-spline layer
// (fc_rq_fused_backward512.h): every bin count the two-launch roles cover (3K -/+ 1 <= 32).
#include "fc_rq_fused_backward512.h"

namespace fc {

hipError_t launch_backward512_any(int K, bool tails, const RQParams& q, const BwdArgs& a, hipStream_t stream) {
#ifdef FC_B5_ONLY_K8      // probe builds: one instance
  return (tails && K == 8) ? launch_backward512<8, true>(q, a, stream) : hipErrorInvalidValue;
#else
  if (tails) {
    switch (K) {
      case 4: return launch_backward512<4, true>(q, a, stream);
      case 5: return launch_backward512<5, true>(q, a, stream);
      case 6: return launch_backward512<6, true>(q, a, stream);
      case 7: return launch_backward512<7, true>(q, a, stream);
      case 8: return launch_backward512<8, true>(q, a, stream);
      case 9: return launch_backward512<9, true>(q, a, stream);
      case 10: return launch_backward512<10, true>(q, a, stream);
      case 11: return launch_backward512<11, true>(q, a, stream);
      default: return hipErrorInvalidValue;
    }
  }
  switch (K) {
    case 4: return launch_backward512<4, false>(q, a, stream);
    case 5: return launch_backward512<5, false>(q, a, stream);
    case 6: return launch_backward512<6, false>(q, a, stream);
    case 7: return launch_backward512<7, false>(q, a, stream);
    case 8: return launch_backward512<8, false>(q, a, stream);
    case 9: return launch_backward512<9, false>(q, a, stream);
    case 10: return launch_backward512<10, false>(q, a, stream);
    default: return hipErrorInvalidValue;
  }
#endif
}

}  // namespace fc
