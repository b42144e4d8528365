// Instantiations of the fused final-Linear + RQ-spline backward kernels (linear tails; fc_rq_fused_backward.h).
#include "fc_rq_fused_backward.h"

namespace fc {

hipError_t launch_backward_tails(int K, int role, const RQParams& q, const BwdArgs& a, hipStream_t stream) {
  switch (K) {
    case 4: return launch_backward<4, true>(role, q, a, stream);
    case 5: return launch_backward<5, true>(role, q, a, stream);
    case 6: return launch_backward<6, true>(role, q, a, stream);
    case 7: return launch_backward<7, true>(role, q, a, stream);
    case 8: return launch_backward<8, true>(role, q, a, stream);
    case 9: return launch_backward<9, true>(role, q, a, stream);
    case 10: return launch_backward<10, true>(role, q, a, stream);
    case 11: return launch_backward<11, true>(role, q, a, stream);
    case 12: return launch_backward<12, true>(role, q, a, stream);
    case 13: return launch_backward<13, true>(role, q, a, stream);
    case 14: return launch_backward<14, true>(role, q, a, stream);
    case 15: return launch_backward<15, true>(role, q, a, stream);
    case 16: return launch_backward<16, true>(role, q, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace fc
