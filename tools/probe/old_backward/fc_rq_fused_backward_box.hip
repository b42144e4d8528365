// Instantiations of the fused final-Linear + RQ-spline backward kernels (no tails; fc_rq_fused_backward.h).
#include "fc_rq_fused_backward.h"

namespace fc {

hipError_t launch_backward_box(int K, int role, const RQParams& q, const BwdArgs& a, hipStream_t stream) {
  switch (K) {
    case 4: return launch_backward<4, false>(role, q, a, stream);
    case 5: return launch_backward<5, false>(role, q, a, stream);
    case 6: return launch_backward<6, false>(role, q, a, stream);
    case 7: return launch_backward<7, false>(role, q, a, stream);
    case 8: return launch_backward<8, false>(role, q, a, stream);
    case 9: return launch_backward<9, false>(role, q, a, stream);
    case 10: return launch_backward<10, false>(role, q, a, stream);
    case 11: return launch_backward<11, false>(role, q, a, stream);
    case 12: return launch_backward<12, false>(role, q, a, stream);
    case 13: return launch_backward<13, false>(role, q, a, stream);
    case 14: return launch_backward<14, false>(role, q, a, stream);
    case 15: return launch_backward<15, false>(role, q, a, stream);
    case 16: return launch_backward<16, false>(role, q, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace fc
