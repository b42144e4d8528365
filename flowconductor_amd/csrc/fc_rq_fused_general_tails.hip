// Instantiations of the general fused final-Linear + RQ-spline kernel with linear tails (fc_rq_fused_general.h).
#include "fc_rq_fused_general.h"

namespace fc {

hipError_t launch_general_tails(int K, const RQParams& q, const GenArgs& a, hipStream_t stream) {
  switch (K) {
    case 4: return launch_general<4, true>(q, a, stream);
    case 5: return launch_general<5, true>(q, a, stream);
    case 6: return launch_general<6, true>(q, a, stream);
    case 7: return launch_general<7, true>(q, a, stream);
    case 8: return launch_general<8, true>(q, a, stream);
    case 9: return launch_general<9, true>(q, a, stream);
    case 10: return launch_general<10, true>(q, a, stream);
    case 11: return launch_general<11, true>(q, a, stream);
    case 12: return launch_general<12, true>(q, a, stream);
    case 13: return launch_general<13, true>(q, a, stream);
    case 14: return launch_general<14, true>(q, a, stream);
    case 15: return launch_general<15, true>(q, a, stream);
    case 16: return launch_general<16, true>(q, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace fc
