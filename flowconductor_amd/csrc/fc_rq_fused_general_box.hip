// Instantiations of the general fused final-Linear + RQ-spline kernel without tails (tails=None: the spline lives on
// [left, right] x [bottom, top], inputs outside raise InputOutsideDomain) -- fc_rq_fused_general.h.
#include "fc_rq_fused_general.h"

namespace fc {

hipError_t launch_general_box(int K, const RQParams& q, const GenArgs& a, hipStream_t stream) {
  switch (K) {
    case 4: return launch_general<4, false>(q, a, stream);
    case 5: return launch_general<5, false>(q, a, stream);
    case 6: return launch_general<6, false>(q, a, stream);
    case 7: return launch_general<7, false>(q, a, stream);
    case 8: return launch_general<8, false>(q, a, stream);
    case 9: return launch_general<9, false>(q, a, stream);
    case 10: return launch_general<10, false>(q, a, stream);
    case 11: return launch_general<11, false>(q, a, stream);
    case 12: return launch_general<12, false>(q, a, stream);
    case 13: return launch_general<13, false>(q, a, stream);
    case 14: return launch_general<14, false>(q, a, stream);
    case 15: return launch_general<15, false>(q, a, stream);
    case 16: return launch_general<16, false>(q, a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace fc
