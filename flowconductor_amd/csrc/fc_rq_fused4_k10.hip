// K = 10 (the reference's default num_bins, coupling.py:507) instance of the K-generic resident-weight fused kernel.
#define FC_F4_K 10
#define FC_F4_EVAL_INC "fc_rq_fused4_eval_k10.inc"
#include "fc_rq_fused4_body.h"
