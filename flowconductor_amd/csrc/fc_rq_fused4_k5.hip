// K = 5, linear tails: instance of the K-generic resident-weight fused kernel (fc_rq_fused4_body.h).
#define FC_F4_K 5
#define FC_F4_TAILS 1
#define FC_F4_NAME k5
#define FC_F4_EVAL_INC "fc_rq_fused4_eval_k5.inc"
#include "fc_rq_fused4_body.h"
