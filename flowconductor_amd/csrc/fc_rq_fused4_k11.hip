// K = 11, linear tails: instance of the K-generic resident-weight fused kernel (fc_rq_fused4_body.h).
#define FC_F4_K 11
#define FC_F4_TAILS 1
#define FC_F4_NAME k11
#define FC_F4_EVAL_INC "fc_rq_fused4_eval_k11.inc"
#include "fc_rq_fused4_body.h"
