// K = 4, no tails (coupling.py:543-547): instance of the K-generic resident-weight fused kernel (fc_rq_fused4_body.h).
#define FC_F4_K 4
#define FC_F4_TAILS 0
#define FC_F4_NAME k4_box
#define FC_F4_EVAL_INC "fc_rq_fused4_eval_k4_box.inc"
#include "fc_rq_fused4_body.h"
