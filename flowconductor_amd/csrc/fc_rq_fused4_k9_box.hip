// K = 9, no tails (coupling.py:543-547): instance of the K-generic resident-weight fused kernel (fc_rq_fused4_body.h).
#define FC_F4_K 9
#define FC_F4_TAILS 0
#define FC_F4_NAME k9_box
#define FC_F4_EVAL_INC "fc_rq_fused4_eval_k9_box.inc"
#include "fc_rq_fused4_body.h"
