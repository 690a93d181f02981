// tu_match_f.hip — block-matching kernels for 8 x 8 patches in the opt-in block-summed distance order
// (NLK_MATCH_ORDER=block; k_match.h: nlk_match_block_sum)
#include "match_launch.h"
NLK_MATCH_PSZ_BS(8)
