// tu_match_b.hip — block-matching kernels for patch sizes 8 (match_launch.h)
#include "match_launch.h"
NLK_MATCH_PSZ(8)
