// tu_match_a.hip — block-matching kernels for patch sizes 4, 6 (match_launch.h)
#include "match_launch.h"
NLK_MATCH_PSZ(4)
NLK_MATCH_PSZ(6)
