// tu_match_c.hip — block-matching kernels for patch sizes 10 (match_launch.h)
#include "match_launch.h"
NLK_MATCH_PSZ(10)
