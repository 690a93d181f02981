// tu_match_d.hip — block-matching kernels for patch sizes 12 (match_launch.h)
#include "match_launch.h"
NLK_MATCH_PSZ(12)
