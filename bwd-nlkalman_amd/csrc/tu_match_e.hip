// tu_match_e.hip — block-matching kernels for patch sizes 16 (match_launch.h)
#include "match_launch.h"
NLK_MATCH_PSZ(16)
