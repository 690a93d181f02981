// tu_tvl1.hip — the TV-L1 optical-flow entry points (tvl1_host.h) and their kernels (k_tvl1.h)
#include <math.h>

#include "k_tvl1.h"
#include "nlk_internal.h"
#include "tvl1_host.h"
