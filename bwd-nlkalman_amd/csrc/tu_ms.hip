// tu_ms.hip — the multiscale entry points (ms_host.h) and their kernels (k_ms.h)
#include <math.h>

#include "k_ms.h"
#include "nlk_internal.h"
#include "ms_host.h"
