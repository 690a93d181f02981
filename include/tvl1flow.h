/* tvl1flow.h — drop-in declaration of the reference's multiscale dual TV-L1 optical flow
 * (reference: lib/tvl1flow/tvl1flow_lib.c:345-361, the function lib/tvl1flow/main.c:170 calls).
 *
 * Same name, argument list and meaning; host pointers in and out. The reference's tool
 * includes the implementation file (`#include "tvl1flow_lib.c"`, main.c:22); a maintainer
 * replaces that line by `#include "tvl1flow.h"` and links libnlkalman.so + libnlk_hip.so.
 * The arithmetic runs in the HIP kernels behind nlk_dev_tvl1_flow (include/nlk_hip.h); there is
 * no CPU fallback: without a usable GPU the call prints a message and exits, like the
 * reference does on allocation failure (lib/tvl1flow/xmalloc.c:15-21). */
#ifndef NLK_TVL1FLOW_H
#define NLK_TVL1FLOW_H

#include <stdbool.h>

void Dual_TVL1_optic_flow_multiscale(
    float *I0,            /* source image (nxx * nyy) */
    float *I1,            /* target image */
    float *u1,            /* out: x component of the flow */
    float *u2,            /* out: y component of the flow */
    const int nxx,        /* image width */
    const int nyy,        /* image height */
    const float tau,      /* time step */
    const float lambda,   /* weight of the data term */
    const float theta,    /* weight of (u - v)^2 */
    const int nscales,    /* number of scales */
    const int fscale,     /* finest scale that is solved */
    const float zfactor,  /* pyramid factor */
    const int warps,      /* warps per scale */
    const float epsilon,  /* stopping tolerance */
    const bool verbose);  /* (no per-warp messages are printed here) */

#endif
