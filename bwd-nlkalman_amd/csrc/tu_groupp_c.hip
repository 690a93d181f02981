// tu_groupp_c.hip — k_groupp for patch sizes 13 14 15 16
#include "groupp_launch.h"

int nlk_launch_groupp_c(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                         float* acc, const uint8_t* active) {
  switch (g.psz) {
    case 13: return nlk_groupp_launch_t<13>(c, g, img, cur, prev, acc, active);
    case 14: return nlk_groupp_launch_t<14>(c, g, img, cur, prev, acc, active);
    case 15: return nlk_groupp_launch_t<15>(c, g, img, cur, prev, acc, active);
    case 16: return nlk_groupp_launch_t<16>(c, g, img, cur, prev, acc, active);
  }
  return fail(c, NLK_EUNSUP, "patch size %d not supported (2 .. 16)", g.psz);
}
