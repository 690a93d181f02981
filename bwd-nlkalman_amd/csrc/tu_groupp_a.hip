// tu_groupp_a.hip — k_groupp for patch sizes 2 3 4 5 6 7 8
#include "groupp_launch.h"

int nlk_launch_groupp_a(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                         float* acc, const uint8_t* active) {
  switch (g.psz) {
    case 2: return nlk_groupp_launch_t<2>(c, g, img, cur, prev, acc, active);
    case 3: return nlk_groupp_launch_t<3>(c, g, img, cur, prev, acc, active);
    case 4: return nlk_groupp_launch_t<4>(c, g, img, cur, prev, acc, active);
    case 5: return nlk_groupp_launch_t<5>(c, g, img, cur, prev, acc, active);
    case 6: return nlk_groupp_launch_t<6>(c, g, img, cur, prev, acc, active);
    case 7: return nlk_groupp_launch_t<7>(c, g, img, cur, prev, acc, active);
    case 8: return nlk_groupp_launch_t<8>(c, g, img, cur, prev, acc, active);
  }
  return fail(c, NLK_EUNSUP, "patch size %d not supported (2 .. 16)", g.psz);
}
