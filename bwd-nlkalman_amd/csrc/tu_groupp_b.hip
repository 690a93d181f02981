// tu_groupp_b.hip — k_groupp for patch sizes 9 10 11 12
#include "groupp_launch.h"

int nlk_launch_groupp_b(nlk_ctx* c, const NlkGeom& g, const float* img, const float* cur, const float* prev,
                         float* acc, const uint8_t* active) {
  switch (g.psz) {
    case 9: return nlk_groupp_launch_t<9>(c, g, img, cur, prev, acc, active);
    case 10: return nlk_groupp_launch_t<10>(c, g, img, cur, prev, acc, active);
    case 11: return nlk_groupp_launch_t<11>(c, g, img, cur, prev, acc, active);
    case 12: return nlk_groupp_launch_t<12>(c, g, img, cur, prev, acc, active);
  }
  return fail(c, NLK_EUNSUP, "patch size %d not supported (2 .. 16)", g.psz);
}
