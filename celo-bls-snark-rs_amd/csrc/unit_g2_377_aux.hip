#include "msm_unit.h"
CELO_DEFINE_MSM_AUX_UNIT(celo::G2_377, g2_377)
