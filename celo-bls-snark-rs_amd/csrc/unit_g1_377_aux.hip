#include "msm_unit.h"
CELO_DEFINE_MSM_AUX_UNIT(celo::G1_377, g1_377)
