#include "msm_unit.h"
CELO_DEFINE_MSM_AUX_UNIT(celo::G_761, 761)
