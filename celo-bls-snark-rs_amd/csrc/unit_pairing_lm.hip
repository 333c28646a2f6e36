// Translation unit: lane-parallel (six lanes per pairing) BLS12-377 Miller loop and GT product kernels (pairing_lanes.h).
#include "pairing_lanes_kernels.h"
namespace celo { CELO_DEFINE_SLOT_MILLER_LAUNCHERS(LaneLaunch377, LPH377) }
