// Translation unit: lane-parallel (six lanes per pairing) BLS12-377 final exponentiation kernel (pairing_lanes.h).
#include "pairing_lanes_kernels.h"
namespace celo { CELO_DEFINE_SLOT_FE_LAUNCHER(LaneLaunch377, LPH377) }
