// Translation unit: lane-parallel BW6-761 final exponentiation kernel (pairing_lanes.h).
#include "pairing_lanes_kernels.h"
namespace celo { CELO_DEFINE_LANE_FE_LAUNCHER(LaneLaunch761, LP761) }
