// Translation unit: lane-parallel BLS12-377 final exponentiation kernel (pairing_lanes.h).
#define CELO_LANES_DEFINE_FE 1
#include "pairing_lanes_kernels.h"
