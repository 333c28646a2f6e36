// Translation unit: lane-parallel BLS12-377 Miller loop and GT product kernels (pairing_lanes.h).
#define CELO_LANES_DEFINE_MILLER 1
#include "pairing_lanes_kernels.h"
