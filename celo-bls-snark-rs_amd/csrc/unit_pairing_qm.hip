// Translation unit: lane-parallel BLS12-377 Miller loop and GT product kernels (pairing_quad.h).
#define CELO_QUAD_DEFINE_MILLER 1
#include "pairing_quad_kernels.h"
