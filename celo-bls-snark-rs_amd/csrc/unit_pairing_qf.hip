// Translation unit: lane-parallel BLS12-377 final exponentiation kernel (pairing_quad.h).
#define CELO_QUAD_DEFINE_FE 1
#include "pairing_quad_kernels.h"
