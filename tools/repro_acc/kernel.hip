// One instantiation of the library's k_accumulate<G2_377> (csrc/msm.h) under the namespace and the signed-pass site mask given on the
// command line; tools/repro_acc/build.sh compiles this file twice (unsigned: namespace celo; signed xyzz_madd: namespace celo_s) and
// links both into one program, so that the SAME inputs run through both kernels in one process.
#include "../../celo-bls-snark-rs_amd/csrc/msm.h"
#ifndef VARIANT
#error "VARIANT"
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
extern "C" void CAT(launch_acc_, VARIANT)(const uint32_t* bases, const uint32_t* sorted, const uint32_t* pstart, const uint32_t* plen, const uint32_t* order,
                                          const uint32_t* nwork, uint32_t* partials, uint32_t slots) {
  hipLaunchKernelGGL((celo::k_accumulate<celo::G2_377>), dim3((slots + 255) / 256), dim3(256), 0, 0, bases, sorted, pstart, plen, order, nwork, partials);
}
