// One instantiation of the library's k_accumulate<G1_377> (csrc/msm.h) under the namespace given on the command line; build.sh compiles this file
// twice - with the SLP vectorizer (namespace celo: the flags of csrc/Makefile's SLP_UNITS) and without it (namespace celo_n, -fno-slp-vectorize) -
// and links both into one program, so that the SAME bucket runs go through both kernels in one process.
#include "../../celo-bls-snark-rs_amd/csrc/msm.h"
#ifndef VARIANT
#error "VARIANT"
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
extern "C" void CAT(launch_acc_, VARIANT)(const uint32_t* bases, const uint32_t* sorted, const uint32_t* pstart, const uint32_t* plen, const uint32_t* order,
                                          const uint32_t* nwork, uint32_t* partials, uint32_t slots) {
  hipLaunchKernelGGL((celo::k_accumulate<celo::G1_377>), dim3((slots + 255) / 256), dim3(256), 0, 0, bases, sorted, pstart, plen, order, nwork, partials);
}
