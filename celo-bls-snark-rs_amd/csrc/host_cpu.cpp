// CPU feature probe for the optional host paths (compiled WITHOUT any -mavx512* flag, unlike host_ifma.cpp, so that it runs
// anywhere): msm.h asks it once whether the IFMA Horner epilogue may be used.  CELO_NO_IFMA=1 forces the 64-bit-limb path (A/B).
#include <cstdlib>
extern "C" int celo_ifma_available() {
  static const int ok = [] {
    if (getenv("CELO_NO_IFMA")) return 0;
    __builtin_cpu_init();
    return (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq") &&
            __builtin_cpu_supports("avx512vl")) ? 1 : 0;
  }();
  return ok;
}
