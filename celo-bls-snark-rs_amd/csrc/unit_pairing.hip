// Translation unit: BLS12-377 pairing kernels + engine (see pairing.h).
#include "pairing.h"
#include <mutex>

namespace celo {
std::mutex& api_mutex();
int api_ensure_init();
static PairingEngine<PP377> eng_pairing;

int pairing_run_377(const uint64_t* g1, const uint8_t* inf1, const uint64_t* g2, const uint8_t* inf2, const uint32_t* offsets, size_t m,
                    uint8_t* is_one, uint64_t* gt, int mode) {
  std::lock_guard<std::mutex> lk(api_mutex());
  if (int rc = api_ensure_init()) return rc;
  return eng_pairing.run(g1, inf1, g2, inf2, offsets, m, is_one, gt, mode, nullptr);
}
int pairing_timings_377(float ms[4]) {
  std::lock_guard<std::mutex> lk(api_mutex());
  ms[0] = eng_pairing.tm.miller; ms[1] = eng_pairing.tm.product; ms[2] = eng_pairing.tm.final_exp; ms[3] = eng_pairing.tm.total;
  return 0;
}
}  // namespace celo
