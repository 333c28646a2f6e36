// Translation unit: BW6-761 pairing kernels + engine (see pairing.h).
#include "pairing.h"
#include <mutex>

namespace celo {
std::mutex& api_mutex();
int api_ensure_init();
static PairingEngine<PP761> eng_pairing761;

int pairing_run_761(const uint64_t* g1, const uint8_t* inf1, const uint64_t* g2, const uint8_t* inf2, const uint32_t* offsets, size_t m,
                    uint8_t* is_one, uint64_t* gt, int mode) {
  std::lock_guard<std::mutex> lk(api_mutex());
  if (int rc = api_ensure_init()) return rc;
  return eng_pairing761.run(g1, inf1, g2, inf2, offsets, m, is_one, gt, mode, nullptr);
}
}  // namespace celo
