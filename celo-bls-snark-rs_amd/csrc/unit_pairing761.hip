// Translation unit: BW6-761 pairing kernels + engine (see pairing.h).
#include "pairing.h"
#include <mutex>

namespace celo {
static EnginePool<PairingEngine<PP761>>& pool_761() { static auto* p = new EnginePool<PairingEngine<PP761>>(); return *p; }
static std::mutex tm_mu_761;
static PairingTimings tm_last_761;

int pairing_run_761(const uint64_t* g1, const uint8_t* inf1, const uint64_t* g2, const uint8_t* inf2, const uint32_t* offsets, size_t m,
                    uint8_t* is_one, uint64_t* gt, int mode) {
  if (int rc = api_enter()) return rc;
  auto e = pool_761().lease();
  const int rc = e->run(g1, inf1, g2, inf2, offsets, m, is_one, gt, mode, e->own_stream());
  if (!rc && m) { std::lock_guard<std::mutex> lk(tm_mu_761); tm_last_761 = e->tm; }
  return rc;
}
}  // namespace celo
