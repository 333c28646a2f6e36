// Translation unit: BLS12-377 pairing kernels + engine (see pairing.h).
#include "pairing.h"
#include <mutex>

namespace celo {
static EnginePool<PairingEngine<PP377>>& pool_377() { static auto* p = new EnginePool<PairingEngine<PP377>>(); return *p; }
static std::mutex tm_mu_377;
static PairingTimings tm_last_377;

int pairing_run_377(const uint64_t* g1, const uint8_t* inf1, const uint64_t* g2, const uint8_t* inf2, const uint32_t* offsets, size_t m,
                    uint8_t* is_one, uint64_t* gt, int mode) {
  if (int rc = api_enter()) return rc;
  auto e = pool_377().lease();
  const int rc = e->run(g1, inf1, g2, inf2, offsets, m, is_one, gt, mode, e->own_stream());
  if (!rc && m) { std::lock_guard<std::mutex> lk(tm_mu_377); tm_last_377 = e->tm; }
  return rc;
}
int pairing_stage_377(uint32_t k, size_t m, PairingStage* st) {
  if (int rc = api_enter()) return rc;
  typedef EnginePool<PairingEngine<PP377>>::Lease L;
  L* l = new L(pool_377().lease());
  PairingEngine<PP377>::Staged sg;
  if ((*l)->stage(k, m, &sg)) { delete l; return 1; }
  *st = {l, sg.d_g1, sg.d_g2, sg.d_i1, sg.d_i2, (*l)->own_stream()};
  return 0;
}
int pairing_run_staged_377(PairingStage* st, const uint32_t* offsets, size_t m, uint8_t* is_one) {
  typedef EnginePool<PairingEngine<PP377>>::Lease L;
  L* l = (L*)st->lease;
  if (!l) return 2;
  int rc = 0;
  if (offsets) {
    rc = (*l)->run_staged(offsets, m, true, true, is_one, nullptr, 0, st->stream);
    if (!rc && m) { std::lock_guard<std::mutex> lk(tm_mu_377); tm_last_377 = (*l)->tm; }
  }
  delete l;
  st->lease = nullptr;
  return rc;
}
int pairing_timings_377(float ms[4]) {
  std::lock_guard<std::mutex> lk(tm_mu_377);
  ms[0] = tm_last_377.miller; ms[1] = tm_last_377.product; ms[2] = tm_last_377.final_exp; ms[3] = tm_last_377.total;
  return 0;
}
}  // namespace celo
