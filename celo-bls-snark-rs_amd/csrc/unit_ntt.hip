// Translation unit: NTTs over Fr(BW6-761) and Fr(BLS12-377) (see ntt.h).
#include "ntt.h"
#include <mutex>

namespace celo {
template <class FR> struct NttUnit {
  static EnginePool<NttEngine<FR>>& pool() { static auto* p = new EnginePool<NttEngine<FR>>(); return *p; }
};
static std::mutex tm_mu_ntt;
static NttTimings tm_last_ntt;     // last transform of either field

template <class FR>
static int ntt_run_t(uint64_t* data, unsigned log_n, const uint64_t* omega, const uint64_t* coset, int coset_after, const uint64_t* scale, int dev, void* stream) {
  if (int rc = api_enter()) return rc;
  if (!data || !omega) return 2;
  auto e = NttUnit<FR>::pool().lease();      // an engine keeps the twiddle table of its last (omega, n): repeated transforms reuse it
  const int rc = dev ? e->run_device(data, log_n, omega, coset, coset_after, scale, (hipStream_t)stream)
                     : e->run_host(data, log_n, omega, coset, coset_after, scale, e->own_stream());
  if (!rc) { std::lock_guard<std::mutex> lk(tm_mu_ntt); tm_last_ntt = e->tm; }
  return rc;
}
int ntt_run(uint64_t* data, unsigned log_n, const uint64_t* omega, const uint64_t* coset, int coset_after, const uint64_t* scale, int dev, void* stream) {
  return ntt_run_t<Fr761>(data, log_n, omega, coset, coset_after, scale, dev, stream);
}
int ntt_run_253(uint64_t* data, unsigned log_n, const uint64_t* omega, const uint64_t* coset, int coset_after, const uint64_t* scale, int dev, void* stream) {
  return ntt_run_t<Fr377>(data, log_n, omega, coset, coset_after, scale, dev, stream);
}
int ntt_timings(float ms[4], int* passes) {
  std::lock_guard<std::mutex> lk(tm_mu_ntt);
  ms[0] = tm_last_ntt.load; ms[1] = tm_last_ntt.passes; ms[2] = tm_last_ntt.store; ms[3] = tm_last_ntt.total;
  if (passes) *passes = tm_last_ntt.npasses;
  return 0;
}
}  // namespace celo
