// Translation unit: NTT over Fr(BW6-761) (see ntt.h).
#include "ntt.h"
#include <mutex>

namespace celo {
std::mutex& api_mutex();
int api_ensure_init();
static NttEngine eng_ntt;

int ntt_run(uint64_t* data, unsigned log_n, const uint64_t* omega, const uint64_t* coset, int coset_after, const uint64_t* scale, int dev, void* stream) {
  std::lock_guard<std::mutex> lk(api_mutex());
  if (int rc = api_ensure_init()) return rc;
  if (!data || !omega) return 2;
  return dev ? eng_ntt.run_device(data, log_n, omega, coset, coset_after, scale, (hipStream_t)stream)
             : eng_ntt.run_host(data, log_n, omega, coset, coset_after, scale, nullptr);
}
int ntt_timings(float ms[4], int* passes) {
  std::lock_guard<std::mutex> lk(api_mutex());
  ms[0] = eng_ntt.tm.load; ms[1] = eng_ntt.tm.passes; ms[2] = eng_ntt.tm.store; ms[3] = eng_ntt.tm.total;
  if (passes) *passes = eng_ntt.tm.npasses;
  return 0;
}
}  // namespace celo
