// One translation unit per group (parallel builds; each unit instantiates the MSM pipeline for one group).
#pragma once
#include "msm.h"
#include <mutex>

namespace celo {
std::mutex& api_mutex();
int api_ensure_init();

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// P_i = k_i * G, k_i = splitmix64(seed, i) | 1, affine output in arkworks layout
template <class F>
__global__ void __launch_bounds__(128) k_gen_points(uint64_t* __restrict__ out, size_t n, uint64_t seed, const uint32_t* __restrict__ gen_dev) {
  typedef PointIO<F> IO;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> g = IO::load_affine(gen_dev);
  uint64_t k = splitmix64_at(seed, i) | 1ULL;
  Xyzz<F> acc = Xyzz<F>::identity();
  for (int b = 63; b >= 0; b--) {
    acc = xyzz_dbl(acc);
    if ((k >> b) & 1) xyzz_madd(acc, g);
  }
  F t = F::inv(F::mul(acc.ZZ, acc.ZZZ));  // x = X/ZZ, y = Y/ZZZ with one inversion
  F x = F::mul(acc.X, F::mul(t, acc.ZZZ));
  F y = F::mul(acc.Y, F::mul(t, acc.ZZ));
  uint64_t* o = out + i * 2 * IO::ARK64;
  x.to_ark(o);
  y.to_ark(o + IO::ARK64);
}

template <class F> int gen_points_impl(void* d_out, size_t n, uint64_t seed, const uint64_t* gen_xy, void* stream) {
  typedef PointIO<F> IO;
  std::lock_guard<std::mutex> lk(api_mutex());
  if (int rc = api_ensure_init()) return rc;
  Affine<F> g = {F::from_ark(gen_xy), F::from_ark(gen_xy + IO::ARK64)};
  uint32_t h[IO::AFF_WORDS];
  IO::store_affine(h, g);
  uint32_t* d_gen = nullptr;
  HIP_OK(hipMalloc(&d_gen, sizeof h));
  HIP_OK(hipMemcpy(d_gen, h, sizeof h, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_gen_points<F>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, (uint64_t*)d_out, n, seed, d_gen);
  HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  HIP_OK(hipGetLastError());
  HIP_OK(hipFree(d_gen));
  return 0;
}

// Host-side plain sum of k Jacobian points (arkworks layout): the multi-GPU fold of per-rank partial MSM results
// and small aggregates (Signature::aggregate / PublicKey::aggregate, crates/bls-crypto/src/bls/signature.rs:61-67,
// public.rs:38-44).  Large aggregates go through the MSM kernels with unit scalars.
template <class F> int sum_jacobian_impl(const uint64_t* jac, size_t k, uint64_t* out) {
  constexpr int A = F::ARK64;
  Xyzz<F> acc = Xyzz<F>::identity();
  for (size_t i = 0; i < k; i++) {
    const uint64_t* p = jac + i * 3 * A;
    F Z = F::from_ark(p + 2 * A);
    if (Z.is_zero_mod_p()) continue;
    F ZZ = F::sqr(Z);
    Xyzz<F> v = {F::norm(F::from_ark(p)), F::norm(F::from_ark(p + A)), ZZ, F::mul(ZZ, Z)};
    xyzz_add(acc, v);
  }
  if (acc.is_identity() || acc.ZZ.is_zero_mod_p()) {
    F::zero().to_ark(out); F::one().to_ark(out + A); F::zero().to_ark(out + 2 * A);
    return 0;
  }
  F::mul(acc.X, acc.ZZ).to_ark(out);
  F::mul(acc.Y, acc.ZZZ).to_ark(out + A);
  acc.ZZ.to_ark(out + 2 * A);
  return 0;
}
}  // namespace celo

// The large-MSM engine and the auxiliary entry points (batched MSMs, generators, host sums) are separate macros so that
// each group compiles as two translation units in parallel (the 28-limb instantiations are the long pole of the build).
#define CELO_DEFINE_MSM_UNIT(G, TAG)                                                                                     \
  namespace celo {                                                                                                       \
  static MsmEngine<G> eng_##TAG;                                                                                         \
  int msm_host_##TAG(const uint64_t* b, const uint8_t* inf, const uint64_t* s, size_t n, uint64_t* out) {                \
    std::lock_guard<std::mutex> lk(api_mutex());                                                                         \
    if (int rc = api_ensure_init()) return rc;                                                                           \
    return eng_##TAG.run_host(b, inf, s, n, out, nullptr);                                                               \
  }                                                                                                                      \
  int msm_dev_##TAG(const void* b, const void* inf, const void* s, size_t n, uint64_t* out, void* st) {                  \
    std::lock_guard<std::mutex> lk(api_mutex());                                                                         \
    if (int rc = api_ensure_init()) return rc;                                                                           \
    return eng_##TAG.run_device((const uint64_t*)b, (const uint8_t*)inf, (const uint32_t*)s, n, out, (hipStream_t)st);   \
  }                                                                                                                      \
  void msm_big_timings_##TAG(float ms[5], int cfg[3]) {                                                                  \
    const MsmTimings& t = eng_##TAG.tm;                                                                                  \
    ms[0] = t.convert; ms[1] = t.sort; ms[2] = t.accumulate; ms[3] = t.reduce; ms[4] = t.total;                          \
    cfg[0] = eng_##TAG.last_c; cfg[1] = eng_##TAG.last_nw; cfg[2] = (int)eng_##TAG.last_buckets;                         \
  }                                                                                                                      \
  void msm_big_set_c_##TAG(int c) { eng_##TAG.force_c = c; }                                                             \
  }

#define CELO_DEFINE_MSM_AUX_UNIT(G, TAG)                                                                                 \
  namespace celo {                                                                                                       \
  static MsmEngine<G> eng_aux_##TAG;                                                                                     \
  static int last_was_batch_##TAG = 0;                                                                                   \
  void msm_big_timings_##TAG(float ms[5], int cfg[3]);                                                                   \
  void msm_big_set_c_##TAG(int c);                                                                                       \
  int msm_batch_host_##TAG(const uint64_t* b, const uint8_t* inf, const uint64_t* s, const uint32_t* off, size_t m, uint64_t* out) { \
    std::lock_guard<std::mutex> lk(api_mutex());                                                                         \
    if (int rc = api_ensure_init()) return rc;                                                                           \
    last_was_batch_##TAG = 1;                                                                                            \
    return eng_aux_##TAG.run_batch_host(b, inf, s, off, m, out, nullptr);                                                \
  }                                                                                                                      \
  int msm_timings_##TAG(float ms[5], int cfg[3]) {                                                                       \
    std::lock_guard<std::mutex> lk(api_mutex());                                                                         \
    if (last_was_batch_##TAG) {                                                                                          \
      const MsmTimings& t = eng_aux_##TAG.tm;                                                                            \
      ms[0] = t.convert; ms[1] = t.sort; ms[2] = t.accumulate; ms[3] = t.reduce; ms[4] = t.total;                        \
      cfg[0] = eng_aux_##TAG.last_c; cfg[1] = eng_aux_##TAG.last_nw; cfg[2] = (int)eng_aux_##TAG.last_buckets;           \
    } else msm_big_timings_##TAG(ms, cfg);                                                                               \
    return 0;                                                                                                            \
  }                                                                                                                      \
  void msm_note_big_call_##TAG() { last_was_batch_##TAG = 0; }                                                           \
  int msm_set_c_##TAG(int c) {                                                                                           \
    std::lock_guard<std::mutex> lk(api_mutex());                                                                         \
    eng_aux_##TAG.force_c = c;                                                                                           \
    msm_big_set_c_##TAG(c);                                                                                              \
    return 0;                                                                                                            \
  }                                                                                                                      \
  int gen_points_##TAG(void* d, size_t n, uint64_t seed, const uint64_t* g, void* st) {                                  \
    return gen_points_impl<G::F>(d, n, seed, g, st);                                                                     \
  }                                                                                                                      \
  int sum_jac_##TAG(const uint64_t* jac, size_t k, uint64_t* out) { return sum_jacobian_impl<G::F>(jac, k, out); }       \
  }
