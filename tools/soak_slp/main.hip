// Differential soak of the headline kernel against the SLP vectorizer (csrc/Makefile SLP_UNITS: empty since the end of round 6, this soak is the
// evidence behind any future exception): the library's k_accumulate<G1_377> built
// WITH the pass against the same kernel built with -fno-slp-vectorize, on identical bucket runs, every partial sum compared limb for limb (same
// additions in the same order: the two kernels must agree exactly).  Round 5 named the SLP vectorizer as the pass that miscompiles the signed
// instantiation of k_accumulate<G2_377> (tools/repro_acc/REPORT.md); VERDICT r5 item 1c asks for this soak before a unit may keep the pass.
// Runs carry the cold paths on purpose: one run in 8 starts with a doubled point (affine + affine doubling), one in 8 with a cancelling pair
// followed by a repeated point (identity accumulator, then the doubling branch of the mixed addition), the rest are random signed points.
//   tools/soak_slp/build.sh && celo-bls-snark-rs_amd/build/soak_slp [runs_log2=18] [len=32] [seeds=13]      (defaults: 1.08e8 additions)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../celo-bls-snark-rs_amd/csrc/curve.h"
using namespace celo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef Fp<P377> F;
constexpr int AW = 2 * F::WORDS, XW = 4 * F::WORDS;
extern "C" void launch_acc_s(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t);
extern "C" void launch_acc_n(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t);

__global__ void __launch_bounds__(128) k_points(uint32_t* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<F> g = {F::from_limbs(T377::G1_GEN_X), F::from_limbs(T377::G1_GEN_Y)};
  const uint32_t k = (i * 2654435761u) | 1u;
  Xyzz<F> acc = Xyzz<F>::identity();
  for (int b = 31; b >= 0; b--) { acc = xyzz_dbl(acc); if ((k >> b) & 1) xyzz_madd(acc, g); }
  const F t = F::inv(F::mul(acc.ZZ, acc.ZZZ));
  const F x = F::mul(acc.X, F::mul(t, acc.ZZZ)), y = F::mul(acc.Y, F::mul(t, acc.ZZ));
  x.store(out + (size_t)i * AW); y.store(out + (size_t)i * AW + F::WORDS);
}
int main(int argc, char** argv) {
  const uint32_t NP = 1u << (argc > 1 ? atoi(argv[1]) : 18), L = argc > 2 ? (uint32_t)atoi(argv[2]) : 32, seeds = argc > 3 ? (uint32_t)atoi(argv[3]) : 13, npts = 1u << 16;
  if (L < 4) { printf("len >= 4\n"); return 1; }
  uint32_t *d_pts, *d_sorted, *d_pstart, *d_plen, *d_order, *d_nwork, *d_ps, *d_pn;
  CK(hipMalloc(&d_pts, (size_t)npts * AW * 4)); CK(hipMalloc(&d_sorted, (size_t)NP * L * 4)); CK(hipMalloc(&d_pstart, NP * 4)); CK(hipMalloc(&d_plen, NP * 4));
  CK(hipMalloc(&d_order, NP * 4)); CK(hipMalloc(&d_nwork, 4)); CK(hipMalloc(&d_ps, (size_t)NP * XW * 4)); CK(hipMalloc(&d_pn, (size_t)NP * XW * 4));
  hipLaunchKernelGGL(k_points, dim3(npts / 128), dim3(128), 0, 0, d_pts, npts);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> sorted((size_t)NP * L), pstart(NP), plen(NP), order(NP), ps((size_t)NP * XW), pn((size_t)NP * XW);
  for (uint32_t i = 0; i < NP; i++) { pstart[i] = i * L; order[i] = i; }
  CK(hipMemcpy(d_pstart, pstart.data(), NP * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_order, order.data(), NP * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_nwork, &NP, 4, hipMemcpyHostToDevice));
  uint64_t total = 0, bad_total = 0, cold = 0;
  for (uint32_t seed = 1; seed <= seeds; seed++) {
    uint32_t h = seed * 0x9E3779B9u;
    for (size_t e = 0; e < sorted.size(); e++) { h = h * 1664525u + 1013904223u; sorted[e] = ((h >> 9) % npts) | ((h & 0x100u) ? 0x80000000u : 0u); }
    for (uint32_t i = 0; i < NP; i++) {
      uint32_t* r = &sorted[(size_t)i * L];
      plen[i] = (i & 63) == 63 ? 1 + (r[0] % L) : L;                                  // ragged runs too (lengths 1 .. L)
      if ((i & 7) == 1) r[1] = r[0];                                                  // P, P: the affine + affine doubling
      if ((i & 7) == 2) { r[1] = r[0] ^ 0x80000000u; r[3] = r[2]; }                   // P, -P, Q, Q: identity, then the mixed addition's doubling branch
      if ((i & 7) == 3) { r[2] = r[1] ^ 0x80000000u; r[3] = r[0] ^ 0x80000000u; }     // P, Q, -Q, -P: two cancellations through the general path
    }
    CK(hipMemcpy(d_sorted, sorted.data(), sorted.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_plen, plen.data(), NP * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_ps, 0xA5, (size_t)NP * XW * 4)); CK(hipMemset(d_pn, 0x5A, (size_t)NP * XW * 4));
    launch_acc_s(d_pts, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_ps, NP);
    launch_acc_n(d_pts, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_pn, NP);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ps.data(), d_ps, ps.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(pn.data(), d_pn, pn.size() * 4, hipMemcpyDeviceToHost));
    uint64_t bad = 0;
    for (uint32_t i = 0; i < NP; i++) {
      total += plen[i];
      if (memcmp(&ps[(size_t)i * XW], &pn[(size_t)i * XW], XW * 4)) bad++;
      if ((i & 7) >= 1 && (i & 7) <= 3 && plen[i] >= 4) cold++;                      // runs built around a doubling / cancellation (see above)
    }
    bad_total += bad;
    printf("seed %u: %u runs, %llu additions so far: %llu runs differ between the SLP and the no-SLP kernel\n", seed, NP, (unsigned long long)total, (unsigned long long)bad);
  }
  printf("soak_slp k_accumulate<G1_377>: %llu additions, %llu differing runs, %llu of the runs carry a doubling or a cancellation by construction\n",
         (unsigned long long)total, (unsigned long long)bad_total, (unsigned long long)cold);
  return bad_total ? 2 : 0;
}
