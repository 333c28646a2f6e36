// Round-3 open finding, capture tool: the library's k_accumulate<G2_377> compiled with the signed form of xyzz_madd's Fq2 pass
// (-DCELO_MUL4K_SGN_SITES=4) against the same kernel compiled unsigned, on identical bucket runs; every partial sum is compared limb for
// limb (same additions in the same order: the two kernels must agree exactly).  For a run that differs the tool finds the first addition
// at which the two kernels part (re-running with the run truncated), replays the prefix on the HOST (the same templates, bit-exact with
// the unsigned kernel) and prints the accumulator, the point, and both results: one wrong addition, fully specified.
//   tools/repro_acc/build.sh && celo-bls-snark-rs_amd/build/repro_acc [runs_log2=16] [len=24] [seeds=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../celo-bls-snark-rs_amd/csrc/curve.h"
#include "../../celo-bls-snark-rs_amd/csrc/fp2.h"
using namespace celo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef Fp<P377> Fq;
typedef Fp2<P377> F;
constexpr int AW = 2 * F::WORDS, XW = 4 * F::WORDS;
extern "C" void launch_acc_u(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t);
extern "C" void launch_acc_s(const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*, uint32_t);

__global__ void __launch_bounds__(128) k_points(uint32_t* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<F> g = {{Fq::from_limbs(T377::G2_GEN_X0), Fq::from_limbs(T377::G2_GEN_X1)}, {Fq::from_limbs(T377::G2_GEN_Y0), Fq::from_limbs(T377::G2_GEN_Y1)}};
  const uint32_t k = (i * 2654435761u) | 1u;
  Xyzz<F> acc = Xyzz<F>::identity();
  for (int b = 31; b >= 0; b--) { acc = xyzz_dbl(acc); if ((k >> b) & 1) xyzz_madd(acc, g); }
  const F t = F::inv(F::mul(acc.ZZ, acc.ZZZ));
  const F x = F::mul(acc.X, F::mul(t, acc.ZZZ)), y = F::mul(acc.Y, F::mul(t, acc.ZZ));
  x.store(out + (size_t)i * AW); y.store(out + (size_t)i * AW + F::WORDS);
}
static void print_f(const char* name, const F& v) {
  printf("    %-8s c0", name); for (int i = 0; i < 14; i++) printf(" %08x", v.c0.l[i]);
  printf("\n    %-8s c1", ""); for (int i = 0; i < 14; i++) printf(" %08x", v.c1.l[i]); printf("\n");
}
int main(int argc, char** argv) {
  const uint32_t NP = 1u << (argc > 1 ? atoi(argv[1]) : 16), L = argc > 2 ? (uint32_t)atoi(argv[2]) : 24, seeds = argc > 3 ? (uint32_t)atoi(argv[3]) : 4, npts = 1u << 14;
  uint32_t *d_pts, *d_sorted, *d_pstart, *d_plen, *d_order, *d_nwork, *d_pu, *d_ps;
  CK(hipMalloc(&d_pts, (size_t)npts * AW * 4)); CK(hipMalloc(&d_sorted, (size_t)NP * L * 4)); CK(hipMalloc(&d_pstart, NP * 4)); CK(hipMalloc(&d_plen, NP * 4));
  CK(hipMalloc(&d_order, NP * 4)); CK(hipMalloc(&d_nwork, 4)); CK(hipMalloc(&d_pu, (size_t)NP * XW * 4)); CK(hipMalloc(&d_ps, (size_t)NP * XW * 4));
  hipLaunchKernelGGL(k_points, dim3(npts / 128), dim3(128), 0, 0, d_pts, npts);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> h_pts((size_t)npts * AW);
  CK(hipMemcpy(h_pts.data(), d_pts, h_pts.size() * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> sorted((size_t)NP * L), pstart(NP), plen(NP), order(NP), pu((size_t)NP * XW), ps((size_t)NP * XW);
  for (uint32_t i = 0; i < NP; i++) { pstart[i] = i * L; order[i] = i; }
  CK(hipMemcpy(d_pstart, pstart.data(), NP * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_order, order.data(), NP * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_nwork, &NP, 4, hipMemcpyHostToDevice));
  uint64_t total = 0, bad_total = 0;
  for (uint32_t seed = 1; seed <= seeds; seed++) {
    uint32_t h = seed * 0x9E3779B9u;
    for (size_t e = 0; e < sorted.size(); e++) { h = h * 1664525u + 1013904223u; sorted[e] = ((h >> 9) % npts) | ((h & 0x100u) ? 0x80000000u : 0u); }
    for (uint32_t i = 0; i < NP; i++) plen[i] = L;
    CK(hipMemcpy(d_sorted, sorted.data(), sorted.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_plen, plen.data(), NP * 4, hipMemcpyHostToDevice));
    launch_acc_u(d_pts, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_pu, NP);
    launch_acc_s(d_pts, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_ps, NP);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(pu.data(), d_pu, pu.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ps.data(), d_ps, ps.size() * 4, hipMemcpyDeviceToHost));
    total += (uint64_t)NP * L;
    std::vector<uint32_t> bad;
    for (uint32_t i = 0; i < NP; i++) if (memcmp(&pu[(size_t)i * XW], &ps[(size_t)i * XW], XW * 4)) bad.push_back(i);
    bad_total += bad.size();
    printf("seed %u: %u runs x %u additions: %zu runs differ between the unsigned and the signed kernel\n", seed, NP, L, bad.size());
    for (size_t q = 0; q < bad.size() && q < 3; q++) {
      const uint32_t i = bad[q];
      // first addition at which the kernels part: shortest prefix with different results (only run i is truncated)
      uint32_t lo = 1, hi = L;
      std::vector<uint32_t> a(XW), b(XW);
      while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        CK(hipMemcpy(d_plen + i, &mid, 4, hipMemcpyHostToDevice));
        launch_acc_u(d_pts, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_pu, NP);
        launch_acc_s(d_pts, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_ps, NP);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(a.data(), d_pu + (size_t)i * XW, XW * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_ps + (size_t)i * XW, XW * 4, hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b.data(), XW * 4)) hi = mid; else lo = mid + 1;
      }
      CK(hipMemcpy(d_plen + i, &L, 4, hipMemcpyHostToDevice));
      const uint32_t k = lo;                       // the k-th point of the run (1-based) is the first whose addition differs
      // host replay of the prefix
      Xyzz<F> acc = Xyzz<F>::identity(), before = acc;
      Affine<F> last = {F::zero(), F::zero()};
      for (uint32_t j = 0; j < k; j++) {
        const uint32_t v = sorted[(size_t)i * L + j];
        Affine<F> p = {F::load(&h_pts[(size_t)(v & 0x7fffffffu) * AW]), F::load(&h_pts[(size_t)(v & 0x7fffffffu) * AW + F::WORDS])};
        if (v >> 31) p = affine_neg(p);
        before = acc; last = p;
        xyzz_madd(acc, p);
      }
      uint32_t hostw[XW];
      acc.X.store(hostw); acc.Y.store(hostw + F::WORDS); acc.ZZ.store(hostw + 2 * F::WORDS); acc.ZZZ.store(hostw + 3 * F::WORDS);
      // device results of exactly that prefix
      CK(hipMemcpy(d_plen + i, &k, 4, hipMemcpyHostToDevice));
      launch_acc_u(d_pts, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_pu, NP);
      launch_acc_s(d_pts, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_ps, NP);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(a.data(), d_pu + (size_t)i * XW, XW * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_ps + (size_t)i * XW, XW * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(d_plen + i, &L, 4, hipMemcpyHostToDevice));
      const bool u_ok = !memcmp(a.data(), hostw, XW * 4), s_ok = !memcmp(b.data(), hostw, XW * 4);
      printf("  run %u (lane %u of wave %u): first differing addition = point %u of %u; host replay == unsigned kernel: %s, == signed kernel: %s\n", i, i & 63, i >> 6, k, L,
             u_ok ? "yes" : "NO", s_ok ? "yes" : "NO");
      printf("  accumulator before the addition (host replay):\n");
      print_f("X", before.X); print_f("Y", before.Y); print_f("ZZ", before.ZZ); print_f("ZZZ", before.ZZZ);
      printf("  point added (affine, sign applied):\n"); print_f("x", last.x); print_f("y", last.y);
      const char* nm[4] = {"X", "Y", "ZZ", "ZZZ"};
      for (int c = 0; c < 4; c++) {
        if (memcmp(&a[c * F::WORDS], &b[c * F::WORDS], F::WORDS * 4)) {
          printf("  coordinate %s differs:\n    unsigned", nm[c]); for (int w = 0; w < F::WORDS; w++) printf(" %08x", a[c * F::WORDS + w]);
          printf("\n    signed  "); for (int w = 0; w < F::WORDS; w++) printf(" %08x", b[c * F::WORDS + w]); printf("\n");
        }
      }
    }
  }
  printf("total: %llu additions per kernel, %llu runs differ\n", (unsigned long long)total, (unsigned long long)bad_total);
  // the UNSIGNED kernel (the instantiation the library ships) against the host replay of the same templates, limb for limb, on the first
  // `hc` runs of the last seed: the regression guard of the shipped kernel (tests/test_msm_gpu.py runs this tool)
  const uint32_t hc = argc > 4 ? (uint32_t)atoi(argv[4]) : 4096;
  uint32_t host_bad = 0;
  for (uint32_t i = 0; i < hc && i < NP; i++) {
    Xyzz<F> acc = Xyzz<F>::identity();
    for (uint32_t j = 0; j < L; j++) {
      const uint32_t v = sorted[(size_t)i * L + j];
      Affine<F> p = {F::load(&h_pts[(size_t)(v & 0x7fffffffu) * AW]), F::load(&h_pts[(size_t)(v & 0x7fffffffu) * AW + F::WORDS])};
      if (v >> 31) p = affine_neg(p);
      xyzz_madd(acc, p);
    }
    uint32_t hostw[XW];
    acc.X.store(hostw); acc.Y.store(hostw + F::WORDS); acc.ZZ.store(hostw + 2 * F::WORDS); acc.ZZZ.store(hostw + 3 * F::WORDS);
    if (memcmp(hostw, &pu[(size_t)i * XW], XW * 4)) host_bad++;
  }
  printf("host replay of %u runs x %u additions vs the unsigned kernel: %u differ\n", hc < NP ? hc : NP, L, host_bad);
  return (bad_total ? 2 : 0) | (host_bad ? 4 : 0);
}
