// Host micro-benchmark of host64.h (the MSM's Horner epilogue arithmetic): ns per Fq product, per XYZZ doubling / addition,
// next to the 28-bit-limb device representation run on the host, and the whole 16-window step list on both epilogue paths
// (host64.h and the AVX-512 IFMA one of host_ifma.cpp).
// g++ -O3 -std=c++17 -Icelo-bls-snark-rs_amd/csrc tools/bench_host64.cpp celo-bls-snark-rs_amd/build/host_ifma.o celo-bls-snark-rs_amd/build/host_cpu.o
#include "host64.h"
#include "curve.h"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace celo;
extern "C" int celo_ifma_available();
extern "C" int celo_ifma_horner_377(const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out, int* inf);
typedef HFp<P377> H;
int main() {
  H a = H::one(), b = H::one() + H::one() + H::one();
  for (int i = 0; i < 50; i++) { a = a * b + b; b = b * b + a; }
  auto t0 = std::chrono::steady_clock::now();
  const int N = 2000000;
  for (int i = 0; i < N; i++) a = a * b;
  auto t1 = std::chrono::steady_clock::now();
  printf("HFp<P377> mul: %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / N, (unsigned long long)a.v[0]);
  typedef Fp<P377> F;
  uint64_t w[6];
  a.store(w);
  F x = F::from_ark(w), y = F::from_ark(b.v);
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N / 4; i++) x = F::mul(x, y);
  t1 = std::chrono::steady_clock::now();
  x.to_ark(w);
  printf("Fp<P377> (28-bit limbs) mul on the host: %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / (N / 4), (unsigned long long)w[0]);
  HXyzz<H> p = {a, b, H::one(), H::one()}, q = {b, a, H::one(), H::one()};
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20000; i++) { p = hxyzz_dbl(p); hxyzz_add(p, q); }
  t1 = std::chrono::steady_clock::now();
  printf("hxyzz dbl+add: %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / 20000, (unsigned long long)p.X.v[0]);
  // the step list of a 2^20-term MSM (16 windows of 16 bits): 256 slots of random-looking coordinates
  const int nw = 16, LB = 15;
  std::vector<uint64_t> pts((size_t)(LB + 1) * nw * 32);
  H r = a;
  for (size_t s = 0; s < (size_t)(LB + 1) * nw; s++)
    for (int e = 0; e < 4; e++) { r = r * b + a; r.store(pts.data() + s * 32 + e * 6); }
  std::vector<int32_t> order;
  for (int w = nw - 1; w >= 0; w--) {
    order.push_back(-1);
    for (int l = 1; l <= LB; l++) order.push_back(l * nw + w);
    order.push_back(w | HORNER_NODBL);
  }
  const int reps = 200;
  t0 = std::chrono::steady_clock::now();
  HXyzz<H> acc;
  for (int i = 0; i < reps; i++) acc = host64_horner<H>(pts.data(), 32, 6, order.data(), (int)order.size());
  t1 = std::chrono::steady_clock::now();
  printf("host64 Horner, 16 windows: %.1f us (%llx)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, (unsigned long long)acc.X.v[0]);
  if (celo_ifma_available()) {
    uint64_t out[24];
    int inf = 0, rc = 0;
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) rc |= celo_ifma_horner_377(pts.data(), 32, order.data(), (int)order.size(), out, &inf);
    t1 = std::chrono::steady_clock::now();
    printf("IFMA Horner, 16 windows: %.1f us (rc %d, %llx)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, rc, (unsigned long long)out[0]);
  } else printf("no AVX-512 IFMA\n");
  return 0;
}
