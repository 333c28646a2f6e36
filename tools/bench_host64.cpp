// Host micro-benchmark of host64.h (the MSM's Horner epilogue arithmetic): ns per Fq product, per XYZZ doubling / addition,
// next to the 28-bit-limb device representation run on the host.  g++ -O3 -std=c++17 -Icelo-bls-snark-rs_amd/csrc
#include "host64.h"
#include "curve.h"
#include <chrono>
#include <cstdio>
using namespace celo;
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
  return 0;
}
