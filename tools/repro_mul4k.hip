// Reproducer for the round-3 open finding: the SIGNED form of Fp::mul4k - the Fq2 pass R t - Y1 PPP of every G2
// mixed addition - gave wrong G2 sums from 2^17 terms up, while the unsigned form matches the oracle everywhere.
// This tool runs chains of xyzz_madd over Fq2 on real G2 points and evaluates BOTH forms of the pass on the SAME operands in the same
// lane at every addition; any difference is recorded with its eight operands.  The two forms are the same function modulo 2^64 per
// column for operands below 2^31, so a recorded difference names either an operand outside that range (an arithmetic bound the host
// tracker does not model) or wrong code for one of the two bodies.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/repro_mul4k tools/repro_mul4k.hip && build/repro_mul4k [lanes_log2] [chain]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../celo-bls-snark-rs_amd/csrc/curve.h"
#include "../celo-bls-snark-rs_amd/csrc/fp2.h"
using namespace celo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef Fp<P377> Fq;
typedef Fp2<P377> F;
constexpr int AW = 2 * F::WORDS;     // words per affine point
constexpr int REC = 8 * 16 + 2 * 16 * 2;   // words per record: 8 operands + two results (c0 pass unsigned/signed)

// P_i = k_i G2, k_i odd 32-bit, affine, device form
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

__device__ bool same(const Fq& a, const Fq& b) { uint32_t o = 0; for (int i = 0; i < Fq::L; i++) o |= a.l[i] ^ b.l[i]; return o == 0; }
__device__ void put(uint32_t* r, const Fq& a) { for (int i = 0; i < 16; i++) r[i] = i < Fq::L ? a.l[i] : 0u; }

// xyzz_madd with both forms of the Y3 pass side by side (curve.h xyzz_madd; the result that continues the chain is the unsigned one)
__device__ void madd_diag(Xyzz<F>& a, const Affine<F>& p, uint32_t* recs, uint32_t* nrec, uint32_t cap, uint32_t* range_hits) {
  if (a.is_identity()) { a = Xyzz<F>::from_affine(p); return; }
  F U2 = F::mul_nn(p.x, a.ZZ), S2 = F::mul_nn(p.y, a.ZZZ);
  F Pd = F::prep(F::template sub<32, 1>(U2, a.X)), R = F::prep(F::template sub<16, 1>(S2, a.Y));
  if (Pd.is_zero_mod_p()) { a = R.is_zero_mod_p() ? xyzz_dbl_affine(p) : Xyzz<F>::identity(); return; }
  F PP = F::sqr_nn(Pd), PPP = F::mul_nn(Pd, PP), Q = F::mul_nn(a.X, PP), R2 = F::sqr_nn(R);
  F s = F::add(F::add(PPP, Q), Q);
  F X3 = F::norm(F::template sub<16, 3>(R2, s));
  F t = F::prep(F::template sub<32, 1>(Q, X3));
  const Fq* ops[8] = {&R.c0, &t.c0, &R.c1, &t.c1, &a.Y.c0, &PPP.c0, &a.Y.c1, &PPP.c1};
  // operand range the signed form needs: every limb (and 5 x the limbs of R.c1 / Y.c1) below 2^31
  uint32_t bad = 0;
  for (int o = 0; o < 8; o++) for (int i = 0; i < Fq::L; i++) { const uint32_t v = ops[o]->l[i]; if (v >> 31 || ((o == 2 || o == 6) && v * 5ull >> 31)) bad = 1; if (i < Fq::L - 1 && v >> 28) bad |= 2; }
  if (bad) atomicAdd(range_hits + (bad & 1 ? 0 : 1), 1u);
  const Fq u0 = Fq::mul4k<-5, false>(R.c0, t.c0, R.c1, t.c1, a.Y.c0, PPP.c0, a.Y.c1, PPP.c1);
  const Fq s0 = Fq::mul4k<-5, true>(R.c0, t.c0, R.c1, t.c1, a.Y.c0, PPP.c0, a.Y.c1, PPP.c1);
  const Fq u1 = Fq::mul4k<1, false>(R.c0, t.c1, R.c1, t.c0, a.Y.c0, PPP.c1, a.Y.c1, PPP.c0);
  const Fq s1 = Fq::mul4k<1, true>(R.c0, t.c1, R.c1, t.c0, a.Y.c0, PPP.c1, a.Y.c1, PPP.c0);
  if (!same(u0, s0) || !same(u1, s1)) {
    const uint32_t k = atomicAdd(nrec, 1u);
    if (k < cap) {
      uint32_t* r = recs + (size_t)k * REC;
      for (int o = 0; o < 8; o++) put(r + 16 * o, *ops[o]);
      put(r + 128, u0); put(r + 144, s0); put(r + 160, u1); put(r + 176, s1);
    }
  }
  a.ZZ = F::mul_nn(a.ZZ, PP); a.ZZZ = F::mul_nn(a.ZZZ, PPP); a.X = X3; a.Y = {u0, u1};
}

__global__ void __launch_bounds__(256) k_chain(const uint32_t* pts, uint32_t npts, uint32_t chain, uint32_t* recs, uint32_t* nrec, uint32_t cap, uint32_t* range_hits, uint32_t* sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = tid * 2654435761u + 12345u;
  Xyzz<F> acc = Xyzz<F>::identity();
  for (uint32_t k = 0; k < chain; k++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t* q = pts + (size_t)((h >> 8) % npts) * AW;
    Affine<F> p = {F::load(q), F::load(q + F::WORDS)};
    if (h & 1) p = affine_neg(p);
    madd_diag(acc, p, recs, nrec, cap, range_hits);
  }
  uint32_t x = 0;
  for (int i = 0; i < Fq::L; i++) x ^= acc.X.c0.l[i] ^ acc.Y.c1.l[i];
  sink[tid] = x;
}

// the same comparison at xyzz_dbl_affine's pass M t - W y (the doubling branch of a mixed addition: equal points in one bucket), for every
// table point and its negation
__global__ void __launch_bounds__(256) k_dbl_affine_diag(const uint32_t* pts, uint32_t npts, uint32_t* recs, uint32_t* nrec, uint32_t cap, uint32_t* range_hits) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= 2 * npts) return;
  const uint32_t* q = pts + (size_t)(tid >> 1) * AW;
  Affine<F> p = {F::load(q), F::load(q + F::WORDS)};
  if (tid & 1) p = affine_neg(p);
  F U = F::prep(F::dbl(p.y)), V = F::sqr_nn(U), W = F::mul_nn(U, V), S = F::mul_nn(p.x, V), xx = F::sqr_nn(p.x);
  F M = F::prep(F::add(F::add(xx, xx), xx)), M2 = F::sqr_nn(M);
  F X3 = F::norm(F::template sub<16, 3>(M2, F::dbl(S)));
  F t = F::prep(F::template sub<32, 1>(S, X3));
  const Fq* ops[8] = {&M.c0, &t.c0, &M.c1, &t.c1, &W.c0, &p.y.c0, &W.c1, &p.y.c1};
  uint32_t bad = 0;
  for (int o = 0; o < 8; o++) for (int i = 0; i < Fq::L; i++) { const uint32_t v = ops[o]->l[i]; if (v >> 31 || ((o == 2 || o == 6) && v * 5ull >> 31)) bad = 1; if (i < Fq::L - 1 && v >> 28) bad |= 2; }
  if (bad) atomicAdd(range_hits + (bad & 1 ? 0 : 1), 1u);
  const Fq u0 = Fq::mul4k<-5, false>(M.c0, t.c0, M.c1, t.c1, W.c0, p.y.c0, W.c1, p.y.c1), s0 = Fq::mul4k<-5, true>(M.c0, t.c0, M.c1, t.c1, W.c0, p.y.c0, W.c1, p.y.c1);
  const Fq u1 = Fq::mul4k<1, false>(M.c0, t.c1, M.c1, t.c0, W.c0, p.y.c1, W.c1, p.y.c0), s1 = Fq::mul4k<1, true>(M.c0, t.c1, M.c1, t.c0, W.c0, p.y.c1, W.c1, p.y.c0);
  if (!same(u0, s0) || !same(u1, s1)) {
    const uint32_t k = atomicAdd(nrec, 1u);
    if (k < cap) {
      uint32_t* r = recs + (size_t)k * REC;
      for (int o = 0; o < 8; o++) put(r + 16 * o, *ops[o]);
      put(r + 128, u0); put(r + 144, s0); put(r + 160, u1); put(r + 176, s1);
    }
  }
}

int main(int argc, char** argv) {
  const uint32_t lanes = 1u << (argc > 1 ? atoi(argv[1]) : 17), chain = argc > 2 ? (uint32_t)atoi(argv[2]) : 64, npts = 1u << 16, cap = 16;
  uint32_t *d_pts, *d_recs, *d_n, *d_sink, *d_rng;
  CK(hipMalloc(&d_pts, (size_t)npts * AW * 4)); CK(hipMalloc(&d_recs, (size_t)cap * REC * 4)); CK(hipMalloc(&d_n, 4)); CK(hipMalloc(&d_rng, 8));
  CK(hipMalloc(&d_sink, (size_t)lanes * 4));
  CK(hipMemset(d_n, 0, 4)); CK(hipMemset(d_rng, 0, 8));
  hipLaunchKernelGGL(k_points, dim3(npts / 128), dim3(128), 0, 0, d_pts, npts);
  hipLaunchKernelGGL(k_chain, dim3(lanes / 256), dim3(256), 0, 0, d_pts, npts, chain, d_recs, d_n, cap, d_rng, d_sink);
  CK(hipDeviceSynchronize());
  uint32_t n = 0, rng[2];
  CK(hipMemcpy(&n, d_n, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rng, d_rng, 8, hipMemcpyDeviceToHost));
  printf("%u lanes x %u mixed additions over Fq2: %u additions where the signed and the unsigned pass differ; operands with a limb >= 2^31 (or 5x one): %u, "
         "with a lower limb >= 2^28: %u\n", lanes, chain, n, rng[0], rng[1]);
  {
    CK(hipMemset(d_n, 0, 4)); CK(hipMemset(d_rng, 0, 8));
    hipLaunchKernelGGL(k_dbl_affine_diag, dim3(2 * npts / 256), dim3(256), 0, 0, d_pts, npts, d_recs, d_n, cap, d_rng);
    CK(hipDeviceSynchronize());
    uint32_t n2 = 0, rng2[2];
    CK(hipMemcpy(&n2, d_n, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rng2, d_rng, 8, hipMemcpyDeviceToHost));
    printf("xyzz_dbl_affine's pass on %u points and their negations: %u differences; operands with a limb >= 2^31 (or 5x one): %u, with a lower limb >= 2^28: %u\n",
           npts, n2, rng2[0], rng2[1]);
    if (n2) n = n2;      // print these records
  }
  std::vector<uint32_t> recs((size_t)cap * REC);
  CK(hipMemcpy(recs.data(), d_recs, recs.size() * 4, hipMemcpyDeviceToHost));
  const char* names[12] = {"a.c0", "b.c0", "a.c1", "b.c1", "c.c0", "d.c0", "c.c1", "d.c1", "c0 unsigned", "c0 signed", "c1 unsigned", "c1 signed"};
  for (uint32_t k = 0; k < n && k < cap; k++) {
    printf("-- difference %u\n", k);
    for (int o = 0; o < 12; o++) { printf("  %-12s", names[o]); for (int i = 0; i < 14; i++) printf(" %08x", recs[(size_t)k * REC + 16 * o + i]); printf("\n"); }
  }
  return n ? 2 : 0;
}
