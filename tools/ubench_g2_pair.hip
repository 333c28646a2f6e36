// Sizing experiment (round 5, DESIGN.md section 10 "(o)"): the mixed addition of G2 (Fq2 coordinates) with the two halves of every Fq2 value on
// two adjacent lanes (QHex377's pair product: Fp::mul2s after one DPP exchange) against the library's one-lane Fp2 xyzz_madd, both as
// register-resident loops.  The one-lane form needs 256 VGPRs + AGPRs and runs one wave per SIMD; the pair form holds half the state per lane.
// Prints mixed additions per second for both and checks that they compute the same point.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Icelo-bls-snark-rs_amd/csrc -o tools/ubench_g2_pair tools/ubench_g2_pair.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "curve.h"
#include "fp2.h"
#include "pairing_lanes.h"
using namespace celo;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef Fp<P377> Fq1;
typedef Fp2<P377> F2;
typedef QHex377 Q;

// acc += p on a lane pair: every value is the lane's half (h = lane & 1) of an Fq2 element, normalised limbs
struct PairAcc { Fq1 X, Y, ZZ, ZZZ; };
__device__ __forceinline__ void madd_pair(PairAcc& a, const Fq1& px, const Fq1& py) {
  const Fq1 U2 = Q::mul(px, a.ZZ), S2 = Q::mul(py, a.ZZZ);
  const Fq1 Pd = Q::template sub<32>(U2, a.X), R = Q::template sub<16>(S2, a.Y);       // (X < 17 p, Y < 5 p from the addition before: curve.h's slack)
  // the exact-zero test of Pd, pair-uniform: both halves zero (the doubling / cancellation branch is cold and left out of this loop)
  const int z = Pd.is_zero_mod_p() ? 1 : 0;
  const int zo = __builtin_amdgcn_mov_dpp(z, 0xB1, 0xF, 0xF, true);
  if (z & zo) { a.ZZ = Fq1::zero(); return; }
  const Fq1 PP = Q::mul(Pd, Pd), PPP = Q::mul(Pd, PP), Qv = Q::mul(a.X, PP), R2 = Q::mul(R, R);
  const Fq1 s = Q::add(Q::add(PPP, Qv), Qv);
  const Fq1 X3 = Q::template sub<16>(R2, s);
  const Fq1 t = Q::template sub<32>(Qv, X3);
  const Fq1 Y3 = Q::template sub<4>(Q::mul(R, t), Q::mul(a.Y, PPP));
  a.ZZ = Q::mul(a.ZZ, PP);
  a.ZZZ = Q::mul(a.ZZZ, PPP);
  a.X = X3; a.Y = Y3;
}

// inputs: per point-add slot i: acc (X, Y, ZZ, ZZZ as Fq2 = 2 x 16 words each: 14 limbs + 2 of padding) and p (x, y); outputs the same layout
__device__ __forceinline__ void desync_wave(int on);
__global__ void __launch_bounds__(256) k_single(const uint32_t* in, uint32_t* out, int iters) {
  desync_wave(iters > 100);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t* s = in + i * 6 * 32;
  Xyzz<F2> acc = {F2::load(s), F2::load(s + 32), F2::load(s + 64), F2::load(s + 96)};
  Affine<F2> p = {F2::load(s + 128), F2::load(s + 160)};
  if (iters < 0) acc.X = F2::mul(acc.X, acc.Y);
  for (int k = 0; k < iters; k++) xyzz_madd(acc, p);
  uint32_t* d = out + i * 4 * 32;
  F2::norm(acc.X).store(d); F2::norm(acc.Y).store(d + 32); F2::norm(acc.ZZ).store(d + 64); F2::norm(acc.ZZZ).store(d + 96);
}
__device__ __forceinline__ void desync_wave(int on) {      // waves enter their loops at different times (instruction-cache behaviour of a real launch)
  if (!on) return;
  unsigned w = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u;
  unsigned spins = (w >> 19) & 0x1fff;
  for (unsigned i = 0; i < spins; i++) __builtin_amdgcn_s_sleep(8);
}
__global__ void __launch_bounds__(256) k_pair(const uint32_t* in, uint32_t* out, int iters) {
  desync_wave(iters > 100);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1; const int h = (int)(t & 1);
  const uint32_t* s = in + i * 6 * 32 + h * 16;
  PairAcc acc = {Fq1::load(s), Fq1::load(s + 32), Fq1::load(s + 64), Fq1::load(s + 96)};
  const Fq1 px = Fq1::load(s + 128), py = Fq1::load(s + 160);
  if (iters < 0) acc.X = Q::mul(acc.X, acc.Y);
  for (int k = 0; k < iters; k++) madd_pair(acc, px, py);
  uint32_t* d = out + i * 4 * 32 + h * 16;
  Fq1::norm(acc.X).store(d); Fq1::norm(acc.Y).store(d + 32); Fq1::norm(acc.ZZ).store(d + 64); Fq1::norm(acc.ZZZ).store(d + 96);
}
// canonical value of a 14-limb lazy element (host): value mod p as 6 x u64
static void canon(const uint32_t* l, uint64_t out[6]) {
  Fq1 a = Fq1::one();                       // (Fp::load takes 16-byte aligned addresses: the limbs are copied one by one here)
  for (int k = 0; k < 14; k++) a.l[k] = l[k];
  a.to_canonical(out);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  const int slots_max = cus * 8 * 256;
  std::vector<uint32_t> h_in((size_t)slots_max * 6 * 32, 0u);
  // inputs: small multiples of one in Montgomery form per limb - any field elements do (the formulas are run as arithmetic, not on curve points)
  uint64_t st = 88172645463325252ull;
  Fq1 one = Fq1::one();
  for (size_t i = 0; i < h_in.size(); i += 16) {                  // 14 limbs of 28 bits in 16 words (Fp::WORDS)
    for (int k = 0; k < 14; k++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h_in[i + k] = (uint32_t)(st & 0x0FFFFFFF); }
    h_in[i + 13] &= 0xFFF;                        // below p: 377 - 13 * 28 = 13 bits in the top limb (p's is 0x1ae3)
  }
  (void)one;
  fprintf(stderr, "inputs ready\n");
  uint32_t *d_in, *d_o1, *d_o2;
  CK(hipMalloc(&d_in, h_in.size() * 4)); CK(hipMalloc(&d_o1, (size_t)slots_max * 4 * 32 * 4)); CK(hipMalloc(&d_o2, (size_t)slots_max * 4 * 32 * 4));
  CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
  // correctness: nothing (0), one product (-1), 1 and 3 additions, 4096 slots
  for (int its : {0, -1, 1, 3}) {
    hipLaunchKernelGGL(k_single, dim3(16), dim3(256), 0, 0, d_in, d_o1, its);
    hipLaunchKernelGGL(k_pair, dim3(32), dim3(256), 0, 0, d_in, d_o2, its);
    CK(hipDeviceSynchronize());
    fprintf(stderr, "kernels ran\n");
    std::vector<uint32_t> a(4096 * 4 * 32), b(4096 * 4 * 32);
    CK(hipMemcpy(a.data(), d_o1, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_o2, b.size() * 4, hipMemcpyDeviceToHost));
    fprintf(stderr, "copied back\n");
    size_t bad = 0;
    for (size_t e = 0; e < a.size(); e += 16) { uint64_t x[6], y[6]; canon(&a[e], x); canon(&b[e], y); for (int k = 0; k < 6; k++) if (x[k] != y[k]) { bad++; break; } }
    printf("pair form vs one-lane form, mode %d: %zu of %zu field elements differ\n", its, bad, a.size() / 16);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 400;
  for (int bpc : {4, 8}) {
    const int blocks = cus * bpc;
    float ms1, ms2;
    fprintf(stderr, "bpc %d single\n", bpc);
    hipLaunchKernelGGL(k_single, dim3(blocks), dim3(256), 0, 0, d_in, d_o1, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_single, dim3(blocks), dim3(256), 0, 0, d_in, d_o1, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms1, e0, e1));
    fprintf(stderr, "bpc %d pair\n", bpc);
    hipLaunchKernelGGL(k_pair, dim3(blocks), dim3(256), 0, 0, d_in, d_o2, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_pair, dim3(blocks), dim3(256), 0, 0, d_in, d_o2, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms2, e0, e1));
    const double adds1 = (double)blocks * 256 * iters, adds2 = adds1 / 2;
    printf("blocks/CU %d: one lane per addition %8.3f ms  %7.2f M madd/s | lane pair per addition %8.3f ms  %7.2f M madd/s  (ratio %.3f)\n", bpc, ms1, adds1 / ms1 / 1e3, ms2,
           adds2 / ms2 / 1e3, (adds2 / ms2) / (adds1 / ms1));
  }
  return 0;
}
