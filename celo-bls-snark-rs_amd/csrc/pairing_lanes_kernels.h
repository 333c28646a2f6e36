// Kernels of the lane-parallel pairings (pairing_lanes.h): 64-thread blocks, 21 groups of three lanes each (lane 63 idles).
// The kernels are templates over a curve policy LP; four translation units instantiate them and define the launchers that
// PairingEngine (pairing.h) calls, so that the engine's own units stay small:
//   unit_pairing_lm.hip    LP377: Miller loops (per pair / per product with a shared accumulator), GT products
//   unit_pairing_lf.hip    LP377: final exponentiation
//   unit_pairing761_lm.hip LP761: Miller loops, GT products
//   unit_pairing761_lf.hip LP761: final exponentiation
#pragma once
#include "pairing.h"

namespace celo {
#ifndef LANES_OCC
#define LANES_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
// Curve / layout policies.  What a kernel needs from one: GROUPS (lane groups per 64-thread block), the backend QB and the
// pairing Pair built on it, and how a lane loads its share of P, Q and of a GT value (device layout: 6 coefficients of
// BP::WORDS words each, tower order).
struct LP377 {            // BLS12-377, three lanes per pairing: lane j holds the Fq2 coefficients j and 3 + j
  typedef Base377 BP;
  typedef QTri377 QB;
  typedef QPairing377<QTri377> Pair;
  typedef QTower<QTri377> Tow;
  typedef Fq F;
  static constexpr int G1W = 12, G2W = 24, GROUPS = 21;       // u64 per affine point (arkworks layout)
  __device__ __forceinline__ static F load_p(const uint64_t* g1, int coord) { return Fq::from_ark(g1 + coord * 6); }
  __device__ __forceinline__ static QB::V load_q(const uint64_t* g2) { return Fq2::from_ark(g2 + (QB::lane() & 1) * 12); }   // lanes 0, 2: Q.x; lane 1: Q.y
  __device__ __forceinline__ static Tow::E12 load12(const uint32_t* p) { const int j = QB::lane(); return {Fq2::load(p + j * 32), Fq2::load(p + (3 + j) * 32)}; }
  __device__ __forceinline__ static void store12(uint32_t* p, const Tow::E12& f) { const int j = QB::lane(); f.a.store(p + j * 32); f.b.store(p + (3 + j) * 32); }
  __device__ __forceinline__ static void to_ark12(const Tow::E12& f, uint64_t* gt) { const int j = QB::lane(); f.a.to_ark(gt + 12 * j); f.b.to_ark(gt + 12 * (3 + j)); }
  __device__ __forceinline__ static bool writer() { return QB::lane() == 0; }
  __device__ __forceinline__ static QB::V load_v(const uint32_t* p) { return Fq2::load(p); }          // this lane's share of one Fq2 at p
  __device__ __forceinline__ static void to_ark_v(const QB::V& v, uint64_t* o) { v.to_ark(o); }
};
struct LPH377 {           // BLS12-377, six lanes per pairing: lane 2 j + h holds half h of the Fq2 coefficients j and 3 + j
  typedef Base377 BP;
  typedef QHex377 QB;
  typedef QPairing377<QHex377> Pair;
  typedef QTower<QHex377> Tow;
  typedef Fq F;
  static constexpr int G1W = 12, G2W = 24, GROUPS = 10;
  __device__ __forceinline__ static F load_p(const uint64_t* g1, int coord) { return Fq::from_ark(g1 + coord * 6); }
  __device__ __forceinline__ static QB::V load_q(const uint64_t* g2) { return Fq::from_ark(g2 + (QB::lane() & 1) * 12 + QB::hsel() * 6); }
  __device__ __forceinline__ static Tow::E12 load12(const uint32_t* p) {
    const int j = QB::lane(), h = QB::hsel();
    return {Fq::load(p + j * 32 + h * 16), Fq::load(p + (3 + j) * 32 + h * 16)};
  }
  __device__ __forceinline__ static void store12(uint32_t* p, const Tow::E12& f) {
    const int j = QB::lane(), h = QB::hsel();
    f.a.store(p + j * 32 + h * 16); f.b.store(p + (3 + j) * 32 + h * 16);
  }
  __device__ __forceinline__ static void to_ark12(const Tow::E12& f, uint64_t* gt) {
    const int j = QB::lane(), h = QB::hsel();
    f.a.to_ark(gt + 12 * j + 6 * h); f.b.to_ark(gt + 12 * (3 + j) + 6 * h);
  }
  __device__ __forceinline__ static bool writer() { return QB::sub() == 0; }
  __device__ __forceinline__ static QB::V load_v(const uint32_t* p) { return Fq::load(p + QB::hsel() * 16); }
  __device__ __forceinline__ static void to_ark_v(const QB::V& v, uint64_t* o) { v.to_ark(o + QB::hsel() * 6); }
};
struct LP761 {            // BW6-761, three lanes per pairing: lane j holds the Fq coefficients j and 3 + j
  typedef Base761 BP;
  typedef QTri761 QB;
  typedef QPairing761<QTri761> Pair;
  typedef QTower<QTri761> Tow;
  typedef Fw F;
  static constexpr int G1W = 24, G2W = 24, GROUPS = 21;
  __device__ __forceinline__ static F load_p(const uint64_t* g1, int coord) { return Fw::from_ark(g1 + coord * 12); }
  __device__ __forceinline__ static QB::V load_q(const uint64_t* g2) { return Fw::from_ark(g2 + (QB::lane() & 1) * 12); }
  __device__ __forceinline__ static Tow::E12 load12(const uint32_t* p) { const int j = QB::lane(); return {Fw::load(p + j * 28), Fw::load(p + (3 + j) * 28)}; }
  __device__ __forceinline__ static void store12(uint32_t* p, const Tow::E12& f) { const int j = QB::lane(); f.a.store(p + j * 28); f.b.store(p + (3 + j) * 28); }
  __device__ __forceinline__ static void to_ark12(const Tow::E12& f, uint64_t* gt) { const int j = QB::lane(); f.a.to_ark(gt + 12 * j); f.b.to_ark(gt + 12 * (3 + j)); }
  __device__ __forceinline__ static bool writer() { return QB::lane() == 0; }
};
// index of this lane's group among all groups of the grid, or -1 for the idle lanes at the end of the wave
template <class LP> __device__ __forceinline__ int lanes_group_index() {
  const int g = LP::QB::group();
  return g >= LP::GROUPS ? -1 : (int)blockIdx.x * LP::GROUPS + g;
}
template <class LP> constexpr int lanes_gt_words() { return 6 * LP::BP::WORDS; }

template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_miller_lanes(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                               const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                               uint32_t* __restrict__ f_out, uint32_t n) {
  typedef typename LP::Tow Tow;
  const int gi = lanes_group_index<LP>();
  if (gi < 0 || (uint32_t)gi >= n) return;
  const uint32_t i = (uint32_t)gi;
  const typename LP::F px = LP::load_p(g1 + (size_t)i * LP::G1W, 0), py = LP::load_p(g1 + (size_t)i * LP::G1W, 1);
  const typename LP::QB::V Qc = LP::load_q(g2 + (size_t)i * LP::G2W);
  typename Tow::E12 f = LP::Pair::miller(px, py, Qc);
  if ((inf1 && inf1[i]) || (inf2 && inf2[i])) f = Tow::one12();
  LP::store12(f_out + (size_t)i * lanes_gt_words<LP>(), f);
}
// one group per PRODUCT of <= 4 pairs, shared accumulator; pairs with a point at infinity are left out (they contribute 1)
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_miller_product_lanes(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                                       const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                                       const uint32_t* __restrict__ offsets, uint32_t* __restrict__ prod, uint32_t m) {
  const int gi = lanes_group_index<LP>();
  if (gi < 0 || (uint32_t)gi >= m) return;
  const uint32_t lo = offsets[gi], hi = offsets[gi + 1];
  typename LP::F px[4], py[4];
  typename LP::QB::V Qc[4];
  int k = 0;
  for (uint32_t i = lo; i < hi && k < 4; i++) {
    if ((inf1 && inf1[i]) || (inf2 && inf2[i])) continue;
    px[k] = LP::load_p(g1 + (size_t)i * LP::G1W, 0); py[k] = LP::load_p(g1 + (size_t)i * LP::G1W, 1);
    Qc[k] = LP::load_q(g2 + (size_t)i * LP::G2W);
    k++;
  }
  LP::store12(prod + (size_t)gi * lanes_gt_words<LP>(), LP::Pair::template miller_multi<4>(k, px, py, Qc));
}
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_gt_product_lanes(const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ offsets,
                                                                   uint32_t* __restrict__ prod, uint32_t m) {
  typedef typename LP::Tow Tow;
  const int gi = lanes_group_index<LP>();
  if (gi < 0 || (uint32_t)gi >= m) return;
  const uint32_t lo = offsets[gi], hi = offsets[gi + 1];
  typename Tow::E12 acc = Tow::one12();
  for (uint32_t k = lo; k < hi; k++) {
    typename Tow::E12 v = LP::load12(f_in + (size_t)k * lanes_gt_words<LP>());
    acc = (k == lo) ? v : Tow::mul12(acc, v);
  }
  LP::store12(prod + (size_t)gi * lanes_gt_words<LP>(), acc);
}
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_gt_tree_lanes(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_in) {
  typedef typename LP::Tow Tow;
  const int gi = lanes_group_index<LP>();
  const uint32_t n_out = (n_in + 1) / 2;
  if (gi < 0 || (uint32_t)gi >= n_out) return;
  const uint32_t t = (uint32_t)gi;
  typename Tow::E12 a = LP::load12(in + (size_t)(2 * t) * lanes_gt_words<LP>());
  if (2 * t + 1 < n_in) a = Tow::mul12(a, LP::load12(in + (size_t)(2 * t + 1) * lanes_gt_words<LP>()));
  LP::store12(out + (size_t)t * lanes_gt_words<LP>(), a);
}
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_final_exp_lanes(const uint32_t* __restrict__ prod, uint8_t* __restrict__ is_one,
                                                                  uint64_t* __restrict__ gt_ark, uint32_t m, int do_final_exp) {
  typedef typename LP::Tow Tow;
  const int gi = lanes_group_index<LP>();
  if (gi < 0 || (uint32_t)gi >= m) return;
  const uint32_t p = (uint32_t)gi;
  typename Tow::E12 r = LP::load12(prod + (size_t)p * lanes_gt_words<LP>());
  if (do_final_exp) r = LP::Pair::final_exponentiation(r);
  const bool one = Tow::is_one12(r);
  if (is_one && LP::writer()) is_one[p] = one ? 1 : 0;
  if (gt_ark) LP::to_ark12(r, gt_ark + (size_t)p * 72);
}

// launcher definitions (declared in pairing.h as LaneLaunch<CURVE>)
#define CELO_DEFINE_LANE_MILLER_LAUNCHERS(LL, LP)                                                                                        \
  void LL::miller(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, uint32_t* f, uint32_t n, hipStream_t s) { \
    hipLaunchKernelGGL((k_miller_lanes<LP>), dim3((n + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, g1, i1, g2, i2, f, n);          \
  }                                                                                                                                       \
  void LL::miller_product(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off,              \
                          uint32_t* prod, uint32_t m, hipStream_t s) {                                                                    \
    hipLaunchKernelGGL((k_miller_product_lanes<LP>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, g1, i1, g2, i2, off,    \
                       prod, m);                                                                                                          \
  }                                                                                                                                       \
  void LL::gt_product(const uint32_t* f, const uint32_t* off, uint32_t* prod, uint32_t m, hipStream_t s) {                               \
    hipLaunchKernelGGL((k_gt_product_lanes<LP>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, f, off, prod, m);           \
  }                                                                                                                                       \
  void LL::gt_tree(const uint32_t* in, uint32_t* out, uint32_t n_in, hipStream_t s) {                                                     \
    const uint32_t n_out = (n_in + 1) / 2;                                                                                                \
    hipLaunchKernelGGL((k_gt_tree_lanes<LP>), dim3((n_out + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, in, out, n_in);            \
  }
#define CELO_DEFINE_LANE_FE_LAUNCHER(LL, LP)                                                                                              \
  void LL::final_exp(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, int do_fe, hipStream_t s) {                         \
    hipLaunchKernelGGL((k_final_exp_lanes<LP>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, prod, is_one, gt, m, do_fe); \
  }

}  // namespace celo
