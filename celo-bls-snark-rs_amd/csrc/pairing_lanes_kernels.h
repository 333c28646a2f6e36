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
struct LP377 {
  typedef Base377 BP;
  typedef QTri377 QB;
  typedef QPairing377<QTri377> Pair;
  typedef Fq F;
  static constexpr int G1W = 12, G2W = 24;       // u64 per affine point (arkworks layout)
};
struct LP761 {
  typedef Base761 BP;
  typedef QTri761 QB;
  typedef QPairing761<QTri761> Pair;
  typedef Fw F;
  static constexpr int G1W = 24, G2W = 24;
};
constexpr int LANES_GROUPS = 21;   // groups (pairings / products) per 64-thread block
// index of this lane's group among all groups of the grid, or -1 for the idle 64th lane
template <class LP> __device__ __forceinline__ int lanes_group_index() {
  const int g = LP::QB::group();
  return g >= LANES_GROUPS ? -1 : (int)blockIdx.x * LANES_GROUPS + g;
}
template <class LP> __device__ __forceinline__ typename QTower<typename LP::QB>::E12 lanes_load(const uint32_t* p) {
  typedef typename LP::BP BP;
  const int j = LP::QB::lane();
  return {BP::load(p + j * BP::WORDS), BP::load(p + (3 + j) * BP::WORDS)};
}
template <class LP> __device__ __forceinline__ void lanes_store(uint32_t* p, const typename QTower<typename LP::QB>::E12& f) {
  typedef typename LP::BP BP;
  const int j = LP::QB::lane();
  BP::store(p + j * BP::WORDS, f.a);
  BP::store(p + (3 + j) * BP::WORDS, f.b);
}
template <class LP> constexpr int lanes_gt_words() { return 6 * LP::BP::WORDS; }

template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_miller_lanes(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                               const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                               uint32_t* __restrict__ f_out, uint32_t n) {
  typedef QTower<typename LP::QB> Tow;
  const int gi = lanes_group_index<LP>();
  if (gi < 0 || (uint32_t)gi >= n) return;
  const uint32_t i = (uint32_t)gi;
  const typename LP::F px = LP::F::from_ark(g1 + (size_t)i * LP::G1W), py = LP::F::from_ark(g1 + (size_t)i * LP::G1W + LP::G1W / 2);
  const typename LP::BP::T Qc = LP::BP::from_ark(g2 + (size_t)i * LP::G2W + (LP::QB::lane() & 1) * (LP::G2W / 2));   // lanes 0, 2: Q.x; lane 1: Q.y
  typename Tow::E12 f = LP::Pair::miller(px, py, Qc);
  if ((inf1 && inf1[i]) || (inf2 && inf2[i])) f = Tow::one12();
  lanes_store<LP>(f_out + (size_t)i * lanes_gt_words<LP>(), f);
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
  typename LP::BP::T Qc[4];
  int k = 0;
  for (uint32_t i = lo; i < hi && k < 4; i++) {
    if ((inf1 && inf1[i]) || (inf2 && inf2[i])) continue;
    px[k] = LP::F::from_ark(g1 + (size_t)i * LP::G1W); py[k] = LP::F::from_ark(g1 + (size_t)i * LP::G1W + LP::G1W / 2);
    Qc[k] = LP::BP::from_ark(g2 + (size_t)i * LP::G2W + (LP::QB::lane() & 1) * (LP::G2W / 2));
    k++;
  }
  lanes_store<LP>(prod + (size_t)gi * lanes_gt_words<LP>(), LP::Pair::template miller_multi<4>(k, px, py, Qc));
}
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_gt_product_lanes(const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ offsets,
                                                                   uint32_t* __restrict__ prod, uint32_t m) {
  typedef QTower<typename LP::QB> Tow;
  const int gi = lanes_group_index<LP>();
  if (gi < 0 || (uint32_t)gi >= m) return;
  const uint32_t lo = offsets[gi], hi = offsets[gi + 1];
  typename Tow::E12 acc = Tow::one12();
  for (uint32_t k = lo; k < hi; k++) {
    typename Tow::E12 v = lanes_load<LP>(f_in + (size_t)k * lanes_gt_words<LP>());
    acc = (k == lo) ? v : Tow::mul12(acc, v);
  }
  lanes_store<LP>(prod + (size_t)gi * lanes_gt_words<LP>(), acc);
}
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_gt_tree_lanes(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_in) {
  typedef QTower<typename LP::QB> Tow;
  const int gi = lanes_group_index<LP>();
  const uint32_t n_out = (n_in + 1) / 2;
  if (gi < 0 || (uint32_t)gi >= n_out) return;
  const uint32_t t = (uint32_t)gi;
  typename Tow::E12 a = lanes_load<LP>(in + (size_t)(2 * t) * lanes_gt_words<LP>());
  if (2 * t + 1 < n_in) a = Tow::mul12(a, lanes_load<LP>(in + (size_t)(2 * t + 1) * lanes_gt_words<LP>()));
  lanes_store<LP>(out + (size_t)t * lanes_gt_words<LP>(), a);
}
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_final_exp_lanes(const uint32_t* __restrict__ prod, uint8_t* __restrict__ is_one,
                                                                  uint64_t* __restrict__ gt_ark, uint32_t m, int do_final_exp) {
  typedef QTower<typename LP::QB> Tow;
  const int gi = lanes_group_index<LP>();
  if (gi < 0 || (uint32_t)gi >= m) return;
  const uint32_t p = (uint32_t)gi;
  const int q = LP::QB::lane();
  typename Tow::E12 r = lanes_load<LP>(prod + (size_t)p * lanes_gt_words<LP>());
  if (do_final_exp) r = LP::Pair::final_exponentiation(r);
  const bool one = Tow::is_one12(r);
  if (is_one && q == 0) is_one[p] = one ? 1 : 0;
  if (gt_ark) {
    LP::BP::to_ark(r.a, gt_ark + (size_t)p * 72 + 12 * q);
    LP::BP::to_ark(r.b, gt_ark + (size_t)p * 72 + 12 * (3 + q));
  }
}

// launcher definitions (declared in pairing.h as LaneLaunch<CURVE>)
#define CELO_DEFINE_LANE_MILLER_LAUNCHERS(LL, LP)                                                                                        \
  void LL::miller(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, uint32_t* f, uint32_t n, hipStream_t s) { \
    hipLaunchKernelGGL((k_miller_lanes<LP>), dim3((n + LANES_GROUPS - 1) / LANES_GROUPS), dim3(64), 0, s, g1, i1, g2, i2, f, n);          \
  }                                                                                                                                       \
  void LL::miller_product(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off,              \
                          uint32_t* prod, uint32_t m, hipStream_t s) {                                                                    \
    hipLaunchKernelGGL((k_miller_product_lanes<LP>), dim3((m + LANES_GROUPS - 1) / LANES_GROUPS), dim3(64), 0, s, g1, i1, g2, i2, off,    \
                       prod, m);                                                                                                          \
  }                                                                                                                                       \
  void LL::gt_product(const uint32_t* f, const uint32_t* off, uint32_t* prod, uint32_t m, hipStream_t s) {                               \
    hipLaunchKernelGGL((k_gt_product_lanes<LP>), dim3((m + LANES_GROUPS - 1) / LANES_GROUPS), dim3(64), 0, s, f, off, prod, m);           \
  }                                                                                                                                       \
  void LL::gt_tree(const uint32_t* in, uint32_t* out, uint32_t n_in, hipStream_t s) {                                                     \
    const uint32_t n_out = (n_in + 1) / 2;                                                                                                \
    hipLaunchKernelGGL((k_gt_tree_lanes<LP>), dim3((n_out + LANES_GROUPS - 1) / LANES_GROUPS), dim3(64), 0, s, in, out, n_in);            \
  }
#define CELO_DEFINE_LANE_FE_LAUNCHER(LL, LP)                                                                                              \
  void LL::final_exp(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, int do_fe, hipStream_t s) {                         \
    hipLaunchKernelGGL((k_final_exp_lanes<LP>), dim3((m + LANES_GROUPS - 1) / LANES_GROUPS), dim3(64), 0, s, prod, is_one, gt, m, do_fe); \
  }

}  // namespace celo
