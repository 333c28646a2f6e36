// Kernels of the lane-parallel BLS12-377 pairing (pairing_lanes.h): 64-thread blocks, 21 groups of three lanes each (lane 63
// idles).  Two translation units define them:
//   unit_pairing_lm.hip (CELO_LANES_DEFINE_MILLER): Miller loops (per pair, and per product with a shared accumulator), GT products
//   unit_pairing_lf.hip (CELO_LANES_DEFINE_FE):     final exponentiation
#pragma once
#include "pairing.h"

namespace celo {
#ifndef LANES_OCC
#define LANES_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
typedef QPairing377<QTri377> QPair;
typedef QTower<QTri377> QTow;
constexpr int LANES_GROUPS = QTri377::GROUPS_PER_WAVE;   // groups (pairings / products) per 64-thread block
// index of this lane's group among all groups of the grid, or -1 for the idle 64th lane
__device__ __forceinline__ int lanes_group_index() { const int g = QTri377::group(); return g >= LANES_GROUPS ? -1 : (int)blockIdx.x * LANES_GROUPS + g; }
__device__ __forceinline__ QTow::E12 lanes_load(const uint32_t* p) {
  const int j = QTri377::lane();
  return {Fq2::load(p + j * Fq2::WORDS), Fq2::load(p + (3 + j) * Fq2::WORDS)};
}
__device__ __forceinline__ void lanes_store(uint32_t* p, const QTow::E12& f) {
  const int j = QTri377::lane();
  f.a.store(p + j * Fq2::WORDS);
  f.b.store(p + (3 + j) * Fq2::WORDS);
}

#if defined(CELO_LANES_DEFINE_MILLER)
__global__ void __launch_bounds__(64) LANES_OCC k_miller_lanes(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                               const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                               uint32_t* __restrict__ f_out, uint32_t n) {
  const int gi = lanes_group_index();
  if (gi < 0 || (uint32_t)gi >= n) return;
  const uint32_t i = (uint32_t)gi;
  const Fq px = Fq::from_ark(g1 + (size_t)i * 12), py = Fq::from_ark(g1 + (size_t)i * 12 + 6);
  const Fq2 Qc = Fq2::from_ark(g2 + (size_t)i * 24 + (QTri377::lane() & 1) * 12);   // lanes 0, 2: Q.x; lane 1: Q.y
  QTow::E12 f = QPair::miller(px, py, Qc);
  if ((inf1 && inf1[i]) || (inf2 && inf2[i])) f = QTow::one12();
  lanes_store(f_out + (size_t)i * FQ12_WORDS, f);
}
// one group per PRODUCT of <= 4 pairs, shared accumulator; pairs with a point at infinity are left out (they contribute 1)
__global__ void __launch_bounds__(64) LANES_OCC k_miller_product_lanes(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                                       const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                                       const uint32_t* __restrict__ offsets, uint32_t* __restrict__ prod, uint32_t m) {
  const int gi = lanes_group_index();
  if (gi < 0 || (uint32_t)gi >= m) return;
  const uint32_t lo = offsets[gi], hi = offsets[gi + 1];
  Fq px[4], py[4];
  Fq2 Qc[4];
  int k = 0;
  for (uint32_t i = lo; i < hi && k < 4; i++) {
    if ((inf1 && inf1[i]) || (inf2 && inf2[i])) continue;
    px[k] = Fq::from_ark(g1 + (size_t)i * 12); py[k] = Fq::from_ark(g1 + (size_t)i * 12 + 6);
    Qc[k] = Fq2::from_ark(g2 + (size_t)i * 24 + (QTri377::lane() & 1) * 12);
    k++;
  }
  lanes_store(prod + (size_t)gi * FQ12_WORDS, QPair::miller_multi<4>(k, px, py, Qc));
}
__global__ void __launch_bounds__(64) LANES_OCC k_gt_product_lanes(const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ offsets,
                                                                   uint32_t* __restrict__ prod, uint32_t m) {
  const int gi = lanes_group_index();
  if (gi < 0 || (uint32_t)gi >= m) return;
  const uint32_t lo = offsets[gi], hi = offsets[gi + 1];
  QTow::E12 acc = QTow::one12();
  for (uint32_t k = lo; k < hi; k++) {
    QTow::E12 v = lanes_load(f_in + (size_t)k * FQ12_WORDS);
    acc = (k == lo) ? v : QTow::mul12(acc, v);
  }
  lanes_store(prod + (size_t)gi * FQ12_WORDS, acc);
}
__global__ void __launch_bounds__(64) LANES_OCC k_gt_tree_lanes(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_in) {
  const int gi = lanes_group_index();
  const uint32_t n_out = (n_in + 1) / 2;
  if (gi < 0 || (uint32_t)gi >= n_out) return;
  const uint32_t t = (uint32_t)gi;
  QTow::E12 a = lanes_load(in + (size_t)(2 * t) * FQ12_WORDS);
  if (2 * t + 1 < n_in) a = QTow::mul12(a, lanes_load(in + (size_t)(2 * t + 1) * FQ12_WORDS));
  lanes_store(out + (size_t)t * FQ12_WORDS, a);
}
#endif

#if defined(CELO_LANES_DEFINE_FE)
__global__ void __launch_bounds__(64) LANES_OCC k_final_exp_lanes(const uint32_t* __restrict__ prod, uint8_t* __restrict__ is_one,
                                                                  uint64_t* __restrict__ gt_ark, uint32_t m, int do_final_exp) {
  const int gi = lanes_group_index();
  if (gi < 0 || (uint32_t)gi >= m) return;
  const uint32_t p = (uint32_t)gi;
  const int q = QTri377::lane();
  QTow::E12 r = lanes_load(prod + (size_t)p * FQ12_WORDS);
  if (do_final_exp) r = QPair::final_exponentiation(r);
  const bool one = QTow::is_one12(r);
  if (is_one && q == 0) is_one[p] = one ? 1 : 0;
  if (gt_ark) {
    r.a.to_ark(gt_ark + (size_t)p * 72 + 12 * q);
    r.b.to_ark(gt_ark + (size_t)p * 72 + 12 * (3 + q));
  }
}
#endif

}  // namespace celo
