// Kernels of the lane-parallel BLS12-377 pairing (pairing_quad.h).  Two translation units define them:
//   unit_pairing_qm.hip (CELO_QUAD_DEFINE_MILLER): Miller loop, GT products
//   unit_pairing_qf.hip (CELO_QUAD_DEFINE_FE):     final exponentiation
#pragma once
#include "pairing.h"

namespace celo {
#ifndef QUAD_OCC
#define QUAD_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
typedef QPairing377<QDev377> QPair;
typedef QTower<QDev377> QTow;
__device__ __forceinline__ QTow::E12 quad_load(const uint32_t* p) {
  const int q = threadIdx.x & 3, j = q < 3 ? q : 0;
  return {Fq2::load(p + j * Fq2::WORDS), Fq2::load(p + (3 + j) * Fq2::WORDS)};
}
__device__ __forceinline__ void quad_store(uint32_t* p, const QTow::E12& f) {
  const int q = threadIdx.x & 3;
  if (q < 3) { f.a.store(p + q * Fq2::WORDS); f.b.store(p + (3 + q) * Fq2::WORDS); }
}

#if defined(CELO_QUAD_DEFINE_MILLER)
__global__ void __launch_bounds__(64) QUAD_OCC k_miller_quad(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                    const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                    uint32_t* __restrict__ f_out, uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 2;
  if (i >= n) return;
  const Fq px = Fq::from_ark(g1 + (size_t)i * 12), py = Fq::from_ark(g1 + (size_t)i * 12 + 6);
  const Fq2 Qc = Fq2::from_ark(g2 + (size_t)i * 24 + (t & 1) * 12);   // lanes 0, 2: Q.x; lanes 1, 3: Q.y
  QTow::E12 f = QPair::miller(px, py, Qc);
  if ((inf1 && inf1[i]) || (inf2 && inf2[i])) f = QTow::one12();
  quad_store(f_out + (size_t)i * FQ12_WORDS, f);
}
__global__ void __launch_bounds__(64) QUAD_OCC k_gt_product_quad(const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ offsets,
                                                        uint32_t* __restrict__ prod, uint32_t m) {
  const uint32_t p = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (p >= m) return;
  const uint32_t lo = offsets[p], hi = offsets[p + 1];
  QTow::E12 acc = QTow::one12();
  for (uint32_t k = lo; k < hi; k++) {
    QTow::E12 v = quad_load(f_in + (size_t)k * FQ12_WORDS);
    acc = (k == lo) ? v : QTow::mul12(acc, v);
  }
  quad_store(prod + (size_t)p * FQ12_WORDS, acc);
}
__global__ void __launch_bounds__(64) QUAD_OCC k_gt_tree_quad(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n_in) {
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const uint32_t n_out = (n_in + 1) / 2;
  if (t >= n_out) return;
  QTow::E12 a = quad_load(in + (size_t)(2 * t) * FQ12_WORDS);
  if (2 * t + 1 < n_in) a = QTow::mul12(a, quad_load(in + (size_t)(2 * t + 1) * FQ12_WORDS));
  quad_store(out + (size_t)t * FQ12_WORDS, a);
}
#endif

#if defined(CELO_QUAD_DEFINE_FE)
__global__ void __launch_bounds__(64) QUAD_OCC k_final_exp_quad(const uint32_t* __restrict__ prod, uint8_t* __restrict__ is_one,
                                                       uint64_t* __restrict__ gt_ark, uint32_t m, int do_final_exp) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, p = t >> 2;
  const int q = t & 3;
  if (p >= m) return;
  QTow::E12 r = quad_load(prod + (size_t)p * FQ12_WORDS);
  if (do_final_exp) r = QPair::final_exponentiation(r);
  const bool one = QTow::is_one12(r);
  if (is_one && q == 0) is_one[p] = one ? 1 : 0;
  if (gt_ark && q < 3) {
    r.a.to_ark(gt_ark + (size_t)p * 72 + 12 * q);
    r.b.to_ark(gt_ark + (size_t)p * 72 + 12 * (3 + q));
  }
}

#endif

}  // namespace celo
