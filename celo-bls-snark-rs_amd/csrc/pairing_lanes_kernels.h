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

// ================================================================== hex kernels with the Fq12 accumulator in LDS
// The out-of-line tower routines take their Fq12 operands by reference, i.e. through per-lane private memory: ~0.3-0.7 KB per
// call and lane, and with 2048 waves resident that working set does not stay in L2 (rocprof r2: 30 GB of HBM traffic per
// 81920-product Miller launch, 33 GB per final-exponentiation launch, all of it argument passing).  With six lanes per pairing an
// Fq12 is 28 words per lane, so a wave's accumulator is 7 KB: it lives in LDS (two slots per wave: 14 KB, 8 waves per CU = 112 of
// the 160 KB) and the out-of-line routines take slot numbers; the G2 point travels in registers (14 words in, 14 out).
//   slot 0: f (Miller) / acc (exp_by_x)        slot 1: (P.x, P.y) of the current pair (Miller) / the base f (exp_by_x)
// Word-planar layout (word i of lane t at i * 64 + t): every ds_read/ds_write hits 64 distinct banks.
template <class LP> struct Slots {
  typedef typename LP::QB QB;
  typedef typename QB::V V;
  typedef typename LP::Tow Tow;
  typedef typename Tow::E12 E12;
  typedef typename LP::Pair Pair;
  static constexpr int NW = QB::NWORDS;
  static constexpr int SLOT_WORDS = 2 * NW * 64;
  static constexpr unsigned LDS_BYTES = 2 * SLOT_WORDS * 4;
  __device__ __forceinline__ static uint32_t* at(int slot, int which) {
    extern __shared__ uint32_t lanes_lds[];
    return lanes_lds + slot * SLOT_WORDS + which * NW * 64 + threadIdx.x;
  }
  __device__ __forceinline__ static V ldv(int slot, int which) {
    const uint32_t* p = at(slot, which);
    V r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.l[i] = p[i * 64];
    return r;
  }
  __device__ __forceinline__ static void stv(int slot, int which, const V& v) {
    uint32_t* p = at(slot, which);
#pragma unroll
    for (int i = 0; i < NW; i++) p[i * 64] = v.l[i];
  }
  __device__ __forceinline__ static E12 ld12(int slot) { return {ldv(slot, 0), ldv(slot, 1)}; }
  __device__ __forceinline__ static void st12(int slot, const E12& f) { stv(slot, 0, f.a); stv(slot, 1, f.b); }

  // LDS is the spill space: a fence keeps the compiler from hoisting the (cheap) LDS loads of a whole Fq12 above the work that
  // does not need it yet, or from keeping a value in registers that can be re-read where it is used again
  __device__ __forceinline__ static void fence() { asm volatile("" ::: "memory"); }
  // f *= line(P): mul_by_034 with the operands of each product group read from the slot where they are used
  // commit = false: everything is computed and nothing is stored (a lane group without a live pair in this round of a
  // multi-pair product keeps its accumulator; LDS stores are per lane, so leaving them out IS the undo)
  template <class PF> __device__ __forceinline__ static void ell_slot(int sf, PF ldp, const typename Pair::Line& l, bool commit) {
    typedef typename LP::QB QB;
    fence();
    const V t = QB::mul_fp(QB::template sel<0>(l.c0, l.c1), QB::pickf(ldp(1), ldp(0), ldp(0)));
    const V s0 = QB::template bcast<0>(t), s3 = QB::template bcast<1>(t), s4 = l.c2;
    fence();
    const V A = QB::mul(ldv(sf, 0), s0);
    fence();
    const V b = Tow::mul6_by_01(ldv(sf, 1), s3, s4);
    fence();
    const V e = Tow::mul6_by_01(QB::add(ldv(sf, 0), ldv(sf, 1)), QB::add(s0, s3), s4);
    const V nb = QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(e, A), b)), na = QB::lred(QB::add_l(A, Tow::template mul_by_gen_k_l<4>(b)));
    // branch-free: an uncommitted lane writes back what the slot holds (a divergent `if` around the stores made the register
    // allocator spill ~140 dwords per step in the surrounding 442 KB loop body; the select costs 56 instructions)
    stv(sf, 1, QB::choose(commit, nb, ldv(sf, 1)));
    stv(sf, 0, QB::choose(commit, na, ldv(sf, 0)));
    fence();
  }
  // f *= l_a(P_a) * l_b(P_b) with the two line values multiplied first (Pair::ell2 / Tow::mul_034_by_034: 8 product rounds where two
  // ell_slot are 10).  The dense half of the line product waits in half `wl` of slot `sl` while the three Fq6 products run; its sparse
  // half (b0, b1, 0) stays in 14 registers.  A pair that is not live contributes the line 1 = (1, 0, 0) - selected by mask arithmetic,
  // no divergent region.  lfa() fetches pair a's line when it is needed (a prepared line comes from global memory: loading it before
  // the point step of pair b would keep 42 registers alive across that step).
  template <class LFA, class PFA, class PFB>
  __device__ __forceinline__ static void ell2_slot(int sf, int sl, int wl, LFA lfa, PFA ldpa, bool live_a, const typename Pair::Line& lb, PFB ldpb, bool live_b) {
    typedef typename LP::QB QB;
    fence();
    V Lb;
    {
      const V zero = QB::zero(), one0 = QB::template sel<0>(QB::one(), zero);       // the line 1: s0 = 1 on lane 0, s3 = 0 on lane 1, s4 = 0
      const V tb = QB::choose(live_b, QB::mul_fp(QB::template sel<0>(lb.c0, lb.c1), QB::pickf(ldpb(1), ldpb(0), ldpb(0))), one0);
      const V t4 = QB::choose(live_b, lb.c2, zero);
      const typename Pair::Line la = lfa();
      const V ta = QB::choose(live_a, QB::mul_fp(QB::template sel<0>(la.c0, la.c1), QB::pickf(ldpa(1), ldpa(0), ldpa(0))), one0);
      const E12 L = Tow::mul_034_by_034(ta, QB::choose(live_a, la.c2, zero), tb, t4);
      stv(sl, wl, L.a);
      Lb = L.b;
    }
    fence();
    const V v0 = Tow::mul6(ldv(sf, 0), ldv(sl, wl));
    fence();
    const V v1 = Tow::mul6_by_01(ldv(sf, 1), QB::template bcast<0>(Lb), QB::template bcast<1>(Lb));
    fence();
    const V t = Tow::mul6(QB::add(ldv(sf, 0), ldv(sf, 1)), QB::add(ldv(sl, wl), Lb));
    stv(sf, 1, QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(t, v0), v1)));
    stv(sf, 0, QB::wred(QB::add_l(v0, Tow::template mul_by_gen_k_l<4>(v1))));
    fence();
  }
  // the same for two pairs that both walk their own point (k_miller_product_slots<LP, 2>): every half slot is taken, so the line product
  // stays in registers (28) across the three Fq6 products; ta / tb are the lines already evaluated at their P (lanes 0 | 1: s0 | s3),
  // taken right after each pair's point step so that no line triple stays alive across the other pair's step
  __device__ __forceinline__ static void ell2_regs(int sf, const V& ta, const V& a4, bool live_a, const V& tb, const V& b4, bool live_b) {
    typedef typename LP::QB QB;
    fence();
    const V zero = QB::zero(), one0 = QB::template sel<0>(QB::one(), zero);
    const E12 L = Tow::mul_034_by_034(QB::choose(live_a, ta, one0), QB::choose(live_a, a4, zero), QB::choose(live_b, tb, one0), QB::choose(live_b, b4, zero));
    fence();
    const V v0 = Tow::mul6(ldv(sf, 0), L.a);
    fence();
    const V v1 = Tow::mul6_by_01(ldv(sf, 1), QB::template bcast<0>(L.b), QB::template bcast<1>(L.b));
    fence();
    const V t = Tow::mul6(QB::add(ldv(sf, 0), ldv(sf, 1)), QB::add(L.a, L.b));
    stv(sf, 1, QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(t, v0), v1)));
    stv(sf, 0, QB::wred(QB::add_l(v0, Tow::template mul_by_gen_k_l<4>(v1))));
    fence();
  }
  // The routines of the inner loops are INLINED into them: an out-of-line call on gfx950 saves and restores the callee-saved
  // half of the ~250 live VGPRs (77 dwords each way per call, measured in the ISA) - the very private-memory traffic the slots
  // are there to remove.  Only whole loops (exp_loop: 63 squarings + 6 products) are out of line.
  __device__ __forceinline__ static void sqr12(int s) { st12(s, Tow::sqr12(ld12(s))); }
  template <class PF> __device__ __forceinline__ static V step_double(V Rc, int sf, PF ldp, bool commit = true) {
    typename Pair::Line l;
    Pair::double_step(Rc, l);
    ell_slot(sf, ldp, l, commit);
    return Rc;
  }
  template <class PF> __device__ __forceinline__ static V step_add(V Rc, V Qc, int sf, PF ldp, bool commit = true) {
    typename Pair::Line l;
    Pair::add_step(Rc, Qc, l);
    ell_slot(sf, ldp, l, commit);
    return Rc;
  }
  // P = (x, y) of up to MAXK pairs per group, stored ONCE per group behind the two Fq12 slots (the six lanes of a group read the
  // same words: a broadcast): word i of coordinate c of pair p of group g at ((p * 2 + c) * NW + i) * 16 + g
  // (the multi-pair kernel keeps f in slot 0 and the running points R_p in slots 1 ..: R_p = half (p & 1) of slot 1 + p / 2)
  static constexpr int product_slots(int maxk) { return 1 + (maxk + 1) / 2; }
  template <int NSLOTS> __device__ __forceinline__ static uint32_t* p_area() { extern __shared__ uint32_t lanes_lds[]; return lanes_lds + NSLOTS * SLOT_WORDS; }
  static constexpr unsigned product_lds_bytes(int maxk) { return (unsigned)(product_slots(maxk) * SLOT_WORDS * 4 + maxk * 2 * NW * 16 * 4); }
  template <int NSLOTS> __device__ __forceinline__ static void st_p(int pair, int coord, const V& v) {   // every lane of the group stores the same value
    uint32_t* q = p_area<NSLOTS>() + (pair * 2 + coord) * NW * 16 + QB::group();
#pragma unroll
    for (int i = 0; i < NW; i++) q[i * 16] = v.l[i];
  }
  template <int NSLOTS> __device__ __forceinline__ static V ld_p(int pair, int coord) {
    const uint32_t* q = p_area<NSLOTS>() + (pair * 2 + coord) * NW * 16 + QB::group();
    V r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.l[i] = q[i * 16];
    return r;
  }
  __device__ __forceinline__ static void mul12(int dst, int a, int b) {     // Tow::mul12_inl, operands re-read per Fq6 product
    typedef typename LP::QB QB;
    const V v0 = Tow::mul6(ldv(a, 0), ldv(b, 0));
    fence();
    const V v1 = Tow::mul6(ldv(a, 1), ldv(b, 1));
    fence();
    const V t = Tow::mul6(QB::add(ldv(a, 0), ldv(a, 1)), QB::add(ldv(b, 0), ldv(b, 1)));
    fence();
    stv(dst, 1, QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(t, v0), v1)));
    stv(dst, 0, QB::wred(QB::add_l(v0, Tow::template mul_by_gen_k_l<4>(v1))));
  }
  // word i of half `which` of slot `slot` as lane t holds it - any lane of the wave may read it (the slot is memory)
  __device__ __forceinline__ static V ldv_at(int slot, int which, int t) {
    extern __shared__ uint32_t lanes_lds[];
    const uint32_t* p = lanes_lds + slot * SLOT_WORDS + which * NW * 64 + t;
    V r;
#pragma unroll
    for (int i = 0; i < NW; i++) r.l[i] = p[i * 64];
    return r;
  }
  __device__ __forceinline__ static void cyclo(int s) {        // Tow::cyclotomic_sqr_inl with f re-read for the last step
    typedef typename LP::QB QB;
    V x, y;
#if defined(CELO_CYCLO_PERM)   // A/B switch: the round-3 form - both halves loaded as the lane owns them, four lane permutes to pair them up
    {
      const V fa = ldv(s, 0), fb = ldv(s, 1);
      x = QB::template sel<1>(QB::template perm<QP(0, 0, 1)>(fb), QB::template perm<QP(0, 0, 1)>(fa));
      y = QB::template sel<1>(QB::template perm<QP(1, 2, 2)>(fa), QB::template perm<QP(1, 1, 2)>(fb));
    }
#else
    {
      // late round 4: the Fq4 pairs (a0, b1), (b0, a2), (a1, b2) are read straight from the slot at the lane that holds them - the slot is
      // LDS, any lane's words are an address away: 28 ds_read instead of 28 ds_read + 56 ds_bpermute + 28 selects
      const int j = QB::lane();
      const int t0 = QB::group() >= LP::GROUPS ? QB::hsel() : (int)threadIdx.x - QB::sub() + QB::hsel();   // this group's tower lane 0, same half (the idle lanes 60 .. 63 read group 0's: in bounds)
      x = ldv_at(s, j == 1 ? 1 : 0, t0 + (j == 2 ? 2 : 0));                          // a0 | b0 | a1
      y = ldv_at(s, j == 1 ? 0 : 1, t0 + (j == 0 ? 2 : 4));                          // b1 | a2 | b2
    }
#endif
    fence();
    const V tmp = QB::mul(x, y);
    const V m = QB::mul(QB::add(x, y), QB::add(QB::template mul_nr_k_l<4>(y), x));
    const V o0 = QB::lred(QB::template sub_l<64>(QB::template sub_l<4>(m, tmp), QB::template mul_nr_k<4>(tmp)));
    static_assert(QB::LAZY, "the slot kernels run on the six-lane backend");
    const V ut = QB::template perm<QP(2, 0, 1)>(tmp);
    const V u = QB::dbl_l(QB::template sel<0>(QB::template mul_nr_k<4>(ut), ut));
    fence();
    const V za = QB::wred(QB::add_l(QB::dbl_l(QB::template sub_l<4>(o0, ldv(s, 0))), o0));
    const V zb = QB::wred(QB::add_l(QB::dbl_l(QB::add_l(u, ldv(s, 1))), u));
    stv(s, 0, za); stv(s, 1, zb);
  }
  __device__ __attribute__((noinline)) static void exp_loop() {        // slot 0 <- slot 1 ^ x
#pragma unroll 1
    for (int i = 62; i >= 0; i--) {
      cyclo(0);
      if ((T377::X >> i) & 1) mul12(0, 0, 1);
    }
  }
  __device__ __forceinline__ static E12 exp_by_x(const E12& f) {       // acc in slot 0, the base in slot 1
    st12(0, f); st12(1, f);
    exp_loop();
    return ld12(0);
  }
};

template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_miller_slots(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                               const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                               uint32_t* __restrict__ f_out, uint32_t n) {
  typedef Slots<LP> S;
  typedef typename LP::Tow Tow;
  typedef typename LP::QB QB;
  const int gi = lanes_group_index<LP>();
  const bool live = gi >= 0 && (uint32_t)gi < n;
  const uint32_t i = live ? (uint32_t)gi : 0;                 // idle groups walk pair 0 and store nothing (no early exit: LDS slots are per wave)
  S::stv(1, 0, LP::load_p(g1 + (size_t)i * LP::G1W, 0));
  S::stv(1, 1, LP::load_p(g1 + (size_t)i * LP::G1W, 1));
  const typename QB::V Qc = LP::load_q(g2 + (size_t)i * LP::G2W);
  typename QB::V Rc = QB::template sel<2>(QB::one(), Qc);
  S::st12(0, Tow::one12());
#pragma unroll 1
  for (int b = 62; b >= 0; b--) {
    S::sqr12(0);
    Rc = S::step_double(Rc, 0, [](int c) { return S::ldv(1, c); });
    if ((T377::X >> b) & 1) Rc = S::step_add(Rc, Qc, 0, [](int c) { return S::ldv(1, c); });
  }
  if (!live) return;
  typename Tow::E12 f = S::ld12(0);
  if ((inf1 && inf1[i]) || (inf2 && inf2[i])) f = Tow::one12();
  LP::store12(f_out + (size_t)i * lanes_gt_words<LP>(), f);
}
// one group per PRODUCT of <= MAXK pairs with a shared accumulator (ark-ec's multi-Miller loop: one squaring of f per iteration
// for all its pairs).  Nothing per pair lives in private memory: f in slot 0, the running points R_p in slots 1 .. (14 words per
// lane each), P_p in the per-group LDS area, Q_p re-read from global memory at the six addition steps.  MAXK is 2 for the verify /
// Batch::verify shapes (16 KB of LDS per wave: 8 waves per CU) and 4 otherwise (25 KB: 6 waves).  Pairs with a point at infinity are left out (they contribute 1); a group with fewer
// live pairs than MAXK computes the surplus steps without committing them.
// the p loops, unrolled by recursion (a `#pragma unroll` over bodies of this size is refused by the optimizer)
template <class LP, int MAXK, int P = 0> struct PairSteps {
  typedef Slots<LP> S;
  typedef typename LP::QB::V V;
  static constexpr int NS = S::product_slots(MAXK);
  __device__ __forceinline__ static void init(uint32_t (&idx)[MAXK], int k, uint32_t i, const uint64_t* g1, const uint64_t* g2) {
    if (P == k) {
      idx[P] = i;
      S::template st_p<NS>(P, 0, LP::load_p(g1 + (size_t)i * LP::G1W, 0));
      S::template st_p<NS>(P, 1, LP::load_p(g1 + (size_t)i * LP::G1W, 1));
      S::stv(1 + P / 2, P & 1, LP::QB::template sel<2>(LP::QB::one(), LP::load_q(g2 + (size_t)i * LP::G2W)));
    }
    PairSteps<LP, MAXK, P + 1>::init(idx, k, i, g1, g2);
  }
  __device__ __forceinline__ static void dbl(int k) {
    const V r = S::step_double(S::ldv(1 + P / 2, P & 1), 0, [](int c) { return S::template ld_p<NS>(P, c); }, P < k);
    S::stv(1 + P / 2, P & 1, r);                  // a surplus slot may advance its dummy point: only f must not change
    S::fence();
    PairSteps<LP, MAXK, P + 1>::dbl(k);
  }
  __device__ __forceinline__ static void add(const uint32_t (&idx)[MAXK], int k, const uint64_t* g2) {
    const V Qc = LP::load_q(g2 + (size_t)idx[P] * LP::G2W);
    const V r = S::step_add(S::ldv(1 + P / 2, P & 1), Qc, 0, [](int c) { return S::template ld_p<NS>(P, c); }, P < k);
    S::stv(1 + P / 2, P & 1, r);
    S::fence();
    PairSteps<LP, MAXK, P + 1>::add(idx, k, g2);
  }
};
template <class LP, int MAXK> struct PairSteps<LP, MAXK, MAXK> {
  __device__ __forceinline__ static void init(uint32_t (&)[MAXK], int, uint32_t, const uint64_t*, const uint64_t*) {}
  __device__ __forceinline__ static void dbl(int) {}
  __device__ __forceinline__ static void add(const uint32_t (&)[MAXK], int, const uint64_t*) {}
};
template <class LP, int MAXK>
__global__ void __launch_bounds__(64) LANES_OCC k_miller_product_slots(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                                       const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                                       const uint32_t* __restrict__ offsets, uint32_t* __restrict__ prod, uint32_t m) {
  typedef Slots<LP> S;
  typedef typename LP::Tow Tow;
  typedef typename LP::QB QB;
  const int gi = lanes_group_index<LP>();
  const bool live = gi >= 0 && (uint32_t)gi < m;
  const uint32_t lo = live ? offsets[gi] : 0, hi = live ? offsets[gi + 1] : 0;
  typedef PairSteps<LP, MAXK> PS;
  uint32_t idx[MAXK];                                        // the pair behind slot p (its Q is re-read at the addition steps)
  // every slot starts on pair 0 of the input (any valid pair); the live pairs of the product then take slots 0 .. k - 1 and the
  // surplus slots walk their dummy pair uncommitted
  for (int q = 0; q < MAXK; q++) PS::init(idx, q, 0, g1, g2);
  int k = 0;
  for (uint32_t i = lo; i < hi && k < MAXK; i++) {
    if ((inf1 && inf1[i]) || (inf2 && inf2[i])) continue;
    PS::init(idx, k, i, g1, g2);
    k++;
  }
  S::st12(0, Tow::one12());
#if !defined(CELO_MILLER_UNMERGED)
  if constexpr (MAXK == 2) {
    // round 4: two pairs = the two lines of a step multiplied with each other first (Slots::ell2_regs), as in k_miller_prepared_slots
    typedef typename LP::Pair Pair;
    typedef typename QB::V V;
    constexpr int NS = S::product_slots(2);
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
      S::sqr12(0);
      V ta, a4, tb, b4;
      {
        typename Pair::Line l;
        V R = S::ldv(1, 0);
        Pair::double_step(R, l);
        S::stv(1, 0, R);
        ta = Pair::eval_line(l, S::template ld_p<NS>(0, 0), S::template ld_p<NS>(0, 1)); a4 = l.c2;
      }
      S::fence();
      {
        typename Pair::Line l;
        V R = S::ldv(1, 1);
        Pair::double_step(R, l);
        S::stv(1, 1, R);
        tb = Pair::eval_line(l, S::template ld_p<NS>(1, 0), S::template ld_p<NS>(1, 1)); b4 = l.c2;
      }
      S::ell2_regs(0, ta, a4, k > 0, tb, b4, k > 1);
      if ((T377::X >> b) & 1) {
        {
          typename Pair::Line l;
          const V Qc = LP::load_q(g2 + (size_t)idx[0] * LP::G2W);
          V R = S::ldv(1, 0);
          Pair::add_step(R, Qc, l);
          S::stv(1, 0, R);
          ta = Pair::eval_line(l, S::template ld_p<NS>(0, 0), S::template ld_p<NS>(0, 1)); a4 = l.c2;
        }
        S::fence();
        {
          typename Pair::Line l;
          const V Qc = LP::load_q(g2 + (size_t)idx[1] * LP::G2W);
          V R = S::ldv(1, 1);
          Pair::add_step(R, Qc, l);
          S::stv(1, 1, R);
          tb = Pair::eval_line(l, S::template ld_p<NS>(1, 0), S::template ld_p<NS>(1, 1)); b4 = l.c2;
        }
        S::ell2_regs(0, ta, a4, k > 0, tb, b4, k > 1);
      }
    }
    if (live) LP::store12(prod + (size_t)gi * lanes_gt_words<LP>(), S::ld12(0));
    return;
  }
#endif
#pragma unroll 1
  for (int b = 62; b >= 0; b--) {
    S::sqr12(0);
    PS::dbl(k);
    if ((T377::X >> b) & 1) PS::add(idx, k, g2);
  }
  if (live) LP::store12(prod + (size_t)gi * lanes_gt_words<LP>(), S::ld12(0));
}
// ---- products whose FIRST pair has the same G2 point everywhere: e(S_b, -g2) * e(H_b, P_b) of every verify / Batch::verify
// (crates/bls-crypto/src/bls/public.rs:102, batch.rs:83).  The point steps of that pair do not depend on the product: its 69
// line-coefficient triples are computed ONCE (k_prepare_lines: what ark-ec keeps in a G2Prepared) and every product only
// evaluates them at its own S_b - a Miller step without the doubling / addition of R (about 40 % of a step).
constexpr int PREPARED_STEPS = 69;                      // 63 doubling + 6 addition steps of x = 0x8508c00000000001
constexpr int PREPARED_LINE_WORDS = 3 * 32;             // c0, c1, c2: an Fq2 each, device form (2 x 16 words)
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_prepare_lines(const uint64_t* __restrict__ q0_xy, uint32_t* __restrict__ lines) {
  typedef typename LP::QB QB;
  typedef typename LP::Pair Pair;
  if (QB::group() != 0) return;
  const typename QB::V Qc = LP::load_q(q0_xy);
  typename QB::V Rc = QB::template sel<2>(QB::one(), Qc);
  const int h = QB::hsel();
  const bool writer = QB::lane() == 0;                  // the coefficients are group-uniform: tower lane 0's two halves write them
  int step = 0;
  auto put = [&](const typename Pair::Line& l) {
    if (writer) {
      uint32_t* o = lines + (size_t)step * PREPARED_LINE_WORDS + h * 16;
      l.c0.store(o); l.c1.store(o + 32); l.c2.store(o + 64);
    }
    step++;
  };
#pragma unroll 1
  for (int b = 62; b >= 0; b--) {
    typename Pair::Line l;
    Pair::double_step(Rc, l);
    put(l);
    if ((T377::X >> b) & 1) { Pair::add_step(Rc, Qc, l); put(l); }
  }
}
// flag[0] stays 1 iff every non-empty product's first pair carries the G2 point of product 0's
template <class LP>     // (a template only so that the kernel may live in this header: it is included by several translation units)
__global__ void __launch_bounds__(256) k_first_q_same(const uint64_t* __restrict__ g2, const uint32_t* __restrict__ offsets, uint32_t m, uint32_t g2w,
                                                      uint32_t* __restrict__ flag) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m || offsets[p] == offsets[p + 1]) return;
  const uint64_t* a = g2 + (size_t)offsets[p] * g2w;
  const uint64_t* b = g2 + (size_t)offsets[0] * g2w;
  uint64_t diff = 0;
  for (uint32_t i = 0; i < g2w; i++) diff |= a[i] ^ b[i];
  if (diff || offsets[0] == offsets[1]) atomicAnd(flag, 0u);
}
// one group per product of <= 2 pairs; pair 0 on the prepared lines, pair 1 as in k_miller_product_slots
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_miller_prepared_slots(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                                        const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ lines,
                                                                        uint32_t* __restrict__ prod, uint32_t m) {
  typedef Slots<LP> S;
  typedef typename LP::Tow Tow;
  typedef typename LP::QB QB;
  typedef typename LP::Pair Pair;
  typedef typename QB::V V;
  constexpr int NS = S::product_slots(2);
  const int gi = lanes_group_index<LP>();
  const bool live = gi >= 0 && (uint32_t)gi < m;
  const uint32_t lo = live ? offsets[gi] : 0, hi = live ? offsets[gi + 1] : 0;
  const bool live_a = lo < hi && !((inf1 && inf1[lo]) || (inf2 && inf2[lo]));
  const bool live_b = lo + 1 < hi && !((inf1 && inf1[lo + 1]) || (inf2 && inf2[lo + 1]));
  const uint32_t ia = lo < hi ? lo : 0, ib = lo + 1 < hi ? lo + 1 : ia;      // dead slots walk a valid pair uncommitted
  S::template st_p<NS>(0, 0, LP::load_p(g1 + (size_t)ia * LP::G1W, 0));
  S::template st_p<NS>(0, 1, LP::load_p(g1 + (size_t)ia * LP::G1W, 1));
  S::template st_p<NS>(1, 0, LP::load_p(g1 + (size_t)ib * LP::G1W, 0));
  S::template st_p<NS>(1, 1, LP::load_p(g1 + (size_t)ib * LP::G1W, 1));
  S::stv(1, 0, QB::template sel<2>(QB::one(), LP::load_q(g2 + (size_t)ib * LP::G2W)));
  S::st12(0, Tow::one12());
  const int h = QB::hsel();
  int step = 0;
  auto line_at = [&](int st) {
    const uint32_t* o = lines + (size_t)st * PREPARED_LINE_WORDS + h * 16;
    typename Pair::Line l;
    l.c0 = V::load(o); l.c1 = V::load(o + 32); l.c2 = V::load(o + 64);
    return l;
  };
  auto ldp_a = [](int c) { return S::template ld_p<NS>(0, c); };
  auto ldp_b = [](int c) { return S::template ld_p<NS>(1, c); };
#if defined(CELO_MILLER_UNMERGED)   // A/B switch (tools/ab_pairing.sh): the round-3 loop, one sparse product per line
#pragma unroll 1
  for (int b = 62; b >= 0; b--) {
    S::sqr12(0);
    S::ell_slot(0, ldp_a, line_at(step), live_a);
    step++;
    {
      const V r = S::step_double(S::ldv(1, 0), 0, ldp_b, live_b);
      S::stv(1, 0, r);
      S::fence();
    }
    if ((T377::X >> b) & 1) {
      S::ell_slot(0, ldp_a, line_at(step), live_a);
      step++;
      const V Qc = LP::load_q(g2 + (size_t)ib * LP::G2W);
      const V r = S::step_add(S::ldv(1, 0), Qc, 0, ldp_b, live_b);
      S::stv(1, 0, r);
      S::fence();
    }
  }
#else
  // round 4: the two lines of a step are multiplied with each other first (Slots::ell2_slot); slot 1's second half - free in this
  // kernel, pair a has no running point - holds the dense half of the line product
#pragma unroll 1
  for (int b = 62; b >= 0; b--) {
    S::sqr12(0);
    {
      typename Pair::Line lb;
      {
        V Rb = S::ldv(1, 0);
        Pair::double_step(Rb, lb);
        S::stv(1, 0, Rb);
      }
      const int st = step;
      S::ell2_slot(0, 1, 1, [&]() { return line_at(st); }, ldp_a, live_a, lb, ldp_b, live_b);
      step++;
    }
    if ((T377::X >> b) & 1) {
      typename Pair::Line lb;
      {
        const V Qc = LP::load_q(g2 + (size_t)ib * LP::G2W);
        V Rb = S::ldv(1, 0);
        Pair::add_step(Rb, Qc, lb);
        S::stv(1, 0, Rb);
      }
      const int st = step;
      S::ell2_slot(0, 1, 1, [&]() { return line_at(st); }, ldp_a, live_a, lb, ldp_b, live_b);
      step++;
    }
  }
#endif
  if (live) LP::store12(prod + (size_t)gi * lanes_gt_words<LP>(), S::ld12(0));
}
// ---- the same product CUT IN TWO BY ITERATION RANGE, for a few thousand verify-shaped products (late round 4: config 3's 4096 verdicts,
// the exposed tail of Batch::verify).  With F_h the value after h iterations, F_63 = F_h^(2^(63 - h)) * G, G the same recurrence over the
// iterations h .. 62 started from 1 with the running point R_h.  Block 2 q computes F_h and squares on (role 0: 17 rounds per iteration,
// then 4); block 2 q + 1 walks the point to R_h without touching an accumulator and then runs G (role 1: 3.7 rounds, then 17): with
// h = 30 both are ~710 product rounds where one group per PAIR - what the engine runs below 16384 products - is ~900 per pair.  Ten
// products per block pair, two GT-shaped values per product at f_out[2 p], f_out[2 p + 1]: k_gt_product_lanes multiplies them.  Products of
// exactly two pairs each (the caller checks), first pair on the prepared lines.
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_miller_prepared_split_slots(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1,
                                                                              const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                                                                              const uint32_t* __restrict__ lines, uint32_t* __restrict__ f_out, uint32_t m, int H) {
  typedef Slots<LP> S;
  typedef typename LP::Tow Tow;
  typedef typename LP::QB QB;
  typedef typename LP::Pair Pair;
  typedef typename QB::V V;
  constexpr int NS = S::product_slots(2);
  const int role = (int)(blockIdx.x & 1);
  const int g = QB::group();
  const int gi = g >= LP::GROUPS ? -1 : (int)(blockIdx.x >> 1) * LP::GROUPS + g;
  const bool live = gi >= 0 && (uint32_t)gi < m;
  const uint32_t ia = live ? 2u * (uint32_t)gi : 0u, ib = ia + 1;                  // dead groups walk product 0 and store nothing
  const bool live_a = !((inf1 && inf1[ia]) || (inf2 && inf2[ia])), live_b = !((inf1 && inf1[ib]) || (inf2 && inf2[ib]));
  S::template st_p<NS>(0, 0, LP::load_p(g1 + (size_t)ia * LP::G1W, 0));
  S::template st_p<NS>(0, 1, LP::load_p(g1 + (size_t)ia * LP::G1W, 1));
  S::template st_p<NS>(1, 0, LP::load_p(g1 + (size_t)ib * LP::G1W, 0));
  S::template st_p<NS>(1, 1, LP::load_p(g1 + (size_t)ib * LP::G1W, 1));
  S::stv(1, 0, QB::template sel<2>(QB::one(), LP::load_q(g2 + (size_t)ib * LP::G2W)));
  S::st12(0, Tow::one12());
  const int h = QB::hsel();
  int step = 0;
  auto line_at = [&](int st) {
    const uint32_t* o = lines + (size_t)st * PREPARED_LINE_WORDS + h * 16;
    typename Pair::Line l;
    l.c0 = V::load(o); l.c1 = V::load(o + 32); l.c2 = V::load(o + 64);
    return l;
  };
  auto ldp_a = [](int c) { return S::template ld_p<NS>(0, c); };
  auto ldp_b = [](int c) { return S::template ld_p<NS>(1, c); };
#pragma unroll 1
  for (int it = 0; it < 63; it++) {
    const int b = 62 - it;
    const bool mine = role == 0 ? it < H : it >= H;          // this role multiplies lines into its accumulator in this iteration
    if (role == 1 || it < H) {                               // role 0 needs no point after its range; role 1 walks it from the start
      if (mine || role == 0) S::sqr12(0);
      typename Pair::Line lb;
      { V Rb = S::ldv(1, 0); Pair::double_step(Rb, lb); S::stv(1, 0, Rb); }
      if (mine) { const int st = step; S::ell2_slot(0, 1, 1, [&]() { return line_at(st); }, ldp_a, live_a, lb, ldp_b, live_b); }
      step++;
      if ((T377::X >> b) & 1) {
        { const V Qc = LP::load_q(g2 + (size_t)ib * LP::G2W); V Rb = S::ldv(1, 0); Pair::add_step(Rb, Qc, lb); S::stv(1, 0, Rb); }
        if (mine) { const int st = step; S::ell2_slot(0, 1, 1, [&]() { return line_at(st); }, ldp_a, live_a, lb, ldp_b, live_b); }
        step++;
      }
    } else {
      S::sqr12(0);                                           // role 0 beyond its range: F_h^(2^(63 - h))
    }
  }
  if (live) LP::store12(f_out + ((size_t)2 * gi + role) * lanes_gt_words<LP>(), S::ld12(0));
}
template <class LP>
__global__ void __launch_bounds__(64) LANES_OCC k_final_exp_slots(const uint32_t* __restrict__ prod, uint8_t* __restrict__ is_one,
                                                                  uint64_t* __restrict__ gt_ark, uint32_t m, int do_final_exp) {
  typedef Slots<LP> S;
  typedef typename LP::Tow Tow;
  const int gi = lanes_group_index<LP>();
  const bool live = gi >= 0 && (uint32_t)gi < m;
  const uint32_t p = live ? (uint32_t)gi : 0;
  typename Tow::E12 r = LP::load12(prod + (size_t)p * lanes_gt_words<LP>());
  if (do_final_exp) r = LP::Pair::final_exponentiation_t(r, [](const typename Tow::E12& v) { return S::exp_by_x(v); });
  const bool one = Tow::is_one12(r);
  if (!live) return;
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
  void LL::miller_product2(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off,             \
                           uint32_t* prod, uint32_t m, hipStream_t s) { miller_product(g1, i1, g2, i2, off, prod, m, s); }                \
  bool LL::has_prepared() { return false; }                                                                                               \
  void LL::first_q_same(const uint64_t*, const uint32_t*, uint32_t, uint32_t*, hipStream_t) {}                                            \
  void LL::prepare_lines(const uint64_t*, uint32_t*, hipStream_t) {}                                                                      \
  void LL::miller_prepared(const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint32_t*, const uint32_t*, uint32_t*, \
                           uint32_t, hipStream_t) {}                                                                                      \
  void LL::gt_product(const uint32_t* f, const uint32_t* off, uint32_t* prod, uint32_t m, hipStream_t s) {                               \
    hipLaunchKernelGGL((k_gt_product_lanes<LP>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, f, off, prod, m);           \
  }                                                                                                                                       \
  bool LL::has_split() { return false; }                                                                                                  \
  void LL::miller_prepared_split(const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint32_t*, uint32_t*, uint32_t, hipStream_t) {} \
  void LL::gt_tree(const uint32_t* in, uint32_t* out, uint32_t n_in, hipStream_t s) {                                                     \
    const uint32_t n_out = (n_in + 1) / 2;                                                                                                \
    hipLaunchKernelGGL((k_gt_tree_lanes<LP>), dim3((n_out + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, in, out, n_in);            \
  }
// the same launchers on the LDS-slot kernels (hex layout)
#define CELO_DEFINE_SLOT_MILLER_LAUNCHERS(LL, LP)                                                                                        \
  void LL::miller(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, uint32_t* f, uint32_t n, hipStream_t s) { \
    hipLaunchKernelGGL((k_miller_slots<LP>), dim3((n + LP::GROUPS - 1) / LP::GROUPS), dim3(64), Slots<LP>::LDS_BYTES, s, g1, i1, g2, i2, f, n); \
  }                                                                                                                                       \
  void LL::miller_product(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off,              \
                          uint32_t* prod, uint32_t m, hipStream_t s) {                                                                    \
    hipLaunchKernelGGL((k_miller_product_slots<LP, 4>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64),                               \
                       Slots<LP>::product_lds_bytes(4), s, g1, i1, g2, i2, off, prod, m);                                    \
  }                                                                                                                                       \
  void LL::miller_product2(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off,             \
                           uint32_t* prod, uint32_t m, hipStream_t s) {                                                                   \
    hipLaunchKernelGGL((k_miller_product_slots<LP, 2>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64),                               \
                       Slots<LP>::product_lds_bytes(2), s, g1, i1, g2, i2, off, prod, m);                                    \
  }                                                                                                                                       \
  bool LL::has_prepared() { return true; }                                                                                                \
  void LL::first_q_same(const uint64_t* g2, const uint32_t* off, uint32_t m, uint32_t* flag, hipStream_t s) {                             \
    hipLaunchKernelGGL((k_first_q_same<LP>), dim3((m + 255) / 256), dim3(256), 0, s, g2, off, m, (uint32_t)LP::G2W, flag);                      \
  }                                                                                                                                       \
  void LL::prepare_lines(const uint64_t* q0, uint32_t* lines, hipStream_t s) {                                                            \
    hipLaunchKernelGGL((k_prepare_lines<LP>), dim3(1), dim3(64), 0, s, q0, lines);                                                        \
  }                                                                                                                                       \
  void LL::miller_prepared(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off,             \
                           const uint32_t* lines, uint32_t* prod, uint32_t m, hipStream_t s) {                                            \
    hipLaunchKernelGGL((k_miller_prepared_slots<LP>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64),                                  \
                       Slots<LP>::product_lds_bytes(2), s, g1, i1, g2, i2, off, lines, prod, m);                                          \
  }                                                                                                                                       \
  void LL::gt_product(const uint32_t* f, const uint32_t* off, uint32_t* prod, uint32_t m, hipStream_t s) {                               \
    hipLaunchKernelGGL((k_gt_product_lanes<LP>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, f, off, prod, m);               \
  }                                                                                                                                       \
  bool LL::has_split() { return true; }                                                                                                   \
  void LL::miller_prepared_split(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* lines,     \
                                 uint32_t* f, uint32_t m, hipStream_t s) {                                                                \
    static const int cut = 30;   /* the cut h: flat between 26 and 34 (the tuning hook of round 4 is gone) */                  \
    hipLaunchKernelGGL((k_miller_prepared_split_slots<LP>), dim3(2 * ((m + LP::GROUPS - 1) / LP::GROUPS)), dim3(64),                      \
                       Slots<LP>::product_lds_bytes(2), s, g1, i1, g2, i2, lines, f, m, cut < 1 ? 1 : cut > 62 ? 62 : cut);               \
  }                                                                                                                                       \
  void LL::gt_tree(const uint32_t* in, uint32_t* out, uint32_t n_in, hipStream_t s) {                                                     \
    const uint32_t n_out = (n_in + 1) / 2;                                                                                                \
    hipLaunchKernelGGL((k_gt_tree_lanes<LP>), dim3((n_out + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, in, out, n_in);                \
  }
#define CELO_DEFINE_SLOT_FE_LAUNCHER(LL, LP)                                                                                              \
  void LL::final_exp(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, int do_fe, hipStream_t s) {                         \
    hipLaunchKernelGGL((k_final_exp_slots<LP>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64), Slots<LP>::LDS_BYTES, s, prod, is_one, gt, m, do_fe); \
  }
#define CELO_DEFINE_LANE_FE_LAUNCHER(LL, LP)                                                                                              \
  void LL::final_exp(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, int do_fe, hipStream_t s) {                         \
    hipLaunchKernelGGL((k_final_exp_lanes<LP>), dim3((m + LP::GROUPS - 1) / LP::GROUPS), dim3(64), 0, s, prod, is_one, gt, m, do_fe); \
  }

}  // namespace celo
