// Lane-parallel BLS12-377 pairing: ONE PAIRING PER GROUP OF THREE ADJACENT LANES (21 groups per wave64), state in registers.
//
// Why: the one-lane-per-pairing kernels of pairing.h keep an Fq12 accumulator (168 words) plus the G2 point, the line and
// the tower temporaries per lane; that does not fit 256 VGPRs, so every Fq2 product round-trips its operands through
// scratch and the kernels run at the scratch (L2 / Infinity Cache) bandwidth, not at the integer-VALU rate: two waves per
// SIMD take 2.3x the time of one (measured), i.e. no latency is being hidden, the memory pipe is simply full.
//
// Shape: an Fq12 element f = (a0 + a1 v + a2 v^2) + (b0 + b1 v + b2 v^2) w is spread over the three lanes of a group, lane j
// holding the Fq2 pair (a_j, b_j) = 56 words.  Every Fq6 product is Karatsuba ACROSS lanes: lane j forms its own product
// x_j*y_j and one cross product; operands and results travel by ds_bpermute_b32 (the LDS crossbar, no LDS storage; ~3 % of
// the instructions), so an Fq12 product is 6 dependent Fq2 products per lane instead of 18, a squaring 4 instead of 12,
// a sparse line product 5 instead of 13, a cyclotomic squaring 2 instead of 6.  The G2 point R = (X, Y, Z) lives one
// coordinate per lane; a doubling step is 3 three-lane product rounds (+ the twist-constant product as two Fq products),
// an addition step 5.  All three lanes work in every round.  (The first version used quads and DPP quad_perm with the
// fourth lane as an extra multiplier for the point arithmetic: 16 pairings per wave instead of 21, and ROCm 7.2's
// DPP-combine pass miscompiled the inlined addition step - see DESIGN.md section 5.)
//
// A product of k <= 4 pairs can run in ONE group with a shared accumulator (one squaring of f per iteration for all its
// pairs, as ark-ec's multi-Miller loop does): miller_multi.  The engine picks it when there are enough products to fill
// the chip and falls back to one group per pair + a GT product otherwise.
//
// Same field elements as ark-ec's bls12 engine at every step that is observable (Miller-loop value, GT value), so the
// outputs stay bit-identical with pairing.h and with the oracle.  Two local formula changes keep lanes uniform:
// h = 2YZ instead of (Y+Z)^2 - Y^2 - Z^2, and x/2 by an exact halving instead of a multiplication by 1/2.
//
// The algorithms are templates over a backend QB: QTri377 (device) and QHost377 (host, three explicit lanes) so the
// same code runs under the bounds-tracking host build (-DCELO_FP_TRACK, host_test.cpp) and against the oracle on the CPU.
// Replaces the arithmetic behind crates/bls-crypto/src/bls/public.rs:102 and signature.rs:149 (product_of_pairings).
#pragma once
#include "tower.h"

namespace celo {

// B' * x for the D-twist constant B' = (0, b1): (0 + b1 u)(x0 + x1 u) = -5 b1 x1 + b1 x0 u  (two Fq products)
HD Fq2 twist_b_times(const Fq2& x) {
  const Fq b1 = Fq::from_limbs(T377::TWIST_B_C1);
  const Fq2 n = Fq2::norm(x);
  const Fq p0 = Fq::mul(n.c0, b1), p1 = Fq::mul(n.c1, b1);
  const Fq p5 = Fq::norm(Fq::add(Fq::dbl(Fq::dbl(p1)), p1));          // 5 b1 x1, vb 10
  return {Fq::wred(Fq::norm(Fq::template neg<16, 1>(p5))), p0};
}

struct QHost377 : QHostT<Base377> {       // BLS12-377 extras: Fq scalars (P's coordinates), conjugation, the twist constant
  struct F { Fq v[NL]; };
  static V conj(const V& a) { return map1(a, [](const Fq2& x) { return Fq2{x.c0, Fq::wred(Fq::norm(Fq::neg<4, 1>(x.c1)))}; }); }
  static V mul_fp(const V& a, const F& k) { V r; for (int i = 0; i < NL; i++) r.v[i] = Fq2::mul_fp(Fq2::norm(a.v[i]), k.v[i]); return r; }
  static V twist_mul(const V& a) { return map1(a, [](const Fq2& x) { return twist_b_times(x); }); }
  static F pickf(const F& a0, const F& a1, const F& a2) { F r; r.v[0] = a0.v[0]; r.v[1] = a1.v[1]; r.v[2] = a2.v[2]; return r; }
  static V constant(const uint32_t* c0, const uint32_t* c1) { return uni(f2_from(c0, c1)); }
};
struct QHost761 : QHostT<Base761> {
  static V constant(const uint32_t* c) { return uni(Fw::from_limbs(c)); }
};

#if defined(__HIPCC__)
struct QTri377 : QTriT<Base377> {
  typedef Fq F;
  QDEV static V conj(const V& a) { return {a.c0, Fq::wred(Fq::norm(Fq::neg<4, 1>(a.c1)))}; }
  QDEV static V mul_fp(const V& a, const F& k) { return Fq2::mul_fp(Fq2::norm(a), k); }
  QDEV static V twist_mul(const V& a) { return twist_b_times(a); }
  QDEV static F pickf(const F& a0, const F& a1, const F& a2) {
    const int q = lane();
    const uint32_t m0 = lane_mask(q == 0), m1 = lane_mask(q == 1);
    F r;
#pragma unroll
    for (int i = 0; i < 14; i++) {
      const uint32_t t = (a1.l[i] & m1) | (a2.l[i] & ~m1);
      r.l[i] = (a0.l[i] & m0) | (t & ~m0);
    }
    return r;
  }
  QDEV static V constant(const uint32_t* c0, const uint32_t* c1) { return f2_from(c0, c1); }
};
struct QTri761 : QTriT<Base761> {
  QDEV static V constant(const uint32_t* c) { return Fw::from_limbs(c); }
};
#endif

// ================================================================== BLS12-377, SIX lanes per pairing ("hex" backends)
// The three-lane layout keeps an Fq2 (28 words) per value and lane; the tower routines hold 8-10 such values at once, which
// does not fit 256 VGPRs: every step spills ~2.3 KB per lane to scratch, and with 2048 waves resident the spill set (~330 MB)
// lives in HBM, not in a cache (rocprof: 124 GB of traffic per 86016-product launch).  Here the two halves (c0, c1) of every Fq2
// sit in two ADJACENT lanes: lane 2j + h of a group of six holds half h of tower lane j.  A value is 14 words, the whole live
// state of a Miller step fits the register file, and a lane executes half the instructions per pairing (half the latency of a
// lone verification).  An Fq2 product is ONE signed two-product Montgomery pass per lane (Fp::mul2s):
//     half 0:  a0 b0 + (-5 a1) b1          half 1:  a0 b1 + a1 b0
// after one exchange of the operands with the partner lane (lane ^ 1).  Multiplication by the non-residue u swaps the halves
// ((x0, x1) u = (-5 x1, x0)).  QTower / QPairing377 below run unchanged on these backends: same field elements at every step.
namespace hex {
// lane-local pieces; h = the half this lane holds, *o = the partner lane's value.  All host+device, bounds-tracked on the host.
// Every value of these backends is kept with normalised limbs (each operation below ends in a carry propagation or is a
// Montgomery product), so the products take their operands as they are: no carry pass per operand (that was 6 x 42 of the
// ~1400 instructions of a product half).  The bounds-tracking host build asserts it (Fp::mul2s: lb <= 1).
// Operand selection (round 3): the pass is X * b + Ys * bo with b, bo the lane's own and the partner's half of the second operand AS
// THEY ARE, and only the first operand's halves selected - X = own (half 0) / partner's (half 1), Ys = -5 * partner's (half 0) / own
// (half 1): 28 selects per product instead of 42.  (-5 x stays the compiler's v_mul_lo_u32, a quarter-rate instruction - 14 per product,
// 4 % of the Miller loop's issue slots.  Both ways around it were built - v_lshl_add_u32 + v_sub_u32 through inline assembly, and a
// 64-bit product kept opaque so that it becomes a v_mad_u64_u32 as in fp.h mont_digit - and both made the Miller kernel 35 % SLOWER
// (fourteen extra live registers or register pairs at the head of every product in a 248-register kernel: spills) and made
// k_miller_product_slots<LPH377, 4> return wrong values, i.e. tripped a code-generation problem on top: DESIGN.md section 5.)
// (A third form, an opaque zero register so that `z - 5 x` cannot be folded, compiles to v_mad_u64_u32 x, -5, z - no v_mul_lo_u32 left,
// same instruction count, results right - and measures THE SAME on one box, tools/ab_pairing.sh: Miller 18.8 vs 18.8 ms, final
// exponentiation 15.3 vs 15.3.  The 14 multiplies per product are not what the issue slots go to; left as plain C.)
HD uint32_t times5(uint32_t x) { return (x << 2) + x; }
HD Fq mul(const Fq& a, const Fq& b, const Fq& ao, const Fq& bo, int h) {
  TRK(assert(ao.lb <= 1 && a.lb <= 1);)
  int32_t cs[14];
  Fq X;
#pragma unroll
  for (int i = 0; i < 14; i++) {
    cs[i] = h ? (int32_t)a.l[i] : (int32_t)(0u - times5(ao.l[i]));
    X.l[i] = h ? ao.l[i] : a.l[i];
  }
  TRK(X.lb = 1; X.vb = h ? ao.vb : a.vb; assert((h ? a.vb : ao.vb) <= 64);)
  return Fq::mul2s(X, b, cs, bo);
}
// (x0 + x1 u) u = -5 x1 + x0 u: half 0 takes -5 * (partner), half 1 takes the partner as it is.  Needs vb(partner) <= 12.
HD Fq mul_nr(const Fq& xo, int h) {
  if (h) return xo;
  const Fq t5 = Fq::norm(Fq::add(Fq::dbl(Fq::dbl(xo)), xo));
  return Fq::norm(Fq::neg<64, 1>(t5));
}
// the same with one carry pass: -5 x = 5 (K p - x), for vb(partner) <= K (K = 4: products and weak-reduced values; 16: sums of a few)
template <int K> HD Fq mul_nr_k(const Fq& xo, int h) {
  if (h) return xo;
  const Fq n = Fq::neg<K, 1>(xo);
  return Fq::norm(Fq::add(Fq::dbl(Fq::dbl(n)), n));
}
// the same WITHOUT the carry pass (limbs up to 10 * 2^28): for results that go straight into an addition which is carried or weakly
// reduced right after (42 instructions less; the host build asserts the limb bound of every such sum)
template <int K> HD Fq mul_nr_k_l(const Fq& xo, int h) {
  if (h) return xo;
  const Fq n = Fq::neg<K, 1>(xo);
  return Fq::add(Fq::dbl(Fq::dbl(n)), n);
}
HD Fq conj(const Fq& x, int h) { return h ? Fq::wred(Fq::norm(Fq::neg<4, 1>(x))) : x; }
// B' x for the twist constant B' = (0, b1): (-5 b1 x1, b1 x0); p = b1 * (own half), po = the partner's p
HD Fq twist_own(const Fq& x) { return Fq::mul(x, Fq::from_limbs(T377::TWIST_B_C1)); }
HD Fq twist_fin(const Fq& po, int h) {
  if (h) return po;
  const Fq p5 = Fq::norm(Fq::add(Fq::dbl(Fq::dbl(po)), po));
  return Fq::wred(Fq::norm(Fq::neg<16, 1>(p5)));
}
// Fq2 inverse: n = x0^2 + 5 x1^2 from the two squares, then x0 / n and -x1 / n
HD Fq inv_norm(const Fq& s, const Fq& so, int h) {
  const Fq s1 = h ? s : so, s0 = h ? so : s;
  return Fq::norm(Fq::add(s0, Fq::add(Fq::dbl(Fq::dbl(s1)), s1)));
}
HD Fq inv_fin(const Fq& x, const Fq& ni, int h) {
  const Fq r = Fq::mul(Fq::norm(x), ni);
  return h ? Fq::norm(Fq::neg<4, 1>(r)) : r;
}
HD bool is_one_half(const Fq& a, int h) { return h ? a.is_zero_mod_p() : Fq::sub<64, 1>(Fq::norm(a), Fq::one()).is_zero_mod_p(); }
HD Fq add(const Fq& a, const Fq& b) { return Fq::norm(Fq::add(a, b)); }
HD Fq dbl(const Fq& a) { return Fq::norm(Fq::add(a, a)); }
HD Fq tpl(const Fq& a) { return Fq::norm(Fq::add(Fq::add(a, a), a)); }
template <int K> HD Fq sub(const Fq& a, const Fq& b) { return Fq::norm(Fq::sub<K, 1>(a, b)); }
template <int K> HD Fq neg(const Fq& a) { return Fq::norm(Fq::neg<K, 1>(a)); }
// lazy forms: no carry pass - the result goes to sub (as the minuend), wred or another lazy form only (limb bound asserted on the host)
#if defined(CELO_HEX_EAGER)   // A/B switch: every form carries
HD Fq add_l(const Fq& a, const Fq& b) { return Fq::norm(Fq::add(a, b)); }
HD Fq dbl_l(const Fq& a) { return Fq::norm(Fq::add(a, a)); }
template <int K> HD Fq sub_l(const Fq& a, const Fq& b) { return Fq::norm(Fq::sub<K, 1>(a, b)); }
#else
HD Fq add_l(const Fq& a, const Fq& b) { return Fq::add(a, b); }
HD Fq dbl_l(const Fq& a) { return Fq::add(a, a); }
template <int K> HD Fq sub_l(const Fq& a, const Fq& b) { return Fq::sub<K, 1>(a, b); }
#endif
}  // namespace hex

// host backend: six explicit lanes, index 2 j + h
struct QHostHex377 {
  static constexpr int NL = 3;
  struct V { Fq v[6]; };
  typedef V F;                                            // P's coordinates: the same Fq in every lane
  template <class Fn> static V map1(const V& a, Fn fn) { V r; for (int i = 0; i < 6; i++) r.v[i] = fn(a.v[i], i & 1); return r; }
  template <class Fn> static V map2(const V& a, const V& b, Fn fn) { V r; for (int i = 0; i < 6; i++) r.v[i] = fn(a.v[i], b.v[i]); return r; }
  static V swap(const V& a) { V r; for (int i = 0; i < 6; i++) r.v[i] = a.v[i ^ 1]; return r; }
  static V uni2(const Fq& c0, const Fq& c1) { V r; for (int i = 0; i < 6; i++) r.v[i] = (i & 1) ? c1 : c0; return r; }
  static V mul(const V& a, const V& b) { V r; for (int i = 0; i < 6; i++) r.v[i] = hex::mul(a.v[i], b.v[i], a.v[i ^ 1], b.v[i ^ 1], i & 1); return r; }
  static V add(const V& a, const V& b) { return map2(a, b, [](const Fq& x, const Fq& y) { return hex::add(x, y); }); }
  static V dbl(const V& a) { return map1(a, [](const Fq& x, int) { return hex::dbl(x); }); }
  static V tpl(const V& a) { return map1(a, [](const Fq& x, int) { return hex::tpl(x); }); }
  template <int K> static V sub(const V& a, const V& b) { return map2(a, b, [](const Fq& x, const Fq& y) { return hex::sub<K>(x, y); }); }
  template <int K> static V neg(const V& a) { return map1(a, [](const Fq& x, int) { return hex::neg<K>(x); }); }
  static V wred(const V& a) { return map1(a, [](const Fq& x, int) { return Fq::wred(x); }); }
  static V lred(const V& a) { return map1(a, [](const Fq& x, int) { return Fq::norm(x); }); }
  static V half(const V& a) { return map1(a, [](const Fq& x, int) { return Fq::half(x); }); }
  static V mul_nr(const V& a) { V r; for (int i = 0; i < 6; i++) r.v[i] = hex::mul_nr(a.v[i ^ 1], i & 1); return r; }
  static constexpr bool LAZY = true;
  template <int K> static V mul_nr_k(const V& a) { V r; for (int i = 0; i < 6; i++) r.v[i] = hex::mul_nr_k<K>(a.v[i ^ 1], i & 1); return r; }
  template <int K> static V mul_nr_k_l(const V& a) { V r; for (int i = 0; i < 6; i++) r.v[i] = hex::mul_nr_k_l<K>(a.v[i ^ 1], i & 1); return r; }
  static V add_l(const V& a, const V& b) { return map2(a, b, [](const Fq& x, const Fq& y) { return hex::add_l(x, y); }); }
  static V dbl_l(const V& a) { return map1(a, [](const Fq& x, int) { return hex::dbl_l(x); }); }
  template <int K> static V sub_l(const V& a, const V& b) { return map2(a, b, [](const Fq& x, const Fq& y) { return hex::sub_l<K>(x, y); }); }
  static V conj(const V& a) { return map1(a, [](const Fq& x, int h) { return hex::conj(x, h); }); }
  static V mul_fp(const V& a, const F& k) { return map2(a, k, [](const Fq& x, const Fq& y) { return Fq::mul(x, y); }); }
  static V twist_mul(const V& a) {
    V p = map1(a, [](const Fq& x, int) { return hex::twist_own(x); }), r;
    for (int i = 0; i < 6; i++) r.v[i] = hex::twist_fin(p.v[i ^ 1], i & 1);
    return r;
  }
  static V inv(const V& a) {
    V s = map1(a, [](const Fq& x, int) { return Fq::sqr(Fq::norm(x)); }), r;
    for (int i = 0; i < 6; i++) r.v[i] = hex::inv_fin(a.v[i], Fq::inv(hex::inv_norm(s.v[i], s.v[i ^ 1], i & 1)), i & 1);
    return r;
  }
  template <int CTRL> static V perm(const V& x) { V r; for (int i = 0; i < 6; i++) r.v[i] = x.v[2 * ((CTRL >> (2 * (i >> 1))) & 3) + (i & 1)]; return r; }
  template <int K> static V bcast(const V& x) { return perm<QP(K, K, K)>(x); }
  template <int K> static V sel(const V& onk, const V& other) { V r; for (int i = 0; i < 6; i++) r.v[i] = ((i >> 1) == K) ? onk.v[i] : other.v[i]; return r; }
  static V pick(const V& a0, const V& a1, const V& a2) { V r; for (int i = 0; i < 6; i++) r.v[i] = (i >> 1) == 0 ? a0.v[i] : (i >> 1) == 1 ? a1.v[i] : a2.v[i]; return r; }
  static F pickf(const F& a0, const F& a1, const F& a2) { return pick(a0, a1, a2); }
  static V zero() { return uni2(Fq::zero(), Fq::zero()); }
  static V one() { return uni2(Fq::one(), Fq::zero()); }
  static V constant(const uint32_t* c0, const uint32_t* c1) { return uni2(Fq::from_limbs(c0), Fq::from_limbs(c1)); }
  static bool is_zero_u(const V& a) { return a.v[0].is_zero_mod_p() && a.v[1].is_zero_mod_p(); }
  static bool is_one3(const V& a, const V& b) {
    bool ok = hex::is_one_half(a.v[0], 0) && a.v[1].is_zero_mod_p();
    for (int i = 2; i < 6; i++) ok = ok && a.v[i].is_zero_mod_p();
    for (int i = 0; i < 6; i++) ok = ok && b.v[i].is_zero_mod_p();
    return ok;
  }
};

#if defined(__HIPCC__)
// device backend: groups of six adjacent lanes, 10 groups per wave64 (lanes 60..63 idle)
struct QHex377 {
  typedef Fq T;
  typedef Fq V;
  typedef Fq F;
  static constexpr int NL = 3, GROUPS_PER_WAVE = 10, NWORDS = 14;
  QDEV static int wave_lane() { return (int)__lane_id(); }
  QDEV static int group() { return (wave_lane() * 43) >> 8; }            // lane / 6 for lane < 64
  QDEV static int sub() { return wave_lane() - 6 * group(); }
  QDEV static int lane() { return sub() >> 1; }                          // tower lane j
  QDEV static int hsel() { return wave_lane() & 1; }                     // which half of the Fq2 (6 g is even)
  QDEV static V from_addr(const V& x, int addr) {
    V r;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) r.l[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)x.l[i]);
    return r;
  }
  // the partner half: lane ^ 1, i.e. DPP quad_perm [1, 0, 3, 2] - a VALU move instead of a ds_bpermute through the LDS crossbar (28
  // per product).  Same-box A/B (tools/ab_pairing.sh): Miller 19.72 -> 18.65 ms, final exponentiation 15.83 -> 15.07 ms per 81920
  // products.  (The three-lane permutes stay ds_bpermute: two of a wave's ten groups straddle a 16-lane DPP row.)
  QDEV static V swap(const V& x) {
    V r;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xB1, 0xF, 0xF, true);
    return r;
  }
  // Per-lane selection by MASK ARITHMETIC, one v_bfi_b32 per word.  Written as `c ? a : b` the compiler turns selections between
  // expensive operands into exec-masked branches (it sinks each operand's computation into "its" lanes' region: 466
  // s_and_saveexec regions in the Miller loop body) - no work is saved on a SIMD, the scalar bookkeeping is pure overhead, and
  // divergent regions push the register allocator into spills (pairing_lanes_kernels.h, ell_slot).
  QDEV static uint32_t lane_mask(bool c) { return 0u - (uint32_t)c; }
  QDEV static V choose(bool c, const V& a, const V& b) {
    const uint32_t m = lane_mask(c);
    V r;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) r.l[i] = (a.l[i] & m) | (b.l[i] & ~m);
    return r;
  }
  // (fetching Y = b's half 0 and D = b's half 1 by two pair-broadcasts instead of one exchange + 28 selects was measured: faster for a
  // lone wave, 4 % slower with the chip full - the LDS crossbar is shared by the CU's four SIMDs)
  // hex::mul with its operand selection done by the exchange itself: X = (h ? partner : own) is the EVEN lane's half in both lanes
  // of a pair (quad_perm [0, 0, 2, 2]), the other factor the ODD lane's (quad_perm [1, 1, 3, 3]), scaled by the lane's -5 or 1 -
  // two DPP moves and a multiplication per limb where the swap + two selects + the multiplication by -5 were four instructions
  QDEV static V mul(const V& a, const V& b) {
    const uint32_t k = hsel() ? 1u : 0u - 5u;
    int32_t cs[NWORDS];
    V X;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) {
      X.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xA0, 0xF, 0xF, true);
      cs[i] = (int32_t)((uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xF5, 0xF, 0xF, true) * k);
    }
    return Fq::mul2s(X, b, cs, swap(b));
  }
  QDEV static V add(const V& a, const V& b) { return hex::add(a, b); }
  QDEV static V dbl(const V& a) { return hex::dbl(a); }
  QDEV static V tpl(const V& a) { return hex::tpl(a); }
  template <int K> QDEV static V sub(const V& a, const V& b) { return hex::sub<K>(a, b); }
  template <int K> QDEV static V neg(const V& a) { return hex::neg<K>(a); }
  QDEV static V wred(const V& a) { return Fq::wred(a); }
  QDEV static V lred(const V& a) { return Fq::norm(a); }
  QDEV static V half(const V& a) { return Fq::half(a); }
  QDEV static V mul_nr(const V& a) {
    const V o = swap(a);
    return choose(hsel() != 0, o, hex::mul_nr(o, 0));
  }
  static constexpr bool LAZY = true;
  template <int K> QDEV static V mul_nr_k(const V& a) {
    const V o = swap(a);
    return choose(hsel() != 0, o, hex::mul_nr_k<K>(o, 0));
  }
  template <int K> QDEV static V mul_nr_k_l(const V& a) {
    const V o = swap(a);
    return choose(hsel() != 0, o, hex::mul_nr_k_l<K>(o, 0));
  }
  QDEV static V add_l(const V& a, const V& b) { return hex::add_l(a, b); }
  QDEV static V dbl_l(const V& a) { return hex::dbl_l(a); }
  template <int K> QDEV static V sub_l(const V& a, const V& b) { return hex::sub_l<K>(a, b); }
  QDEV static V conj(const V& a) { return choose(hsel() != 0, hex::conj(a, 1), a); }
  QDEV static V mul_fp(const V& a, const F& k) { return Fq::mul(a, k); }
  QDEV static V twist_mul(const V& a) {
    const V po = swap(hex::twist_own(a));
    return choose(hsel() != 0, po, hex::twist_fin(po, 0));
  }
  QDEV static V inv(const V& a) {
    const V s = Fq::sqr(Fq::norm(a));
    const V so = swap(s);
    const V n = choose(hsel() != 0, hex::inv_norm(s, so, 1), hex::inv_norm(s, so, 0));
    const V r = Fq::mul(Fq::norm(a), Fq::inv(n));
    return choose(hsel() != 0, Fq::norm(Fq::neg<4, 1>(r)), r);
  }
  template <int CTRL> QDEV static V perm(const V& x) {
    const int j = lane();
    return from_addr(x, (wave_lane() - sub() + 2 * ((CTRL >> (2 * j)) & 3) + hsel()) << 2);
  }
  template <int K> QDEV static V bcast(const V& x) { return perm<QP(K, K, K)>(x); }
  template <int K> QDEV static V sel(const V& onk, const V& other) { return choose(lane() == K, onk, other); }
  QDEV static V pick(const V& a0, const V& a1, const V& a2) {
    const int q = lane();
    const uint32_t m0 = lane_mask(q == 0), m1 = lane_mask(q == 1);
    V r;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) {
      const uint32_t t = (a1.l[i] & m1) | (a2.l[i] & ~m1);
      r.l[i] = (a0.l[i] & m0) | (t & ~m0);
    }
    return r;
  }
  QDEV static F pickf(const F& a0, const F& a1, const F& a2) { return pick(a0, a1, a2); }
  QDEV static V zero() { return Fq::zero(); }
  QDEV static V one() { return choose(hsel() != 0, Fq::zero(), Fq::one()); }
  QDEV static V constant(const uint32_t* c0, const uint32_t* c1) { return choose(hsel() != 0, Fq::from_limbs(c1), Fq::from_limbs(c0)); }
  QDEV static bool is_zero_u(const V& a) {                               // group-uniform Fq2: zero iff both halves are
    const int z = a.is_zero_mod_p() ? 1 : 0;
    return (z & __builtin_amdgcn_ds_bpermute((wave_lane() ^ 1) << 2, z)) != 0;
  }
  QDEV static bool is_one3(const V& a, const V& b) {
    const bool first = sub() == 0;
    int ok = ((first ? hex::is_one_half(a, 0) : a.is_zero_mod_p()) && b.is_zero_mod_p()) ? 1 : 0;
    const int base = (wave_lane() - sub()) << 2;
    int all = 1;
#pragma unroll
    for (int i = 0; i < 6; i++) all &= __builtin_amdgcn_ds_bpermute(base + 4 * i, ok);
    return all != 0;
  }
};
#endif

// ================================================================== tower arithmetic, one Fq12 per lane group
template <class QB> struct QTower {
  typedef typename QB::V V;
  struct E12 { V a, b; };  // lane j < 3: a = c0.c_j, b = c1.c_j (tower order of QuadIO: coefficient j and 3 + j)

  QFN static E12 one12() { return {QB::template sel<0>(QB::one(), QB::zero()), QB::zero()}; }
  // multiplication of an Fq6 element (one coefficient per lane) by v: (xi*x2, x0, x1); needs vb <= 12, result vb <= 64
  QFN static V mul_by_gen(const V& x) {
    V r = QB::template perm<QP(2, 0, 1)>(x);
    return QB::template sel<0>(QB::mul_nr(r), r);
  }
  template <int K> QFN static V mul_by_gen_k(const V& x) {      // for vb(x) <= K; result vb <= 5 K
    V r = QB::template perm<QP(2, 0, 1)>(x);
    return QB::template sel<0>(QB::template mul_nr_k<K>(r), r);
  }
  template <int K> QFN static V mul_by_gen_k_l(const V& x) {    // ... uncarried (mul_nr_k_l): for a sum that is carried right after
    V r = QB::template perm<QP(2, 0, 1)>(x);
    return QB::template sel<0>(QB::template mul_nr_k_l<K>(r), r);
  }
  // Fq6 product, Karatsuba across lanes: lane j computes v_j = x_j y_j and the cross product it needs.  Inputs vb <= 40,
  // output weak-reduced.
  QFN static V mul6(const V& x, const V& y) {
    // Lane j multiplies x_j y_j and the cross pair (j, j + 1): c_j = (x_j + x_{j+1})(y_j + y_{j+1}), t_j = c_j - v_j - v_{j+1}
    // = x_j y_{j+1} + x_{j+1} y_j.  Every lane then needs ONE neighbour for its operands (the rotation R: lane j reads lane j + 1)
    // and z0 = v0 + xi t1, z1 = t0 + xi v2, z2 = t2 + v1 take one exchange of t and one of v: five lane permutes per Fq6 product
    // (with "lane k computes the cross product z_k needs" it was seven - lane 0 needed two neighbours for each operand).  Each
    // permute is 14 ds_bpermute of the six-lane backend, whose waits are what its kernels lose time to.
    constexpr int R = QP(1, 2, 0);
    V v = QB::mul(x, y);
    V xs = QB::add(x, QB::template perm<R>(x));
    V ys = QB::add(y, QB::template perm<R>(y));
    V c = QB::mul(xs, ys);
    V rv = QB::template perm<R>(v);                                            // lane 0: v1, lane 1: v2, lane 2: v0
    V t = QB::template sub<4>(QB::template sub_l<4>(c, v), rv);                // vb <= 11
    V ts = QB::template perm<QP(1, 0, 2)>(t);                                  // lane 0: t1, lane 1: t0, lane 2: t2
    V v1 = QB::template perm<QP(0, 1, 1)>(v);                                  // lane 2: v1 (lanes 0, 1: their own, unused)
    V w = QB::template sel<0>(ts, rv);                                         // what xi multiplies: t1 on lane 0, v2 on lane 1
    V xw = QB::template mul_nr_k_l<16>(w);
    V lhs = QB::template sel<0>(v, ts);                                        // v0 | t0 | t2
    V rhs = QB::template sel<2>(v1, xw);                                       // xi t1 | xi v2 | v1
    return QB::wred(QB::add_l(lhs, rhs));
  }
  QNI static E12 mul12(const E12& x, const E12& y) { return mul12_inl(x, y); }
  QFN static E12 mul12_inl(const E12& x, const E12& y) {
    V v0 = mul6(x.a, y.a);
    V v1 = mul6(x.b, y.b);
    V t = mul6(QB::add(x.a, x.b), QB::add(y.a, y.b));
    E12 r;
    r.b = QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(t, v0), v1));
    r.a = QB::wred(QB::add_l(v0, mul_by_gen_k_l<4>(v1)));
    return r;
  }
  // complex squaring: 2 Fq6 products.  lred = a carry pass where the six-lane backend's value bounds allow it (a weak reduction costs ~2.5x
  // as much; lanes.h backends: lred = wred).  Contract (asserted by the host build, tests/test_host_tower.py): x.b <= 4 p, i.e. a
  // weak-reduced value - the output's b = 2 ab (<= 6 p) goes to mul_by_034 / mul12 next, as in every Miller loop, not into sqr12 again.
  QFN static E12 sqr12(const E12& x) {
    V ab = mul6(x.a, x.b);
    V s2 = QB::lred(QB::add_l(x.a, mul_by_gen_k_l<4>(x.b)));
    V t = mul6(QB::add(x.a, x.b), s2);
    V c0 = QB::template sub_l<64>(QB::template sub_l<4>(t, ab), mul_by_gen_k<4>(ab));
    return {QB::wred(c0), QB::lred(QB::dbl_l(ab))};
  }
  QFN static E12 conj12(const E12& x) { return {x.a, QB::lred(QB::template neg<8>(x.b))}; }   // vb(x.b) <= 8: a squaring's doubled product included
  // x * (d0 + d1 v) for group-uniform d0, d1: lane j: x_j d0 + x_{j-1} d1 (xi on the wrapped term of lane 0)
  QFN static V mul6_by_01(const V& x, const V& d0, const V& d1) {
    V p = QB::mul(x, d0);
    V qv = QB::mul(QB::template perm<QP(2, 0, 1)>(x), d1);
    return QB::wred(QB::add_l(p, QB::template sel<0>(QB::template mul_nr_k_l<4>(qv), qv)));
  }
  // f *= s0 + (s3 + s4 v) w   (ark-ff Fp12::mul_by_034), s* group-uniform
  QFN static void mul_by_034(E12& f, const V& s0, const V& s3, const V& s4) {
    V A = QB::mul(f.a, s0);
    V b = mul6_by_01(f.b, s3, s4);
    V e = mul6_by_01(QB::add(f.a, f.b), QB::add(s0, s3), s4);
    f.b = QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(e, A), b));
    f.a = QB::lred(QB::add_l(A, mul_by_gen_k_l<4>(b)));
  }
  // Two line values of ONE Miller step multiplied with each other before they meet f (round 4): for group-uniform coefficients
  //   (s0 + (s3 + s4 v) w)(t0 + (t3 + t4 v) w) = (s0 t0 + xi s4 t4) + s3 t3 v + (s3 t4 + s4 t3) v^2 + ((s0 t3 + s3 t0) + (s0 t4 + s4 t0) v) w
  // is six Fq2 products - s0 t0, s3 t3, s4 t4 and the three Karatsuba cross products - i.e. TWO product rounds with every lane busy,
  // and the result times f is one Fq12 product whose second Fq6 factor has no v^2 term (mul12_by_line_pair: 2 + 2 + 2 rounds): 8 rounds
  // where two mul_by_034 are 10.  Same field elements: f l_a l_b either way.  Returned as {a = dense Fq6, b = (b0, b1, 0)}.
  // The lines arrive as the tower code produces them: ts / tt hold s0 | s3 (t0 | t3) on lanes 0 | 1 - the two scalings of Pair::ell's
  // mul_fp, lane 2 unused - and s4 / t4 are group-uniform, so the operand patterns are two lane permutes and two selects per line
  // instead of broadcasts and three-way picks.
  QFN static E12 mul_034_by_034(const V& ts, const V& s4, const V& tt, const V& t4) {
    const V v = QB::mul(QB::template sel<2>(s4, ts), QB::template sel<2>(t4, tt));           // s0 t0 | s3 t3 | s4 t4
    const V xs = QB::add(QB::template perm<QP(0, 0, 1)>(ts), QB::template sel<0>(QB::template bcast<1>(ts), s4));   // s0 + s3 | s0 + s4 | s3 + s4
    const V ys = QB::add(QB::template perm<QP(0, 0, 1)>(tt), QB::template sel<0>(QB::template bcast<1>(tt), t4));
    const V c = QB::mul(xs, ys);                                                              // cross sums 03 | 04 | 34
    const V vr = QB::template perm<QP(2, 0, 1)>(v);                                           // s4 t4 | s0 t0 | s3 t3
    const V vs = QB::template perm<QP(1, 2, 0)>(v);                                           // s3 t3 | s4 t4 | s0 t0
    // lane 0: c03 - s0 t0 - s3 t3 = b0;  lane 1: c04 - s0 t0 - s4 t4 = b1;  lane 2: c34 - s3 t3 - s4 t4 = a2
    const V d = QB::template sub<4>(QB::template sub_l<4>(c, QB::template sel<0>(v, vr)), QB::template sel<2>(v, vs));
    const V a0 = QB::lred(QB::add_l(v, QB::template mul_nr_k_l<4>(vr)));                      // lane 0: s0 t0 + xi s4 t4
    return {QB::pick(a0, v, d), QB::template sel<2>(QB::zero(), d)};
  }
  // x * L for L = mul_034_by_034(...): L.a dense, L.b = (b0, b1, 0) one coefficient per lane
  QFN static E12 mul12_by_line_pair(const E12& x, const V& La, const V& Lb) {
    const V v0 = mul6(x.a, La);
    const V v1 = mul6_by_01(x.b, QB::template bcast<0>(Lb), QB::template bcast<1>(Lb));
    const V t = mul6(QB::add(x.a, x.b), QB::add(La, Lb));
    E12 r;
    r.b = QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(t, v0), v1));
    r.a = QB::wred(QB::add_l(v0, mul_by_gen_k_l<4>(v1)));
    return r;
  }
  // Granger-Scott cyclotomic squaring: lane k squares the Fq4 pair k: (a0, b1), (b0, a2), (a1, b2)
  QNI static E12 cyclotomic_sqr(const E12& f) { return cyclotomic_sqr_inl(f); }
  QFN static E12 cyclotomic_sqr_inl(const E12& f) {
    V x = QB::template sel<1>(QB::template perm<QP(0, 0, 1)>(f.b), QB::template perm<QP(0, 0, 1)>(f.a));
    V y = QB::template sel<1>(QB::template perm<QP(1, 2, 2)>(f.a), QB::template perm<QP(1, 1, 2)>(f.b));
    V tmp = QB::mul(x, y);
    V m = QB::mul(QB::add(x, y), QB::add(QB::template mul_nr_k_l<4>(y), x));
    V o0 = QB::lred(QB::template sub_l<64>(QB::template sub_l<4>(m, tmp), QB::template mul_nr_k<4>(tmp)));
    // a_j' = 3 o0 - 2 a_j on every lane;  b_j' = 3 u + 2 b_j with u = o1 = 2 tmp of the previous lane (xi on the wrap to lane 0)
    V u;
    if constexpr (QB::LAZY) {
      V ut = QB::template perm<QP(2, 0, 1)>(tmp);
      u = QB::dbl_l(QB::template sel<0>(QB::template mul_nr_k<4>(ut), ut));
    } else {
      u = QB::template perm<QP(2, 0, 1)>(QB::dbl(tmp));
      u = QB::template sel<0>(QB::lred(QB::mul_nr(u)), u);
    }
    E12 z;
    z.a = QB::wred(QB::add_l(QB::dbl_l(QB::template sub_l<4>(o0, f.a)), o0));
    z.b = QB::wred(QB::add_l(QB::dbl_l(QB::add_l(u, f.b)), u));
    return z;
  }
  // Fq6 inverse, coefficients one per lane (ark-ff Fp6::inverse)
  QFN static V inv6(const V& x) {
    V x0 = QB::template bcast<0>(x), x1 = QB::template bcast<1>(x), x2 = QB::template bcast<2>(x);
    V sA = QB::pick(x0, x2, x1);
    V s = QB::mul(sA, sA);                                     // x0^2, x2^2, x1^2
    V m = QB::mul(QB::pick(x1, x0, x0), QB::pick(x2, x1, x2));  // x1 x2, x0 x1, x0 x2
    // t0 = x0^2 - xi x1 x2;  t1 = xi x2^2 - x0 x1;  t2 = x1^2 - x0 x2
    V u1 = QB::template sel<1>(QB::mul_nr(s), s);
    V u2 = QB::template sel<0>(QB::mul_nr(m), m);
    V t = QB::wred(QB::template sub<64>(u1, u2));
    V pm = QB::mul(sA, t);                                     // x0 t0, x2 t1, x1 t2
    V d = QB::lred(QB::add(QB::template bcast<0>(pm), QB::mul_nr(QB::add(QB::template bcast<1>(pm), QB::template bcast<2>(pm)))));
    return QB::mul(t, QB::inv(d));
  }
  QFN static E12 inv12(const E12& x) {
    V s0 = mul6(x.a, x.a);
    V s1 = mul6(x.b, x.b);
    V d = QB::wred(QB::template sub<64>(s0, mul_by_gen(s1)));
    V di = inv6(d);
    E12 r;
    r.a = mul6(x.a, di);
    r.b = QB::lred(QB::template neg<4>(mul6(x.b, di)));
    return r;
  }
  QFN static bool is_one12(const E12& x) { return QB::is_one3(x.a, x.b); }
};

// ================================================================== the pairing
template <class QB> struct QPairing377 {
  typedef typename QB::V V;
  typedef typename QB::F F;
  typedef QTower<QB> TW;
  typedef typename TW::E12 E12;
  struct Line { V c0, c1, c2; };  // group-uniform

  // x^(q^I): coefficient k (of w^k) is conjugated I times and scaled by xi^(k (q^I - 1) / 6); lane j: a -> k = 2j, b -> 2j + 1
  template <int I> QNI static E12 frob12(const E12& x) {
#define QFC(k) QB::constant(T377::FROB##k##_C0, T377::FROB##k##_C1)
    V ca, cb;
    if constexpr (I == 1) { ca = QB::pick(QB::one(), QFC(1_2), QFC(1_4)); cb = QB::pick(QFC(1_1), QFC(1_3), QFC(1_5)); }
    else if constexpr (I == 2) { ca = QB::pick(QB::one(), QFC(2_2), QFC(2_4)); cb = QB::pick(QFC(2_1), QFC(2_3), QFC(2_5)); }
    else { ca = QB::pick(QB::one(), QFC(3_2), QFC(3_4)); cb = QB::pick(QFC(3_1), QFC(3_3), QFC(3_5)); }
#undef QFC
    V a = (I & 1) ? QB::conj(x.a) : x.a, b = (I & 1) ? QB::conj(x.b) : x.b;
    return {QB::mul(a, ca), QB::mul(b, cb)};
  }

  // ark-ec bls12/g2.rs doubling_step on R = (X, Y, Z), one coordinate per lane.  Statement order keeps few values alive across
  // the three product rounds (each value is 14-28 registers): the line coefficients are finished as soon as their inputs exist and
  // the operands of round 3 are picked per lane before anything else is computed.
  QFN static void double_step(V& Rc, Line& l) {
    V b, e;
    {
      const V r1 = QB::mul(Rc, Rc);                                   // X^2, Y^2, Z^2
      l.c1 = QB::lred(QB::tpl(QB::template bcast<0>(r1)));
      b = QB::template bcast<1>(r1);
      e = QB::twist_mul(QB::tpl(QB::template bcast<2>(r1)));          // B' * 3 Z^2: two Fq products, every lane
    }
    // round 2: lane 0: Y Z, lane 1: X Y, lane 2: e^2
    const V r2 = QB::mul(QB::pick(QB::template bcast<1>(Rc), QB::template bcast<0>(Rc), e), QB::pick(QB::template bcast<2>(Rc), QB::template bcast<1>(Rc), e));
    const V h = QB::dbl(QB::template bcast<0>(r2));    // 2YZ = (Y+Z)^2 - (b + c); vb 6
    l.c0 = QB::lred(QB::template neg<16>(h));
    l.c2 = QB::lred(QB::template sub<4>(e, b));
    const V f3 = QB::tpl(e);                           // vb 9
    const V e23 = QB::tpl(QB::template bcast<2>(r2));  // 3 e^2
    // round 3: lane 0: (XY/2) (b - f3) = X', lane 1: g^2 with g = (b + f3)/2, lane 2: b h = Z'
    const V g = QB::half(QB::add(b, f3));              // vb 6.5
    const V opA = QB::pick(QB::half(QB::template bcast<1>(r2)), g, b);
    const V opB = QB::pick(QB::template sub<16>(b, f3), g, h);
    const V r3 = QB::mul(opA, opB);
    const V y3 = QB::wred(QB::template sub<16>(r3, e23));   // lane 1: g^2 - 3 e^2
    Rc = QB::template sel<1>(y3, r3);
  }
  // ark-ec bls12/g2.rs addition_step; Qc: lane 0 = Q.x, lane 1 = Q.y
  QFN static void add_step(V& Rc, const V& Qc, Line& l) {
    V X = QB::template bcast<0>(Rc), Y = QB::template bcast<1>(Rc), Z = QB::template bcast<2>(Rc);
    V qx = QB::template bcast<0>(Qc), qy = QB::template bcast<1>(Qc);
    V r1 = QB::mul(QB::template sel<0>(qy, qx), Z);                       // lane 0: qy Z, others: qx Z
    V theta = QB::template sub<4>(Y, QB::template bcast<0>(r1)), lambda = QB::template sub<4>(X, QB::template bcast<1>(r1));
    V r2 = QB::mul(QB::pick(theta, lambda, theta), QB::pick(theta, lambda, qx));  // c = theta^2, d = lambda^2, theta qx
    V c = QB::template bcast<0>(r2), d = QB::template bcast<1>(r2);
    V r3 = QB::mul(QB::pick(lambda, Z, X), QB::pick(d, c, d));            // e, f, g
    V e = QB::template bcast<0>(r3), g = QB::template bcast<2>(r3);
    V h = QB::template sub<8>(QB::add(e, QB::template bcast<1>(r3)), QB::dbl(g));  // vb 14
    // round 4: lane 0: lambda qy, lane 1: e Y, lane 2: Z e = Z'
    V r4 = QB::mul(QB::pick(lambda, e, Z), QB::pick(qy, Y, e));
    l.c2 = QB::lred(QB::template sub<4>(QB::template bcast<2>(r2), QB::template bcast<0>(r4)));
    // round 5: lane 0: lambda h = X', lane 1: theta (g - h)
    V r5 = QB::mul(QB::template sel<0>(lambda, theta), QB::template sel<0>(h, QB::template sub<16>(g, h)));
    V y3 = QB::lred(QB::template sub<4>(r5, r4));                         // lane 1: theta (g - h) - e Y
    Rc = QB::pick(r5, y3, r4);
    l.c0 = QB::lred(lambda);
    l.c1 = QB::lred(QB::template neg<8>(theta));
  }
  // f *= line evaluated at P (D-twist: c0 *= P.y, c1 *= P.x): lane 0 scales c0, lane 1 scales c1, then both are broadcast
  QFN static void ell(E12& f, const Line& l, const F& px, const F& py) {
    V t = QB::mul_fp(QB::template sel<0>(l.c0, l.c1), QB::pickf(py, px, px));
    TW::mul_by_034(f, QB::template bcast<0>(t), QB::template bcast<1>(t), l.c2);
  }
  // a line evaluated at P: lane 0: c0 P.y, lane 1: c1 P.x (the 034 coefficients s0, s3; s4 = l.c2 stays group-uniform)
  QFN static V eval_line(const Line& l, const F& px, const F& py) { return QB::mul_fp(QB::template sel<0>(l.c0, l.c1), QB::pickf(py, px, px)); }
  // f *= l_a(P_a) * l_b(P_b), the two lines multiplied first (QTower::mul_034_by_034)
  QFN static void ell2(E12& f, const Line& la, const F& pxa, const F& pya, const Line& lb, const F& pxb, const F& pyb) {
    const E12 L = TW::mul_034_by_034(eval_line(la, pxa, pya), la.c2, eval_line(lb, pxb, pyb), lb.c2);
    f = TW::mul12_by_line_pair(f, L.a, L.b);
  }
  // Miller value of a product of exactly TWO pairs, shared accumulator, the lines of every step merged: the same field element as
  // miller_multi<2> (and as ark-ec's multi-Miller loop) - what k_miller_prepared_slots computes with pair a on prepared lines
  QFN static E12 miller_pair2(const F* px, const F* py, const V* Qc) {
    V Ra = QB::template sel<2>(QB::one(), Qc[0]), Rb = QB::template sel<2>(QB::one(), Qc[1]);
    E12 f = TW::one12();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 62; i >= 0; i--) {
      f = TW::sqr12(f);
      Line la, lb;
      double_step(Ra, la); double_step(Rb, lb);
      ell2(f, la, px[0], py[0], lb, px[1], py[1]);
      if ((T377::X >> i) & 1) {
        add_step(Ra, Qc[0], la); add_step(Rb, Qc[1], lb);
        ell2(f, la, px[0], py[0], lb, px[1], py[1]);
      }
    }
    return f;
  }
  QNI static void step_double(V& Rc, E12& f, const F& px, const F& py) { Line l; double_step(Rc, l); ell(f, l, px, py); }
  QNI static void step_add(V& Rc, const V& Qc, E12& f, const F& px, const F& py) { Line l; add_step(Rc, Qc, l); ell(f, l, px, py); }
  // f_{x,Q}(P): Qc lane 0 = Q.x, lane 1 = Q.y (clean values); px, py group-uniform
  QFN static E12 miller(const F& px, const F& py, const V& Qc) {
    V Rc = QB::template sel<2>(QB::one(), Qc);
    E12 f = TW::one12();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 62; i >= 0; i--) {
      f = TW::sqr12(f);
      step_double(Rc, f, px, py);
      if ((T377::X >> i) & 1) step_add(Rc, Qc, f, px, py);
    }
    return f;
  }
  // Miller value of a whole product of k <= MAXK pairs with ONE accumulator (ark-ec's multi-Miller loop: f is squared once per
  // iteration for all pairs).  px/py/Qc as in miller(), one entry per pair.
  template <int MAXK> QFN static E12 miller_multi(int k, const F* px, const F* py, const V* Qc) {
    V Rc[MAXK];
    for (int p = 0; p < k; p++) Rc[p] = QB::template sel<2>(QB::one(), Qc[p]);
    E12 f = TW::one12();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 62; i >= 0; i--) {
      f = TW::sqr12(f);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
      for (int p = 0; p < k; p++) step_double(Rc[p], f, px[p], py[p]);
      if ((T377::X >> i) & 1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int p = 0; p < k; p++) step_add(Rc[p], Qc[p], f, px[p], py[p]);
      }
    }
    return f;
  }
  QNI static E12 exp_by_x(const E12& f) {
    E12 acc = f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 62; i >= 0; i--) {
      acc = TW::cyclotomic_sqr(acc);
      if ((T377::X >> i) & 1) acc = TW::mul12(acc, f);
    }
    return acc;
  }
  // ark-ec bls12 final_exponentiation (same chain as pairing.h).  EX: the exp_by_x routine (the kernels bring one that keeps its
  // accumulator in LDS instead of passing Fq12 values through private memory).
  QFN static E12 final_exponentiation(const E12& f) { return final_exponentiation_t(f, [](const E12& v) { return exp_by_x(v); }); }
  template <class EX> QFN static E12 final_exponentiation_t(const E12& f, EX exp_by_x) {
    E12 f2 = TW::inv12(f);
    E12 r = TW::mul12(TW::conj12(f), f2);
    f2 = r;
    r = TW::mul12(frob12<2>(r), f2);
    E12 y0 = TW::conj12(TW::cyclotomic_sqr(r));
    E12 y5 = exp_by_x(r);
    E12 y1 = TW::cyclotomic_sqr(y5);
    E12 y3 = TW::mul12(y0, y5);
    y0 = exp_by_x(y3);
    E12 y2 = exp_by_x(y0);
    E12 y4 = TW::mul12(exp_by_x(y2), y1);
    y1 = exp_by_x(y4);
    y3 = TW::conj12(y3);
    y1 = TW::mul12(TW::mul12(y1, y3), r);
    y3 = TW::conj12(r);
    y0 = frob12<3>(TW::mul12(y0, r));
    y4 = frob12<1>(TW::mul12(y4, y3));
    y5 = frob12<2>(TW::mul12(y5, y2));
    y5 = TW::mul12(TW::mul12(y5, y0), y4);
    return TW::mul12(y5, y1);
  }
};

// ================================================================== BW6-761 (Groth16 verify, crates/epoch-snark/src/api/verifier.rs:35)
// Same lane-parallel tower (QTower is generic over the base policy: here one Fq coefficient per lane and per half), ark-ec
// models/bw6 formulas as in pairing.h: two Miller loops (x+1, and the signed digits of x^3-x^2-x), f1 * frob(f2), then
// (q^3-1)(q+1) and the hard part m^R0(x) * (m^q)^R1(x).  G2 coordinates are in Fq (M-type twist, B' = 4).
template <class QB> struct QPairing761 {
  typedef typename QB::V V;
  typedef V F;                       // P's coordinates live in the same field as the tower base
  typedef QTower<QB> TW;
  typedef typename TW::E12 E12;      // an Fq6 element of BW6-761 (the name is the tower template's)
  struct Line { V c0, c1, c2; };

  // x * (d1 v) for a group-uniform d1: lane j: x_{j-1} d1 (nonresidue on the wrap to lane 0)
  QFN static V mul6_by_1(const V& x, const V& d1) {
    V qv = QB::mul(QB::template perm<QP(2, 0, 1)>(x), d1);
    return QB::wred(QB::template sel<0>(QB::mul_nr(qv), qv));
  }
  // f *= (s0 + s1 u) + (s4 u) v   (ark-ff Fp6_2over3::mul_by_014)
  QFN static void mul_by_014(E12& f, const V& s0, const V& s1, const V& s4) {
    V v0 = TW::mul6_by_01(f.a, s0, s1);
    V v1 = mul6_by_1(f.b, s4);
    V t = TW::mul6_by_01(QB::add(f.a, f.b), s0, QB::add(s1, s4));
    f.b = QB::wred(QB::template sub<4>(QB::template sub<4>(t, v0), v1));
    f.a = QB::wred(QB::add(v0, TW::mul_by_gen(v1)));
  }
  QFN static void double_step(V& Rc, Line& l) {
    V r1 = QB::mul(Rc, Rc);                                           // X^2, Y^2, Z^2
    V b = QB::template bcast<1>(r1), c = QB::template bcast<2>(r1);
    V c3 = QB::tpl(c);
    V e = QB::wred(QB::dbl(QB::dbl(c3)));                             // B' * 3c = 12 c
    V e_2 = QB::dbl(e);
    // round 2: lane 0: Y Z, lane 1: X Y, lane 2: (2e)^2
    V r2 = QB::mul(QB::pick(QB::template bcast<1>(Rc), QB::template bcast<0>(Rc), e_2), QB::pick(QB::template bcast<2>(Rc), QB::template bcast<1>(Rc), e_2));
    V h = QB::dbl(QB::template bcast<0>(r2));                         // 2YZ = (Y+Z)^2 - (b + c)
    V f3 = QB::tpl(e);                                                // vb 9
    V g = QB::add(b, f3);
    V a2 = QB::dbl(QB::template bcast<1>(r2));                        // 2XY
    V i = QB::template sub<4>(e, b);
    V e2s = QB::template bcast<2>(r2);
    // round 3: lane 0: 2a (b - f3) = X', lane 1: g^2, lane 2: 4b h = Z'
    V r3 = QB::mul(QB::pick(a2, g, QB::dbl(QB::dbl(b))), QB::pick(QB::template sub<16>(b, f3), g, h));
    V y3 = QB::wred(QB::template sub<8>(r3, QB::tpl(e2s)));           // lane 1: g^2 - 3 (2e)^2
    Rc = QB::template sel<1>(y3, r3);
    l.c0 = QB::wred(i);
    l.c1 = QB::wred(QB::tpl(QB::template bcast<0>(r1)));
    l.c2 = QB::wred(QB::template neg<16>(h));
  }
  QFN static void add_step(V& Rc, const V& Qc, Line& l) {            // Qc: lane 0 = Q.x, lane 1 = +-Q.y
    V X = QB::template bcast<0>(Rc), Y = QB::template bcast<1>(Rc), Z = QB::template bcast<2>(Rc);
    V qx = QB::template bcast<0>(Qc), qy = QB::template bcast<1>(Qc);
    V r1 = QB::mul(QB::template sel<0>(qy, qx), Z);
    V theta = QB::template sub<4>(Y, QB::template bcast<0>(r1)), lambda = QB::template sub<4>(X, QB::template bcast<1>(r1));
    V r2 = QB::mul(QB::pick(theta, lambda, theta), QB::pick(theta, lambda, qx));
    V c = QB::template bcast<0>(r2), d = QB::template bcast<1>(r2);
    V r3 = QB::mul(QB::pick(lambda, Z, X), QB::pick(d, c, d));
    V e = QB::template bcast<0>(r3), g = QB::template bcast<2>(r3);
    V h = QB::template sub<8>(QB::add(e, QB::template bcast<1>(r3)), QB::dbl(g));
    V r4 = QB::mul(QB::pick(lambda, e, Z), QB::pick(qy, Y, e));
    l.c0 = QB::wred(QB::template sub<4>(QB::template bcast<2>(r2), QB::template bcast<0>(r4)));
    V r5 = QB::mul(QB::template sel<0>(lambda, theta), QB::template sel<0>(h, QB::template sub<16>(g, h)));
    V y3 = QB::wred(QB::template sub<4>(r5, r4));
    Rc = QB::pick(r5, y3, r4);
    l.c1 = QB::wred(QB::template neg<8>(theta));
    l.c2 = QB::wred(lambda);
  }
  // f *= c0 + (c1 P.x) u + (c2 P.y) u v: lane 0 scales c1, lane 1 scales c2, both broadcast
  QFN static void ell(E12& f, const Line& l, const F& px, const F& py) {
    V t = QB::mul(QB::template sel<0>(l.c1, l.c2), QB::pick(px, py, py));
    mul_by_014(f, l.c0, QB::template bcast<0>(t), QB::template bcast<1>(t));
  }
  QNI static void step_double(V& Rc, E12& f, const F& px, const F& py) { Line l; double_step(Rc, l); ell(f, l, px, py); }
  QNI static void step_add(V& Rc, const V& Qc, E12& f, const F& px, const F& py) { Line l; add_step(Rc, Qc, l); ell(f, l, px, py); }
  QNI static E12 frob1(const E12& x) {
    V ca = QB::pick(QB::one(), QB::constant(T761::FROB1_2), QB::constant(T761::FROB1_4));
    V cb = QB::pick(QB::constant(T761::FROB1_1), QB::constant(T761::FROB1_3), QB::constant(T761::FROB1_5));
    return {QB::mul(x.a, ca), QB::mul(x.b, cb)};
  }
  // the two Miller loops of the optimal ate pairing: f1 = f_{x+1,Q}(P), f2 = f_{x^3-x^2-x,Q}(P) (signed digits); e = f1 * frob(f2).
  // They are independent: the single-product latency path runs them in different waves (pairing_lanes_kernels.h, "wide" kernels).
  QFN static E12 miller_f1(const F& px, const F& py, const V& Qc) {
    V Rc = QB::template sel<2>(QB::one(), Qc);
    E12 f1 = TW::one12();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 62; i >= 0; i--) {
      f1 = TW::sqr12(f1);
      step_double(Rc, f1, px, py);
      if ((T761::LOOP1 >> i) & 1) step_add(Rc, Qc, f1, px, py);
    }
    return f1;
  }
  QFN static E12 miller_f2(const F& px, const F& py, const V& Qc) {
    V Rc = QB::template sel<2>(QB::one(), Qc);
    const V Qn = QB::template sel<1>(QB::wred(QB::template neg<4>(Qc)), Qc);   // (Q.x, -Q.y)
    E12 f2 = TW::one12();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = T761::LOOP2_LEN - 1; i >= 1; i--) {
      if (i != T761::LOOP2_LEN - 1) f2 = TW::sqr12(f2);
      step_double(Rc, f2, px, py);
      const int d = T761::LOOP2_NAF[i - 1];
      if (d > 0) step_add(Rc, Qc, f2, px, py);
      else if (d < 0) step_add(Rc, Qn, f2, px, py);
    }
    return f2;
  }
  QFN static E12 miller(const F& px, const F& py, const V& Qc) {
    const E12 f1 = miller_f1(px, py, Qc);
    return TW::mul12(f1, frob1(miller_f2(px, py, Qc)));
  }
  // whole product in one group (k <= MAXK pairs, one accumulator per loop): same value as the product of the per-pair loops
  template <int MAXK> QFN static E12 miller_multi(int k, const F* px, const F* py, const V* Qc) {
    E12 acc = TW::one12();
    for (int p = 0; p < k; p++) acc = (p == 0) ? miller(px[p], py[p], Qc[p]) : TW::mul12(acc, miller(px[p], py[p], Qc[p]));
    return acc;
  }
  QNI static E12 pow(const E12& f, const uint64_t* e, int bits, bool neg) {
    E12 acc = f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = bits - 2; i >= 0; i--) {
      acc = TW::sqr12(acc);
      if ((e[i >> 6] >> (i & 63)) & 1) acc = TW::mul12(acc, f);
    }
    return neg ? TW::conj12(acc) : acc;
  }
  // f^e for the signed-digit (NAF) expansion d[0 .. len) of |e|, least significant first, d[len - 1] = 1; f unitary (inverse =
  // conjugate: after the easy part).  A third fewer multiplications than the binary ladder; same field element.
  QFN static E12 pow_naf(const E12& f, const int8_t* d, int len, bool neg) {
    const E12 fc = TW::conj12(f);
    E12 acc = f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = len - 2; i >= 0; i--) {
      acc = TW::sqr12(acc);
      if (d[i] > 0) acc = TW::mul12(acc, f);
      else if (d[i] < 0) acc = TW::mul12(acc, fc);
    }
    return neg ? TW::conj12(acc) : acc;
  }
  QFN static E12 easy_part(const E12& f) {
    E12 a = TW::mul12(TW::conj12(f), TW::inv12(f));        // f^(q^3 - 1)
    return TW::mul12(frob1(a), a);                         // ^(q + 1)
  }
  QFN static E12 final_exponentiation(const E12& f) {
    E12 a = TW::mul12(TW::conj12(f), TW::inv12(f));        // f^(q^3 - 1)
    E12 m = TW::mul12(frob1(a), a);                        // ^(q + 1)
    E12 p0 = pow(m, T761::R0_MAG, T761::R0_BITS, T761::R0_NEG);
    E12 p1 = pow(frob1(m), T761::R1_MAG, T761::R1_BITS, T761::R1_NEG);
    return TW::mul12(p0, p1);
  }
};

}  // namespace celo
