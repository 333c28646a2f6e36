// Quadratic extension Fq2 = Fq[u]/(u^2 + 5) of the BLS12-377 base field, same interface and
// bounds contract as Fp (fp.h) so the curve formulas (curve.h) and the MSM pipeline (msm.h) are
// generic over it.  Replaces ark-ff's Fp2<Fq2Parameters> (ark-bls12-377, NONRESIDUE = -5) on the
// G2 paths: crates/bls-crypto/src/bls/public.rs:61 (G2 MSM) and the Miller loop's G2 arithmetic.
//
// Every product is two sum-of-products passes (Fp::mul2): c0 = a0 b0 - 5 a1 b1, c1 = a0 b1 + a1 b0,
// i.e. 4 limb-product sweeps + 2 Montgomery reductions, no Karatsuba operand additions; a difference of two products (the
// curve formulas' Y3) is two passes over FOUR products each (Fp::mul4k), a squaring's real part one symmetric pass (Fp::sqr2m5).
// mul2 needs normalised inputs, so mul/sqr normalise theirs (cheap next to 6 L^2 mads).
// Outputs: lb = 1, vb <= 3.
#pragma once
#include "fp.h"
// Reproducer builds of the round-3 open finding only: -DCELO_MUL4K_SGN_SITES=<mask> builds the curve formulas' Fq2 pass a b - c d (Y3) in
// its SIGNED form (Fp::mul4k<KC, true>) at the call sites whose bit is set - 1: xyzz_dbl_affine, 2: xyzz_dbl, 4: xyzz_madd, 8:
// xyzz_add_affine, 16: xyzz_add (curve.h).  The shipped library builds with the mask 0: every site unsigned.
#ifndef CELO_MUL4K_SGN_SITES
#define CELO_MUL4K_SGN_SITES 0
#endif

namespace celo {

template <class P> struct Fp2 {
  typedef Fp<P> B;
  B c0, c1;
  static constexpr int WORDS = 2 * B::WORDS;
  static constexpr int ARK64 = 2 * P::N64;

  HD static Fp2 zero() { return {B::zero(), B::zero()}; }
  HD static Fp2 one() { return {B::one(), B::zero()}; }
  HD static Fp2 mul(const Fp2& a, const Fp2& b) {
    B a0 = B::norm(a.c0), a1 = B::norm(a.c1), b0 = B::norm(b.c0), b1 = B::norm(b.c1);
    return {B::template mul2<true>(a0, b0, a1, b1), B::template mul2<false>(a0, b1, a1, b0)};
  }
  HD static Fp2 sqr(const Fp2& a) {     // (a0^2 - 5 a1^2) + 2 a0 a1 u: the real part with the symmetric limb products taken once
    B a0 = B::norm(a.c0), a1 = B::norm(a.c1);
    return {B::sqr2m5(a0, a1), B::mul(B::dbl(a0), a1)};
  }
  // a b - c d with ONE reduction per half: c0 = a0 b0 - 5 a1 b1 - c0 d0 + 5 c1 d1, c1 = a0 b1 + a1 b0 - c0 d1 - c1 d0
  HD static Fp2 mul_sub(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d) {
    const B a0 = B::norm(a.c0), a1 = B::norm(a.c1), b0 = B::norm(b.c0), b1 = B::norm(b.c1);
    const B c0 = B::norm(c.c0), c1 = B::norm(c.c1), d0 = B::norm(d.c0), d1 = B::norm(d.c1);
    return {B::template mul4k<-5>(a0, b0, a1, b1, c0, d0, c1, d1), B::template mul4k<1>(a0, b1, a1, b0, c0, d1, c1, d0)};
  }
  // operands prepared once by the caller (curve.h): every half normalised - see Fp::prep
  HD static Fp2 prep(const Fp2& a) { return norm(a); }
  HD static Fp2 mul_nn(const Fp2& a, const Fp2& b) {
    return {B::template mul2<true>(a.c0, b.c0, a.c1, b.c1), B::template mul2<false>(a.c0, b.c1, a.c1, b.c0)};
  }
  HD static Fp2 sqr_nn(const Fp2& a) { return {B::sqr2m5(a.c0, a.c1), B::mul(B::dbl(a.c0), a.c1)}; }
  template <bool SGN = false> HD static Fp2 mul_sub_nn(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d) {
    return {B::template mul4k<-5, SGN>(a.c0, b.c0, a.c1, b.c1, c.c0, d.c0, c.c1, d.c1), B::template mul4k<1, SGN>(a.c0, b.c1, a.c1, b.c0, c.c0, d.c1, c.c1, d.c0)};
  }
  template <int SITE> HD static Fp2 mul_sub_nn_at(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d) {
    return mul_sub_nn<((CELO_MUL4K_SGN_SITES) >> SITE & 1) != 0>(a, b, c, d);
  }
  HD static Fp2 mul_fp(const Fp2& a, const B& k) {  // k normalised or lb*lb within Fp::mul's bound
    return {B::mul(a.c0, k), B::mul(a.c1, k)};
  }
  HD static Fp2 add(const Fp2& a, const Fp2& b) { return {B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
  HD static Fp2 dbl(const Fp2& a) { return add(a, a); }
  template <int K, int M = 1> HD static Fp2 sub(const Fp2& a, const Fp2& b) {
    return {B::template sub<K, M>(a.c0, b.c0), B::template sub<K, M>(a.c1, b.c1)};
  }
  template <int K, int M = 1> HD static Fp2 neg(const Fp2& b) { return sub<K, M>(zero(), b); }
  HD static Fp2 norm(const Fp2& a) { return {B::norm(a.c0), B::norm(a.c1)}; }
  HD static Fp2 conj(const Fp2& a) { return {a.c0, B::norm(B::template neg<64, 1>(B::norm(a.c1)))}; }
  // (a0 + a1 u) * u = -5 a1 + a0 u
  HD static Fp2 mul_by_u(const Fp2& a) {
    B t = B::norm(a.c1);
    B t5 = B::norm(B::add(B::dbl(B::dbl(t)), t));  // 5*a1: lb 1, vb 5*vb
    return {B::norm(B::template neg<64, 1>(t5)), a.c0};
  }
  HD bool is_zero_mod_p() const { return c0.is_zero_mod_p() && c1.is_zero_mod_p(); }
  HD bool limbs_all_zero() const { return c0.limbs_all_zero() && c1.limbs_all_zero(); }
  HD static Fp2 load(const uint32_t* p) { return {B::load(p), B::load(p + B::WORDS)}; }
  HD void store(uint32_t* p) const { c0.store(p); c1.store(p + B::WORDS); }
  HD static Fp2 from_ark(const uint64_t* s) { return {B::from_ark(s), B::from_ark(s + P::N64)}; }
  HD void to_ark(uint64_t* d) const { c0.to_ark(d); c1.to_ark(d + P::N64); }
  HD static Fp2 inv(const Fp2& a) {
    B a0 = B::norm(a.c0), a1 = B::norm(a.c1);
    // norm = a0^2 + 5 a1^2
    B s1 = B::sqr(a1);
    B n = B::add(B::sqr(a0), B::add(B::dbl(B::dbl(s1)), s1));  // lb 6
    B ni = B::inv(B::norm(n));
    return {B::mul(a0, ni), B::norm(B::template neg<4, 1>(B::mul(a1, ni)))};
  }
};

}  // namespace celo
