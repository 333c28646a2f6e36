// BLS12-377 pairing tower over the lazy 28-bit-limb base field:
//   Fq2 = Fq[u]/(u^2+5) (fp2.h),  Fq6 = Fq2[v]/(v^3 - u),  Fq12 = Fq6[w]/(w^2 - v).
// Replaces ark-ff's Fp6_3over2 / Fp12_2over6 as used by ark-ec's Bls12 pairing engine, which the reference enters
// through Bls12_377::product_of_pairings (crates/bls-crypto/src/bls/public.rs:102, signature.rs:149).
//
// Value-growth discipline (fp.h contract): every Fq6/Fq12-level result is passed through Fp::wred (value < ~1.05p,
// normalised limbs), Fq2 products are < 3p; inside a function additions/subtractions are lazy and every multiply input
// stays below ~80p (checked at run time by the CELO_FP_TRACK host build, tests/test_host_tower.py).
//
// The Fq2-multiply-and-up functions are deliberately NOT inlined on the device: an Fq12 multiplication is 36 Montgomery
// passes (~22k instructions); operands live in per-lane private memory and the code stays a few tens of KB.
#pragma once
#include "fp2.h"

namespace celo {

typedef Fp<P377> Fq;
typedef Fp2<P377> Fq2;

#if defined(__HIPCC__)
#define TW_FN __host__ __device__ __attribute__((noinline))
#else
#define TW_FN inline
#endif

struct Fq6 { Fq2 c0, c1, c2; };
struct Fq12 { Fq6 c0, c1; };

// ------------------------------------------------------------------ Fq2 helpers (lazy; results normalised)
HD Fq2 f2_wred(const Fq2& a) { return {Fq::wred(a.c0), Fq::wred(a.c1)}; }
HD Fq2 f2_add(const Fq2& a, const Fq2& b) { return Fq2::norm(Fq2::add(a, b)); }
HD Fq2 f2_dbl(const Fq2& a) { return Fq2::norm(Fq2::add(a, a)); }
HD Fq2 f2_tpl(const Fq2& a) { return Fq2::norm(Fq2::add(Fq2::add(a, a), a)); }
template <int K> HD Fq2 f2_sub(const Fq2& a, const Fq2& b) { return Fq2::norm(Fq2::template sub<K, 1>(a, b)); }
template <int K> HD Fq2 f2_neg(const Fq2& a) { return Fq2::norm(Fq2::template neg<K, 1>(a)); }
HD Fq2 f2_mul_xi(const Fq2& a) { return Fq2::mul_by_u(a); }  // needs vb(a.c1) <= 12; result vb <= 64 / vb(a.c0)
HD Fq2 f2_from(const uint32_t* c0, const uint32_t* c1) { return {Fq::from_limbs(c0), Fq::from_limbs(c1)}; }
TW_FN void f2_mul(Fq2& r, const Fq2& a, const Fq2& b) { r = Fq2::mul(a, b); }
TW_FN void f2_sqr(Fq2& r, const Fq2& a) { r = Fq2::sqr(a); }
TW_FN void f2_mul_fp(Fq2& r, const Fq2& a, const Fq& k) { r = Fq2::mul_fp(Fq2::norm(a), k); }
TW_FN void f2_inv(Fq2& r, const Fq2& a) { r = Fq2::inv(a); }

// ------------------------------------------------------------------ Fq6
HD Fq6 f6_zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
HD Fq6 f6_one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
HD Fq6 f6_add(const Fq6& a, const Fq6& b) { return {f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; }
template <int K> HD Fq6 f6_sub(const Fq6& a, const Fq6& b) { return {f2_sub<K>(a.c0, b.c0), f2_sub<K>(a.c1, b.c1), f2_sub<K>(a.c2, b.c2)}; }
HD Fq6 f6_wred(const Fq6& a) { return {f2_wred(a.c0), f2_wred(a.c1), f2_wred(a.c2)}; }
HD Fq6 f6_neg(const Fq6& a) { return f6_wred({f2_neg<4>(a.c0), f2_neg<4>(a.c1), f2_neg<4>(a.c2)}); }  // a clean (vb <= 4)
HD Fq6 f6_mul_by_v(const Fq6& a) { return {f2_mul_xi(a.c2), a.c0, a.c1}; }                               // vb(a.c2) <= 12

// r = a*b; inputs vb <= 40, output wred'ed
TW_FN void f6_mul(Fq6& r, const Fq6& a, const Fq6& b) {
  Fq2 v0, v1, v2, t;
  f2_mul(v0, a.c0, b.c0);
  f2_mul(v1, a.c1, b.c1);
  f2_mul(v2, a.c2, b.c2);
  f2_mul(t, f2_add(a.c1, a.c2), f2_add(b.c1, b.c2));
  t = f2_sub<4>(f2_sub<4>(t, v1), v2);                       // vb <= 11
  Fq2 r0 = f2_wred(f2_add(v0, f2_mul_xi(t)));
  f2_mul(t, f2_add(a.c0, a.c1), f2_add(b.c0, b.c1));
  t = f2_sub<4>(f2_sub<4>(t, v0), v1);
  Fq2 r1 = f2_wred(f2_add(t, f2_mul_xi(v2)));
  f2_mul(t, f2_add(a.c0, a.c2), f2_add(b.c0, b.c2));
  t = f2_sub<4>(f2_sub<4>(t, v0), v2);
  Fq2 r2 = f2_wred(f2_add(t, v1));
  r.c0 = r0; r.c1 = r1; r.c2 = r2;
}
// r = x * (b0 + b1 v)
TW_FN void f6_mul_by_01(Fq6& r, const Fq6& x, const Fq2& b0, const Fq2& b1) {
  Fq2 p, q;
  f2_mul(p, x.c0, b0);
  f2_mul(q, x.c2, b1);
  Fq2 t0 = f2_wred(f2_add(p, f2_mul_xi(q)));
  f2_mul(p, x.c0, b1);
  f2_mul(q, x.c1, b0);
  Fq2 t1 = f2_wred(f2_add(p, q));
  f2_mul(p, x.c1, b1);
  f2_mul(q, x.c2, b0);
  Fq2 t2 = f2_wred(f2_add(p, q));
  r.c0 = t0; r.c1 = t1; r.c2 = t2;
}
TW_FN void f6_inv(Fq6& r, const Fq6& a) {
  Fq2 s, m, t0, t1, t2, d;
  f2_sqr(s, a.c0); f2_mul(m, a.c1, a.c2);
  t0 = f2_wred(f2_sub<64>(s, f2_mul_xi(m)));
  f2_sqr(s, a.c2); f2_mul(m, a.c0, a.c1);
  t1 = f2_wred(f2_sub<4>(f2_mul_xi(s), m));
  f2_sqr(s, a.c1); f2_mul(m, a.c0, a.c2);
  t2 = f2_wred(f2_sub<4>(s, m));
  f2_mul(d, a.c0, t0);
  f2_mul(m, a.c2, t1); d = f2_add(d, f2_mul_xi(m));
  f2_mul(m, a.c1, t2); d = f2_wred(f2_add(f2_wred(d), f2_mul_xi(m)));
  Fq2 di;
  f2_inv(di, d);
  f2_mul(r.c0, t0, di);
  f2_mul(r.c1, t1, di);
  f2_mul(r.c2, t2, di);
}

// ------------------------------------------------------------------ Fq12
HD Fq12 f12_one() { return {f6_one(), f6_zero()}; }
TW_FN void f12_mul(Fq12& r, const Fq12& a, const Fq12& b) {
  Fq6 v0, v1, t;
  f6_mul(v0, a.c0, b.c0);
  f6_mul(v1, a.c1, b.c1);
  f6_mul(t, f6_add(a.c0, a.c1), f6_add(b.c0, b.c1));
  r.c1 = f6_wred(f6_sub<4>(f6_sub<4>(t, v0), v1));
  r.c0 = f6_wred(f6_add(v0, f6_mul_by_v(v1)));
}
TW_FN void f12_sqr(Fq12& r, const Fq12& a) {
  Fq6 ab, t;
  f6_mul(ab, a.c0, a.c1);
  Fq6 s2 = f6_wred(f6_add(a.c0, f6_mul_by_v(a.c1)));
  f6_mul(t, f6_add(a.c0, a.c1), s2);
  Fq6 vab = f6_mul_by_v(ab);
  Fq6 c0 = f6_sub<4>(t, ab);
  c0 = {f2_sub<64>(c0.c0, vab.c0), f2_sub<4>(c0.c1, vab.c1), f2_sub<4>(c0.c2, vab.c2)};
  r.c0 = f6_wred(c0);
  r.c1 = f6_wred(f6_add(ab, ab));
}
HD Fq12 f12_conj(const Fq12& a) { return {a.c0, f6_neg(a.c1)}; }
// f *= s0 + (s3 + s4 v) w        (ark-ff Fp12::mul_by_034; D-twist line placement)
TW_FN void f12_mul_by_034(Fq12& f, const Fq2& s0, const Fq2& s3, const Fq2& s4) {
  Fq6 a, b, e;
  f2_mul(a.c0, f.c0.c0, s0);
  f2_mul(a.c1, f.c0.c1, s0);
  f2_mul(a.c2, f.c0.c2, s0);
  f6_mul_by_01(b, f.c1, s3, s4);
  f6_mul_by_01(e, f6_add(f.c0, f.c1), f2_add(s0, s3), s4);
  f.c1 = f6_wred(f6_sub<4>(f6_sub<4>(e, a), b));
  f.c0 = f6_wred(f6_add(a, f6_mul_by_v(b)));
}
TW_FN void f12_inv(Fq12& r, const Fq12& a) {
  Fq6 s0, s1, d, di;
  f6_mul(s0, a.c0, a.c0);
  f6_mul(s1, a.c1, a.c1);
  Fq6 vs1 = f6_mul_by_v(s1);
  d = f6_wred({f2_sub<64>(s0.c0, vs1.c0), f2_sub<4>(s0.c1, vs1.c1), f2_sub<4>(s0.c2, vs1.c2)});
  f6_inv(di, d);
  f6_mul(r.c0, a.c0, di);
  Fq6 m;
  f6_mul(m, a.c1, di);
  r.c1 = f6_neg(m);
}
// a^(q^i), i in {1,2,3}: coefficient k (of w^k) is conjugated i times and scaled by g_i^k, g_i = xi^((q^i-1)/6)
template <int I> HD Fq2 frob_coeff(int k) {
  if constexpr (I == 1) {
    switch (k) {
      case 1: return f2_from(T377::FROB1_1_C0, T377::FROB1_1_C1);
      case 2: return f2_from(T377::FROB1_2_C0, T377::FROB1_2_C1);
      case 3: return f2_from(T377::FROB1_3_C0, T377::FROB1_3_C1);
      case 4: return f2_from(T377::FROB1_4_C0, T377::FROB1_4_C1);
      default: return f2_from(T377::FROB1_5_C0, T377::FROB1_5_C1);
    }
  } else if constexpr (I == 2) {
    switch (k) {
      case 1: return f2_from(T377::FROB2_1_C0, T377::FROB2_1_C1);
      case 2: return f2_from(T377::FROB2_2_C0, T377::FROB2_2_C1);
      case 3: return f2_from(T377::FROB2_3_C0, T377::FROB2_3_C1);
      case 4: return f2_from(T377::FROB2_4_C0, T377::FROB2_4_C1);
      default: return f2_from(T377::FROB2_5_C0, T377::FROB2_5_C1);
    }
  } else {
    switch (k) {
      case 1: return f2_from(T377::FROB3_1_C0, T377::FROB3_1_C1);
      case 2: return f2_from(T377::FROB3_2_C0, T377::FROB3_2_C1);
      case 3: return f2_from(T377::FROB3_3_C0, T377::FROB3_3_C1);
      case 4: return f2_from(T377::FROB3_4_C0, T377::FROB3_4_C1);
      default: return f2_from(T377::FROB3_5_C0, T377::FROB3_5_C1);
    }
  }
}
template <int I> TW_FN void f12_frob(Fq12& r, const Fq12& a) {
  auto cj = [](const Fq2& x) -> Fq2 { return (I & 1) ? Fq2{x.c0, Fq::wred(Fq::norm(Fq::template neg<4, 1>(x.c1)))} : x; };
  Fq2 t;
  Fq12 o;
  o.c0.c0 = cj(a.c0.c0);
  f2_mul(t, cj(a.c0.c1), frob_coeff<I>(2)); o.c0.c1 = t;
  f2_mul(t, cj(a.c0.c2), frob_coeff<I>(4)); o.c0.c2 = t;
  f2_mul(t, cj(a.c1.c0), frob_coeff<I>(1)); o.c1.c0 = t;
  f2_mul(t, cj(a.c1.c1), frob_coeff<I>(3)); o.c1.c1 = t;
  f2_mul(t, cj(a.c1.c2), frob_coeff<I>(5)); o.c1.c2 = t;
  r = o;
}
// Granger-Scott squaring in the cyclotomic subgroup (ark-ff Fp12::cyclotomic_square)
TW_FN void f12_cyclotomic_sqr(Fq12& r, const Fq12& a) {
  auto fp4sq = [](const Fq2& x, const Fq2& y, Fq2& o0, Fq2& o1) {
    Fq2 tmp, m;
    f2_mul(tmp, x, y);
    f2_mul(m, f2_add(x, y), f2_add(f2_mul_xi(y), x));
    o0 = f2_wred(f2_sub<64>(f2_sub<4>(m, tmp), f2_mul_xi(tmp)));
    o1 = f2_dbl(tmp);
  };
  const Fq2 &r0 = a.c0.c0, &r4 = a.c0.c1, &r3 = a.c0.c2, &r2 = a.c1.c0, &r1 = a.c1.c1, &r5 = a.c1.c2;
  Fq2 t0, t1, t2, t3, t4, t5;
  fp4sq(r0, r1, t0, t1);
  fp4sq(r2, r3, t2, t3);
  fp4sq(r4, r5, t4, t5);
  Fq12 z;
  z.c0.c0 = f2_wred(f2_add(f2_dbl(f2_sub<4>(t0, r0)), t0));
  z.c1.c1 = f2_wred(f2_add(f2_dbl(f2_add(t1, r1)), t1));
  Fq2 tmp = f2_wred(f2_mul_xi(t5));
  z.c1.c0 = f2_wred(f2_add(f2_dbl(f2_add(tmp, r2)), tmp));
  z.c0.c2 = f2_wred(f2_add(f2_dbl(f2_sub<4>(t4, r3)), t4));
  z.c0.c1 = f2_wred(f2_add(f2_dbl(f2_sub<4>(t2, r4)), t2));
  z.c1.c2 = f2_wred(f2_add(f2_dbl(f2_add(t3, r5)), t3));
  r = z;
}
HD bool f12_is_one(const Fq12& a) {
  bool z = Fq::template sub<64, 1>(Fq::norm(a.c0.c0.c0), Fq::one()).is_zero_mod_p();
  z = z && a.c0.c0.c1.is_zero_mod_p();
  z = z && a.c0.c1.is_zero_mod_p() && a.c0.c2.is_zero_mod_p();
  z = z && a.c1.c0.is_zero_mod_p() && a.c1.c1.is_zero_mod_p() && a.c1.c2.is_zero_mod_p();
  return z;
}

// device-memory layout of an Fq12: 12 Fq coefficients of Fq::WORDS words, tower order
//   c0.c0.c0, c0.c0.c1, c0.c1.c0, c0.c1.c1, c0.c2.c0, c0.c2.c1, c1.c0.c0, ... (same order as the arkworks in-memory Fq12)
constexpr int FQ12_WORDS = 12 * Fq::WORDS;
HD void f12_store(uint32_t* p, const Fq12& a) {
  const Fq2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; i++) c[i]->store(p + i * Fq2::WORDS);
}
HD Fq12 f12_load(const uint32_t* p) {
  Fq12 a;
  Fq2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; i++) *c[i] = Fq2::load(p + i * Fq2::WORDS);
  return a;
}
HD void f12_to_ark(const Fq12& a, uint64_t* out72) {
  const Fq2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; i++) c[i]->to_ark(out72 + 12 * i);
}
HD Fq12 f12_from_ark(const uint64_t* in72) {
  Fq12 a;
  Fq2* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
  for (int i = 0; i < 6; i++) *c[i] = Fq2::from_ark(in72 + 12 * i);
  return a;
}

}  // namespace celo
