// Pairing towers over the lazy 28-bit-limb fields, as ONE "cubic over a base, then quadratic over the cubic" template:
//   BLS12-377:  base = Fq2 = Fq[u]/(u^2+5), Fq6 = Fq2[v]/(v^3 - u), Fq12 = Fq6[w]/(w^2 - v)
//   BW6-761  :  base = Fq,                  Fq3 = Fq[u]/(u^3 + 4),  Fq6  = Fq3[v]/(v^2 - u)
// Replaces ark-ff's Fp6_3over2/Fp12_2over6 (BLS12-377) and Fp3/Fp6_2over3 (BW6-761) as used by ark-ec's pairing engines,
// which the reference enters through Bls12_377::product_of_pairings (crates/bls-crypto/src/bls/public.rs:102,
// signature.rs:149) and ark_groth16::verify_proof over BW6_761 (crates/epoch-snark/src/api/verifier.rs:35).
//
// Value-growth discipline (fp.h contract): every cubic/quadratic-level result is passed through Fp::wred (value < ~2.1p,
// normalised limbs), base products are < 3p; inside a function additions/subtractions are lazy and every multiply input
// stays below ~80p (checked at run time by the CELO_FP_TRACK host build, tests/test_host_tower.py).
//
// The base-multiply-and-up functions are deliberately NOT inlined on the device: a BLS12-377 Fq12 multiplication is 36
// Montgomery passes (~22k instructions); operands live in per-lane private memory and the code stays tens of KB.
#pragma once
#include "fp2.h"
#include "lanes.h"

namespace celo {


#if defined(__HIPCC__)
#define TW_FN __host__ __device__ inline __attribute__((noinline))
#else
#define TW_FN inline
#endif

// ------------------------------------------------------------------ base-field policies
struct Base377 {  // base = Fq2, cubic non-residue xi = u
  typedef Fq2 T;
  HD static T zero() { return Fq2::zero(); }
  HD static T one() { return Fq2::one(); }
  TW_FN static void mul(T& r, const T& a, const T& b) { r = Fq2::mul(a, b); }
  TW_FN static void sqr(T& r, const T& a) { r = Fq2::sqr(a); }
  TW_FN static void inv(T& r, const T& a) { r = Fq2::inv(a); }
  HD static T mul_inl(const T& a, const T& b) { return Fq2::mul(a, b); }   // inlined variants for the lane-parallel kernels
  HD static T inv_inl(const T& a) { return Fq2::inv(a); }
  HD static T half(const T& a) { return {Fq::half(a.c0), Fq::half(a.c1)}; }
  HD static T add(const T& a, const T& b) { return Fq2::norm(Fq2::add(a, b)); }
  HD static T dbl(const T& a) { return Fq2::norm(Fq2::add(a, a)); }
  HD static T tpl(const T& a) { return Fq2::norm(Fq2::add(Fq2::add(a, a), a)); }
  template <int K> HD static T sub(const T& a, const T& b) { return Fq2::norm(Fq2::template sub<K, 1>(a, b)); }
  template <int K> HD static T neg(const T& a) { return Fq2::norm(Fq2::template neg<K, 1>(a)); }
  HD static T mul_nr(const T& a) { return Fq2::mul_by_u(a); }  // needs vb(a.c1) <= 12; result vb <= 64
  HD static T wred(const T& a) { return {Fq::wred(a.c0), Fq::wred(a.c1)}; }
  HD static bool is_zero(const T& a) { return a.is_zero_mod_p(); }
  HD static bool is_one(const T& a) {
    return Fq::template sub<64, 1>(Fq::norm(a.c0), Fq::one()).is_zero_mod_p() && a.c1.is_zero_mod_p();
  }
  static constexpr int WORDS = Fq2::WORDS;
  static constexpr int ARK64 = 12;
  HD static T load(const uint32_t* p) { return Fq2::load(p); }
  HD static void store(uint32_t* p, const T& a) { a.store(p); }
  HD static T from_ark(const uint64_t* s) { return Fq2::from_ark(s); }
  HD static void to_ark(const T& a, uint64_t* d) { a.to_ark(d); }
};
struct Base761 {  // base = Fq (761 bits), cubic non-residue -4
  typedef Fw T;
  HD static T zero() { return Fw::zero(); }
  HD static T one() { return Fw::one(); }
  TW_FN static void mul(T& r, const T& a, const T& b) { r = Fw::mul(a, b); }
  TW_FN static void sqr(T& r, const T& a) { r = Fw::sqr(a); }
  TW_FN static void inv(T& r, const T& a) { r = Fw::inv(a); }
  HD static T mul_inl(const T& a, const T& b) { return Fw::mul(a, b); }
  HD static T inv_inl(const T& a) { return Fw::inv(a); }
  HD static T half(const T& a) { return Fw::half(a); }
  HD static T add(const T& a, const T& b) { return Fw::norm(Fw::add(a, b)); }
  HD static T dbl(const T& a) { return Fw::norm(Fw::add(a, a)); }
  HD static T tpl(const T& a) { return Fw::norm(Fw::add(Fw::add(a, a), a)); }
  template <int K> HD static T sub(const T& a, const T& b) { return Fw::norm(Fw::template sub<K, 1>(a, b)); }
  template <int K> HD static T neg(const T& a) { return Fw::norm(Fw::template neg<K, 1>(a)); }
  HD static T mul_nr(const T& a) {  // -4a; needs vb(a) <= 16; result vb <= 64
    T t = Fw::norm(a);
    T q = Fw::norm(Fw::dbl(Fw::dbl(t)));
    return Fw::norm(Fw::template neg<64, 1>(q));
  }
  HD static T wred(const T& a) { return Fw::wred(a); }
  HD static bool is_zero(const T& a) { return a.is_zero_mod_p(); }
  HD static bool is_one(const T& a) { return Fw::template sub<64, 1>(Fw::norm(a), Fw::one()).is_zero_mod_p(); }
  static constexpr int WORDS = Fw::WORDS;
  static constexpr int ARK64 = 12;
  HD static T load(const uint32_t* p) { return Fw::load(p); }
  HD static void store(uint32_t* p, const T& a) { a.store(p); }
  HD static T from_ark(const uint64_t* s) { return Fw::from_ark(s); }
  HD static void to_ark(const T& a, uint64_t* d) { a.to_ark(d); }
};

template <class BP> struct Cubic { typename BP::T c0, c1, c2; };
template <class BP> struct Quad { Cubic<BP> c0, c1; };

// ------------------------------------------------------------------ cubic extension  B[g]/(g^3 - nr)
template <class BP> HD Cubic<BP> cub_zero() { return {BP::zero(), BP::zero(), BP::zero()}; }
template <class BP> HD Cubic<BP> cub_one() { return {BP::one(), BP::zero(), BP::zero()}; }
template <class BP> HD Cubic<BP> cub_add(const Cubic<BP>& a, const Cubic<BP>& b) {
  return {BP::add(a.c0, b.c0), BP::add(a.c1, b.c1), BP::add(a.c2, b.c2)};
}
template <int K, class BP> HD Cubic<BP> cub_sub(const Cubic<BP>& a, const Cubic<BP>& b) {
  return {BP::template sub<K>(a.c0, b.c0), BP::template sub<K>(a.c1, b.c1), BP::template sub<K>(a.c2, b.c2)};
}
template <class BP> HD Cubic<BP> cub_wred(const Cubic<BP>& a) { return {BP::wred(a.c0), BP::wred(a.c1), BP::wred(a.c2)}; }
template <class BP> HD Cubic<BP> cub_neg(const Cubic<BP>& a) {  // a clean (vb <= 4)
  return cub_wred<BP>({BP::template neg<4>(a.c0), BP::template neg<4>(a.c1), BP::template neg<4>(a.c2)});
}
// multiplication by the generator g of the cubic extension: (nr*c2, c0, c1); needs vb(a.c2) <= 12
template <class BP> HD Cubic<BP> cub_mul_by_gen(const Cubic<BP>& a) { return {BP::mul_nr(a.c2), a.c0, a.c1}; }

// r = a*b; inputs vb <= 40, output wred'ed
template <class BP> TW_FN void cub_mul(Cubic<BP>& r, const Cubic<BP>& a, const Cubic<BP>& b) {
  typedef typename BP::T T;
  T v0, v1, v2, t;
  BP::mul(v0, a.c0, b.c0);
  BP::mul(v1, a.c1, b.c1);
  BP::mul(v2, a.c2, b.c2);
  BP::mul(t, BP::add(a.c1, a.c2), BP::add(b.c1, b.c2));
  t = BP::template sub<4>(BP::template sub<4>(t, v1), v2);  // vb <= 11
  T r0 = BP::wred(BP::add(v0, BP::mul_nr(t)));
  BP::mul(t, BP::add(a.c0, a.c1), BP::add(b.c0, b.c1));
  t = BP::template sub<4>(BP::template sub<4>(t, v0), v1);
  T r1 = BP::wred(BP::add(t, BP::mul_nr(v2)));
  BP::mul(t, BP::add(a.c0, a.c2), BP::add(b.c0, b.c2));
  t = BP::template sub<4>(BP::template sub<4>(t, v0), v2);
  T r2 = BP::wred(BP::add(t, v1));
  r.c0 = r0; r.c1 = r1; r.c2 = r2;
}
// r = x * (b0 + b1 g)
template <class BP> TW_FN void cub_mul_by_01(Cubic<BP>& r, const Cubic<BP>& x, const typename BP::T& b0, const typename BP::T& b1) {
  typedef typename BP::T T;
  T p, q;
  BP::mul(p, x.c0, b0);
  BP::mul(q, x.c2, b1);
  T t0 = BP::wred(BP::add(p, BP::mul_nr(q)));
  BP::mul(p, x.c0, b1);
  BP::mul(q, x.c1, b0);
  T t1 = BP::wred(BP::add(p, q));
  BP::mul(p, x.c1, b1);
  BP::mul(q, x.c2, b0);
  T t2 = BP::wred(BP::add(p, q));
  r.c0 = t0; r.c1 = t1; r.c2 = t2;
}
template <class BP> TW_FN void cub_inv(Cubic<BP>& r, const Cubic<BP>& a) {
  typedef typename BP::T T;
  T s, m, t0, t1, t2, d;
  BP::sqr(s, a.c0); BP::mul(m, a.c1, a.c2);
  t0 = BP::wred(BP::template sub<64>(s, BP::mul_nr(m)));
  BP::sqr(s, a.c2); BP::mul(m, a.c0, a.c1);
  t1 = BP::wred(BP::template sub<4>(BP::mul_nr(s), m));
  BP::sqr(s, a.c1); BP::mul(m, a.c0, a.c2);
  t2 = BP::wred(BP::template sub<4>(s, m));
  BP::mul(d, a.c0, t0);
  BP::mul(m, a.c2, t1); d = BP::add(d, BP::mul_nr(m));
  BP::mul(m, a.c1, t2); d = BP::wred(BP::add(BP::wred(d), BP::mul_nr(m)));
  T di;
  BP::inv(di, d);
  BP::mul(r.c0, t0, di);
  BP::mul(r.c1, t1, di);
  BP::mul(r.c2, t2, di);
}

// ------------------------------------------------------------------ quadratic extension  C[h]/(h^2 - g)
template <class BP> HD Quad<BP> quad_one() { return {cub_one<BP>(), cub_zero<BP>()}; }
template <class BP> TW_FN void quad_mul(Quad<BP>& r, const Quad<BP>& a, const Quad<BP>& b) {
  Cubic<BP> v0, v1, t;
  cub_mul(v0, a.c0, b.c0);
  cub_mul(v1, a.c1, b.c1);
  cub_mul(t, cub_add(a.c0, a.c1), cub_add(b.c0, b.c1));
  r.c1 = cub_wred(cub_sub<4>(cub_sub<4>(t, v0), v1));
  r.c0 = cub_wred(cub_add(v0, cub_mul_by_gen(v1)));
}
template <class BP> TW_FN void quad_sqr(Quad<BP>& r, const Quad<BP>& a) {
  Cubic<BP> ab, t;
  cub_mul(ab, a.c0, a.c1);
  Cubic<BP> s2 = cub_wred(cub_add(a.c0, cub_mul_by_gen(a.c1)));
  cub_mul(t, cub_add(a.c0, a.c1), s2);
  Cubic<BP> vab = cub_mul_by_gen(ab);
  Cubic<BP> c0 = cub_sub<4>(t, ab);
  c0 = {BP::template sub<64>(c0.c0, vab.c0), BP::template sub<4>(c0.c1, vab.c1), BP::template sub<4>(c0.c2, vab.c2)};
  r.c0 = cub_wred(c0);
  r.c1 = cub_wred(cub_add(ab, ab));
}
template <class BP> HD Quad<BP> quad_conj(const Quad<BP>& a) { return {a.c0, cub_neg(a.c1)}; }
template <class BP> TW_FN void quad_inv(Quad<BP>& r, const Quad<BP>& a) {
  Cubic<BP> s0, s1, d, di;
  cub_mul(s0, a.c0, a.c0);
  cub_mul(s1, a.c1, a.c1);
  Cubic<BP> vs1 = cub_mul_by_gen(s1);
  d = cub_wred<BP>({BP::template sub<64>(s0.c0, vs1.c0), BP::template sub<4>(s0.c1, vs1.c1), BP::template sub<4>(s0.c2, vs1.c2)});
  cub_inv(di, d);
  cub_mul(r.c0, a.c0, di);
  Cubic<BP> m;
  cub_mul(m, a.c1, di);
  r.c1 = cub_neg(m);
}
template <class BP> HD bool quad_is_one(const Quad<BP>& a) {
  bool z = BP::is_one(a.c0.c0);
  z = z && BP::is_zero(a.c0.c1) && BP::is_zero(a.c0.c2);
  z = z && BP::is_zero(a.c1.c0) && BP::is_zero(a.c1.c1) && BP::is_zero(a.c1.c2);
  return z;
}
// device-memory / arkworks layouts: the six base coefficients in tower order c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2
// (the arkworks in-memory order of Fq12 / Fq6); 72 u64 either way.
template <class BP> struct QuadIO {
  static constexpr int WORDS = 6 * BP::WORDS;
  HD static void store(uint32_t* p, const Quad<BP>& a) {
    const typename BP::T* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    for (int i = 0; i < 6; i++) BP::store(p + i * BP::WORDS, *c[i]);
  }
  HD static Quad<BP> load(const uint32_t* p) {
    Quad<BP> a;
    typename BP::T* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    for (int i = 0; i < 6; i++) *c[i] = BP::load(p + i * BP::WORDS);
    return a;
  }
  HD static void to_ark(const Quad<BP>& a, uint64_t* out72) {
    const typename BP::T* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    for (int i = 0; i < 6; i++) BP::to_ark(*c[i], out72 + 12 * i);
  }
  HD static Quad<BP> from_ark(const uint64_t* in72) {
    Quad<BP> a;
    typename BP::T* c[6] = {&a.c0.c0, &a.c0.c1, &a.c0.c2, &a.c1.c0, &a.c1.c1, &a.c1.c2};
    for (int i = 0; i < 6; i++) *c[i] = BP::from_ark(in72 + 12 * i);
    return a;
  }
};

// ================================================================== BLS12-377 names and curve-specific pieces
typedef Cubic<Base377> Fq6;
typedef Quad<Base377> Fq12;
HD Fq2 f2_wred(const Fq2& a) { return Base377::wred(a); }
HD Fq2 f2_add(const Fq2& a, const Fq2& b) { return Base377::add(a, b); }
HD Fq2 f2_dbl(const Fq2& a) { return Base377::dbl(a); }
HD Fq2 f2_tpl(const Fq2& a) { return Base377::tpl(a); }
template <int K> HD Fq2 f2_sub(const Fq2& a, const Fq2& b) { return Base377::sub<K>(a, b); }
template <int K> HD Fq2 f2_neg(const Fq2& a) { return Base377::neg<K>(a); }
HD Fq2 f2_mul_xi(const Fq2& a) { return Base377::mul_nr(a); }
HD Fq2 f2_from(const uint32_t* c0, const uint32_t* c1) { return {Fq::from_limbs(c0), Fq::from_limbs(c1)}; }
HD void f2_mul(Fq2& r, const Fq2& a, const Fq2& b) { Base377::mul(r, a, b); }
HD void f2_sqr(Fq2& r, const Fq2& a) { Base377::sqr(r, a); }
TW_FN void f2_mul_fp(Fq2& r, const Fq2& a, const Fq& k) { r = Fq2::mul_fp(Fq2::norm(a), k); }
HD Fq12 f12_one() { return quad_one<Base377>(); }
HD void f12_mul(Fq12& r, const Fq12& a, const Fq12& b) { quad_mul(r, a, b); }
HD void f12_sqr(Fq12& r, const Fq12& a) { quad_sqr(r, a); }
HD Fq12 f12_conj(const Fq12& a) { return quad_conj(a); }
HD void f12_inv(Fq12& r, const Fq12& a) { quad_inv(r, a); }
HD bool f12_is_one(const Fq12& a) { return quad_is_one(a); }
constexpr int FQ12_WORDS = QuadIO<Base377>::WORDS;
HD void f12_store(uint32_t* p, const Fq12& a) { QuadIO<Base377>::store(p, a); }
HD Fq12 f12_load(const uint32_t* p) { return QuadIO<Base377>::load(p); }
HD void f12_to_ark(const Fq12& a, uint64_t* o) { QuadIO<Base377>::to_ark(a, o); }
HD Fq12 f12_from_ark(const uint64_t* i) { return QuadIO<Base377>::from_ark(i); }

// f *= s0 + (s3 + s4 v) w        (ark-ff Fp12::mul_by_034; D-twist line placement)
TW_FN void f12_mul_by_034(Fq12& f, const Fq2& s0, const Fq2& s3, const Fq2& s4) {
  Fq6 a, b, e;
  f2_mul(a.c0, f.c0.c0, s0);
  f2_mul(a.c1, f.c0.c1, s0);
  f2_mul(a.c2, f.c0.c2, s0);
  cub_mul_by_01(b, f.c1, s3, s4);
  cub_mul_by_01(e, cub_add(f.c0, f.c1), f2_add(s0, s3), s4);
  f.c1 = cub_wred(cub_sub<4>(cub_sub<4>(e, a), b));
  f.c0 = cub_wred(cub_add(a, cub_mul_by_gen(b)));
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

// ================================================================== BW6-761 names and curve-specific pieces
typedef Cubic<Base761> Fw3;
typedef Quad<Base761> Fw6;
// f *= (s0 + s1 u) + (s4 u) v       (ark-ff Fp6_2over3::mul_by_014; M-twist line placement)
TW_FN void fw6_mul_by_014(Fw6& f, const Fw& s0, const Fw& s1, const Fw& s4) {
  Fw6 o = {{s0, s1, Fw::zero()}, {Fw::zero(), s4, Fw::zero()}};
  Fw6 t;
  quad_mul(t, f, o);
  f = t;
}
// a^q: basis element u^a v^b = w^(2a+b) (w = v) is scaled by h^(2a+b), h = (-4)^((q-1)/6) in Fq
TW_FN void fw6_frob1(Fw6& r, const Fw6& a) {
  Fw6 o;
  o.c0.c0 = a.c0.c0;
  Base761::mul(o.c0.c1, a.c0.c1, Fw::from_limbs(T761::FROB1_2));
  Base761::mul(o.c0.c2, a.c0.c2, Fw::from_limbs(T761::FROB1_4));
  Base761::mul(o.c1.c0, a.c1.c0, Fw::from_limbs(T761::FROB1_1));
  Base761::mul(o.c1.c1, a.c1.c1, Fw::from_limbs(T761::FROB1_3));
  Base761::mul(o.c1.c2, a.c1.c2, Fw::from_limbs(T761::FROB1_5));
  r = o;
}

}  // namespace celo
