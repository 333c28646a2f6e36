// Host-side field and XYZZ point arithmetic on 64-bit limbs, for the sequential epilogue of the big MSM only: the Horner
// recombination of the per-window results (c doublings + c additions per window, ~500 dependent point operations for a
// 253-bit scalar) is latency-bound on one host core, and the 28-bit-limb device representation (fp.h) costs ~60 ns per
// product there.  Elements are in Montgomery form with R = 2^(64 N) - arkworks' own representation, which is also what the
// last k_bitsum launches store - so no conversion happens on the host.  Plain CIOS with unsigned __int128; always fully
// reduced (no lazy bounds to track).  Formulas: dbl-2008-s-1 / add-2008-s as in curve.h.
#pragma once
#include <cstdint>
#include <cstring>
#include "fp.h"
#include "fp2.h"

namespace celo {

template <class P> struct HFp {
  static constexpr int N = P::N64;
  uint64_t v[N];

  struct Consts {
    uint64_t inv;       // -p^-1 mod 2^64
    uint64_t one[N];    // R mod p
    Consts() {
      uint64_t x = 1;   // Newton: x <- x (2 - p0 x) doubles the correct low bits
      for (int i = 0; i < 6; i++) x *= 2 - P::P64[0] * x;
      inv = (uint64_t)0 - x;
      Fp<P>::one().to_ark(one);
    }
  };
  static const Consts& C() { static const Consts c; return c; }

  static HFp zero() { HFp r; memset(r.v, 0, sizeof r.v); return r; }
  static HFp one() { HFp r; memcpy(r.v, C().one, sizeof r.v); return r; }
  static HFp load(const uint64_t* p) { HFp r; memcpy(r.v, p, sizeof r.v); return r; }
  void store(uint64_t* p) const { memcpy(p, v, sizeof v); }
  bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= v[i]; return a == 0; }

  static bool geq_p(const uint64_t* a) {
    for (int i = N - 1; i >= 0; i--) if (a[i] != P::P64[i]) return a[i] > P::P64[i];
    return true;
  }
  static void sub_p(uint64_t* a) {
    unsigned __int128 b = 0;
    for (int i = 0; i < N; i++) { unsigned __int128 d = (unsigned __int128)a[i] - P::P64[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 64) & 1; }
  }
  friend HFp operator+(const HFp& a, const HFp& b) {
    HFp r;
    unsigned __int128 c = 0;
    for (int i = 0; i < N; i++) { c += (unsigned __int128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_p(r.v)) sub_p(r.v);     // p < 2^(64N - 1) for both fields: no carry out, kept for generality
    return r;
  }
  friend HFp operator-(const HFp& a, const HFp& b) {
    HFp r;
    unsigned __int128 br = 0;
    for (int i = 0; i < N; i++) { unsigned __int128 d = (unsigned __int128)a.v[i] - b.v[i] - (uint64_t)br; r.v[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { unsigned __int128 c = 0; for (int i = 0; i < N; i++) { c += (unsigned __int128)r.v[i] + P::P64[i]; r.v[i] = (uint64_t)c; c >>= 64; } }
    return r;
  }
  HFp dbl() const { return *this + *this; }
  // CIOS Montgomery product, both inner loops fused (valid because the top limb of p has spare bits: the two carry
  // chains never overflow one word together - both moduli here are > 2 bits short of 64 N)
  friend HFp operator*(const HFp& a, const HFp& b) {
    const uint64_t inv = C().inv;
    uint64_t t[N];
    memset(t, 0, sizeof t);
#pragma GCC unroll 12
    for (int i = 0; i < N; i++) {
      unsigned __int128 c1 = (unsigned __int128)a.v[0] * b.v[i] + t[0];
      const uint64_t m = (uint64_t)c1 * inv;
      unsigned __int128 c2 = (unsigned __int128)m * P::P64[0] + (uint64_t)c1;
      c1 >>= 64;
      c2 >>= 64;
#pragma GCC unroll 12
      for (int j = 1; j < N; j++) {
        c1 += (unsigned __int128)a.v[j] * b.v[i] + t[j];
        c2 += (unsigned __int128)m * P::P64[j] + (uint64_t)c1;
        t[j - 1] = (uint64_t)c2;
        c1 >>= 64;
        c2 >>= 64;
      }
      t[N - 1] = (uint64_t)c1 + (uint64_t)c2;
    }
    HFp r;
    memcpy(r.v, t, sizeof r.v);
    if (geq_p(r.v)) sub_p(r.v);
    return r;
  }
  HFp sqr() const { return *this * *this; }
};

template <class P> struct HFp2 {   // Fp[u] / (u^2 + 5)
  typedef HFp<P> B;
  B c0, c1;
  static HFp2 zero() { return {B::zero(), B::zero()}; }
  static HFp2 one() { return {B::one(), B::zero()}; }
  static HFp2 load(const uint64_t* p) { return {B::load(p), B::load(p + B::N)}; }
  void store(uint64_t* p) const { c0.store(p); c1.store(p + B::N); }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  friend HFp2 operator+(const HFp2& a, const HFp2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  friend HFp2 operator-(const HFp2& a, const HFp2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  HFp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  static B mul5(const B& a) { const B a2 = a.dbl(), a4 = a2.dbl(); return a4 + a; }
  friend HFp2 operator*(const HFp2& a, const HFp2& b) {   // Karatsuba: (a0 b0 - 5 a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u
    const B v0 = a.c0 * b.c0, v1 = a.c1 * b.c1;
    return {v0 - mul5(v1), (a.c0 + a.c1) * (b.c0 + b.c1) - v0 - v1};
  }
  HFp2 sqr() const {                                        // (a0^2 - 5 a1^2) + 2 a0 a1 u
    const B v = c0 * c1;
    return {c0.sqr() - mul5(c1.sqr()), v.dbl()};
  }
};

template <class F> struct HostField;
template <class P> struct HostField<Fp<P>> { typedef HFp<P> type; };
template <class P> struct HostField<Fp2<P>> { typedef HFp2<P> type; };

template <class H> struct HXyzz {
  H X, Y, ZZ, ZZZ;   // identity: ZZ == 0
  static HXyzz identity() { return {H::zero(), H::zero(), H::zero(), H::zero()}; }
  bool is_identity() const { return ZZ.is_zero(); }
  static HXyzz load(const uint64_t* p, int stride) { return {H::load(p), H::load(p + stride), H::load(p + 2 * stride), H::load(p + 3 * stride)}; }
};
template <class H> inline HXyzz<H> hxyzz_dbl(const HXyzz<H>& a) {
  if (a.is_identity() || a.Y.is_zero()) return HXyzz<H>::identity();
  const H U = a.Y.dbl(), V = U.sqr(), W = U * V, S = a.X * V, X2 = a.X.sqr(), M = X2.dbl() + X2;
  const H X3 = M.sqr() - S.dbl();
  return {X3, M * (S - X3) - W * a.Y, V * a.ZZ, W * a.ZZZ};
}
template <class H> inline void hxyzz_add(HXyzz<H>& a, const HXyzz<H>& b) {
  if (b.is_identity()) return;
  if (a.is_identity()) { a = b; return; }
  const H U1 = a.X * b.ZZ, U2 = b.X * a.ZZ, S1 = a.Y * b.ZZZ, S2 = b.Y * a.ZZZ;
  const H Pd = U2 - U1, R = S2 - S1;
  if (Pd.is_zero()) {
    if (R.is_zero()) a = hxyzz_dbl(a);
    else a = HXyzz<H>::identity();
    return;
  }
  const H PP = Pd.sqr(), PPP = Pd * PP, Q = U1 * PP;
  const H X3 = R.sqr() - PPP - Q.dbl();
  a = {X3, R * (Q - X3) - S1 * PPP, a.ZZ * b.ZZ * PP, a.ZZZ * b.ZZZ * PPP};
}

// The Horner recombination of the big MSM's per-window results as a list of steps over point slots (msm.h builds the list):
// step k doubles the accumulator (unless bit 30 of order[k] is set), then adds slot order[k] & 0x3fffffff (order[k] = -1: no
// addend).  A slot is `stride` u64: X, Y, ZZ, ZZZ at multiples of `cs` words, arkworks form, ZZ = 0: identity.  host_ifma.cpp
// runs the same list eight products at a time where the CPU has AVX-512 IFMA.
constexpr int32_t HORNER_NODBL = 0x40000000;
template <class H> inline HXyzz<H> host64_horner(const uint64_t* pts, size_t stride, int cs, const int32_t* order, int steps) {
  HXyzz<H> acc = HXyzz<H>::identity();
  for (int k = 0; k < steps; k++) {
    if (order[k] < 0 || !(order[k] & HORNER_NODBL)) acc = hxyzz_dbl(acc);
    if (order[k] >= 0) hxyzz_add(acc, HXyzz<H>::load(pts + (size_t)(order[k] & 0x3fffffff) * stride, cs));
  }
  return acc;
}

}  // namespace celo
