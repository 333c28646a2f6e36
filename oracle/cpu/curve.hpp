// ORACLE — test infrastructure only (see field.hpp header).
//
// CPU restatement of ark-ec 0.1.0 short-Weierstrass Jacobian arithmetic (a = 0),
// VariableBaseMSM::multi_scalar_mul and batch normalisation, as called by the
// reference at
//   crates/bls-crypto/src/bls/signature.rs:82-85  (G1: batch_normalization_into_affine + multi_scalar_mul)
//   crates/bls-crypto/src/bls/public.rs:58-61     (G2: same)
//   crates/bls-crypto/src/bls/signature.rs:61-67, public.rs:38-44 (aggregate = plain sums)
// Algorithms: SURVEY.md Appendix B.1 (Pippenger windowing rule) and B.6
// (dbl-2009-l, madd-2007-bl, add-2007-bl).
#pragma once
#include "field.hpp"
#include <vector>
#include <thread>
#include <atomic>

namespace orc {

template <class F> struct Affine {
  F x, y;
  bool inf;
  static Affine identity() { return {F::zero(), F::zero(), true}; }
  Affine neg() const { return {x, -y, inf}; }
};

template <class F> struct Jac {
  F x, y, z;  // identity: z == 0 (arkworks convention x = y = 1? arkworks uses (0,1,0))
  static Jac identity() { return {F::zero(), F::one(), F::zero()}; }
  bool is_identity() const { return z.is_zero(); }
  static Jac from_affine(const Affine<F>& a) {
    if (a.inf) return identity();
    return {a.x, a.y, F::one()};
  }
  // dbl-2009-l (a = 0)
  Jac dbl() const {
    if (is_identity()) return *this;
    F A = x.sqr(), B = y.sqr(), C = B.sqr();
    F D = ((x + B).sqr() - A - C).dbl();
    F E = A.dbl() + A;
    F Fv = E.sqr();
    Jac r;
    r.x = Fv - D.dbl();
    r.z = (y * z).dbl();
    r.y = E * (D - r.x) - C.dbl().dbl().dbl();
    return r;
  }
  // madd-2007-bl
  Jac add_mixed(const Affine<F>& q) const {
    if (q.inf) return *this;
    if (is_identity()) return from_affine(q);
    F Z1Z1 = z.sqr();
    F U2 = q.x * Z1Z1;
    F S2 = q.y * z * Z1Z1;
    if (x == U2 && y == S2) return dbl();
    F H = U2 - x;
    F HH = H.sqr();
    F I = HH.dbl().dbl();
    F J = H * I;
    F rr = (S2 - y).dbl();
    F V = x * I;
    Jac r;
    r.x = rr.sqr() - J - V.dbl();
    r.y = rr * (V - r.x) - (y * J).dbl();
    r.z = (z + H).sqr() - Z1Z1 - HH;
    return r;
  }
  // add-2007-bl
  Jac add(const Jac& o) const {
    if (is_identity()) return o;
    if (o.is_identity()) return *this;
    F Z1Z1 = z.sqr(), Z2Z2 = o.z.sqr();
    F U1 = x * Z2Z2, U2 = o.x * Z1Z1;
    F S1 = y * o.z * Z2Z2, S2 = o.y * z * Z1Z1;
    if (U1 == U2 && S1 == S2) return dbl();
    F H = U2 - U1;
    F I = H.dbl().sqr();
    F J = H * I;
    F rr = (S2 - S1).dbl();
    F V = U1 * I;
    Jac r;
    r.x = rr.sqr() - J - V.dbl();
    r.y = rr * (V - r.x) - (S1 * J).dbl();
    r.z = ((z + o.z).sqr() - Z1Z1 - Z2Z2) * H;
    return r;
  }
  Jac neg() const { return {x, -y, z}; }
  Affine<F> to_affine() const {
    if (is_identity()) return Affine<F>::identity();
    F zi = z.inverse();
    F zi2 = zi.sqr();
    return {x * zi2, y * zi2 * zi, false};
  }
  // double-and-add, MSB first (ark-ec AffineCurve::mul / ProjectiveCurve::mul)
  template <int M> Jac mul(const u64* k) const {
    Jac r = identity();
    for (int i = big_bits<M>(k) - 1; i >= 0; i--) {
      r = r.dbl();
      if (big_bit(k, i)) r = r.add(*this);
    }
    return r;
  }
  bool eq(const Jac& o) const {  // projective equality
    if (is_identity() || o.is_identity()) return is_identity() && o.is_identity();
    F Z1Z1 = z.sqr(), Z2Z2 = o.z.sqr();
    return x * Z2Z2 == o.x * Z1Z1 && y * o.z * Z2Z2 == o.y * z * Z1Z1;
  }
};

// ProjectiveCurve::batch_normalization_into_affine — Montgomery's trick over the z's
template <class F> static void batch_normalize(const Jac<F>* in, Affine<F>* out, size_t n) {
  std::vector<F> pre(n);
  F acc = F::one();
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    if (!in[i].is_identity()) acc = acc * in[i].z;
  }
  F ai = acc.inverse();
  for (size_t i = n; i-- > 0;) {
    if (in[i].is_identity()) { out[i] = Affine<F>::identity(); continue; }
    F zi = ai * pre[i];
    ai = ai * in[i].z;
    F zi2 = zi.sqr();
    out[i] = {in[i].x * zi2, in[i].y * zi2 * zi, false};
  }
}

static inline int ark_log2_ceil(size_t x) {  // ark_std::log2: ceil(log2(x)), 0 for x <= 1
  if (x <= 1) return 0;
  int b = 0;
  size_t v = x - 1;
  while (v) { b++; v >>= 1; }
  return b;
}

// VariableBaseMSM::multi_scalar_mul.
//   SL  = scalar limbs (4 for Fr of BLS12-377, 6 for Fr of BW6-761), canonical form
//   num_bits = Fr::MODULUS_BITS (253 / 377)
// One task per window (the rayon par_iter), here std::thread with `threads` workers.
template <class F, int SL>
static Jac<F> msm_pippenger(const Affine<F>* bases, const u64* scalars, size_t size, int num_bits, int threads) {
  int c = size < 32 ? 3 : (ark_log2_ceil(size) * 69 / 100) + 2;
  std::vector<int> starts;
  for (int w = 0; w < num_bits; w += c) starts.push_back(w);
  std::vector<Jac<F>> wsum(starts.size());
  u64 one_[SL];
  memset(one_, 0, sizeof one_);
  one_[0] = 1;
  auto window = [&](size_t wi) {
    int w_start = starts[wi];
    Jac<F> res = Jac<F>::identity();
    std::vector<Jac<F>> buckets((size_t(1) << c) - 1, Jac<F>::identity());
    for (size_t i = 0; i < size; i++) {
      const u64* s = scalars + i * SL;
      if (big_is_zero<SL>(s)) continue;
      if (big_cmp<SL>(s, one_) == 0) {
        if (w_start == 0) res = res.add_mixed(bases[i]);
        continue;
      }
      // digit = (scalar >> w_start) mod 2^c
      u64 d = 0;
      int limb = w_start >> 6, off = w_start & 63;
      d = s[limb] >> off;
      if (off + c > 64 && limb + 1 < SL) d |= s[limb + 1] << (64 - off);
      d &= (u64(1) << c) - 1;
      if (d) buckets[d - 1] = buckets[d - 1].add_mixed(bases[i]);
    }
    Jac<F> running = Jac<F>::identity();
    for (size_t b = buckets.size(); b-- > 0;) {
      running = running.add(buckets[b]);
      res = res.add(running);
    }
    wsum[wi] = res;
  };
  if (threads <= 1) {
    for (size_t wi = 0; wi < starts.size(); wi++) window(wi);
  } else {
    std::atomic<size_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
      th.emplace_back([&] {
        for (;;) {
          size_t wi = next.fetch_add(1);
          if (wi >= starts.size()) break;
          window(wi);
        }
      });
    for (auto& t : th) t.join();
  }
  // lowest + fold(rev(rest)): total = (((w_last)*2^c + w_{last-1})*2^c ... ) then + lowest
  Jac<F> lowest = wsum[0];
  Jac<F> total = Jac<F>::identity();
  for (size_t wi = starts.size(); wi-- > 1;) {
    total = total.add(wsum[wi]);
    for (int k = 0; k < c; k++) total = total.dbl();
  }
  return lowest.add(total);
}

// Naive definition: sum of double-and-add scalar muls (used to validate Pippenger)
template <class F, int SL> static Jac<F> msm_naive(const Affine<F>* bases, const u64* scalars, size_t n) {
  Jac<F> acc = Jac<F>::identity();
  for (size_t i = 0; i < n; i++) acc = acc.add(Jac<F>::from_affine(bases[i]).template mul<SL>(scalars + i * SL));
  return acc;
}

template <class F> static bool on_curve(const Affine<F>& p, const F& b) {
  if (p.inf) return true;
  return p.y.sqr() == p.x.sqr() * p.x + b;
}

}  // namespace orc
