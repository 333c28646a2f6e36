// ORACLE — test infrastructure only (see field.hpp header).
//
// CPU restatement of ark-ec 0.1.0 `models/bls12` and `models/bw6` pairing
// engines as called by the reference at
//   crates/bls-crypto/src/bls/public.rs:102      Bls12_377::product_of_pairings (2 pairs)
//   crates/bls-crypto/src/bls/signature.rs:149   Bls12_377::product_of_pairings (n+1 pairs)
//   crates/epoch-snark/src/api/verifier.rs:35    ark_groth16::verify_proof over BW6_761
// Algorithms: SURVEY.md Appendix B.2 (G2Prepared / ell / shared-squaring multi Miller
// loop), B.3 (BLS12 final exponentiation chain), B.4 (BW6-761 two-loop optimal ate,
// El Housni-Guillevic eprint 2020/351 final exponentiation), B.5 (Groth16 verify).
#pragma once
#include "curve.hpp"

namespace orc {

static const u64 BLS_X = 0x8508c00000000001ULL;

// ============================================================ BLS12-377
struct Bls12_377 {
  typedef Fq377 Fq;
  typedef Fq2_377 Fq2;
  typedef Fq12_377 Gt;
  typedef Affine<Fq> G1A;
  typedef Affine<Fq2> G2A;
  struct Ell { Fq2 c0, c1, c2; };
  struct G2Prepared { std::vector<Ell> coeffs; bool inf; };

  static Fq2 twist_b() {  // B' = 1/u = -u/5  ->  (0, -1/5)
    static Fq2 b = {Fq::zero(), -(Fq::from_u64(5).inverse())};
    return b;
  }
  static Fq g1_b() { return Fq::one(); }

  // ark-ec bls12/g2.rs doubling_step (homogeneous projective, D-twist output order)
  static Ell doubling_step(Fq2& rx, Fq2& ry, Fq2& rz, const Fq& two_inv) {
    Fq2 a = (rx * ry).mul_fp(two_inv);
    Fq2 b = ry.sqr();
    Fq2 c = rz.sqr();
    Fq2 e = twist_b() * (c.dbl() + c);
    Fq2 f = e.dbl() + e;
    Fq2 g = (b + f).mul_fp(two_inv);
    Fq2 h = (ry + rz).sqr() - (b + c);
    Fq2 i = e - b;
    Fq2 j = rx.sqr();
    Fq2 e_sq = e.sqr();
    rx = a * (b - f);
    ry = g.sqr() - (e_sq.dbl() + e_sq);
    rz = b * h;
    return {-h, j.dbl() + j, i};
  }
  static Ell addition_step(Fq2& rx, Fq2& ry, Fq2& rz, const G2A& q) {
    Fq2 theta = ry - q.y * rz;
    Fq2 lambda = rx - q.x * rz;
    Fq2 c = theta.sqr();
    Fq2 d = lambda.sqr();
    Fq2 e = lambda * d;
    Fq2 f = rz * c;
    Fq2 g = rx * d;
    Fq2 h = e + f - g.dbl();
    rx = lambda * h;
    ry = theta * (g - h) - e * ry;
    rz = rz * e;
    Fq2 j = theta * q.x - lambda * q.y;
    return {lambda, -theta, j};
  }
  static G2Prepared prepare(const G2A& q) {
    G2Prepared p;
    p.inf = q.inf;
    if (q.inf) return p;
    Fq two_inv = Fq::from_u64(2).inverse();
    Fq2 rx = q.x, ry = q.y, rz = Fq2::one();
    for (int i = 62; i >= 0; i--) {  // bits of X below the MSB (bit 63)
      p.coeffs.push_back(doubling_step(rx, ry, rz, two_inv));
      if ((BLS_X >> i) & 1) p.coeffs.push_back(addition_step(rx, ry, rz, q));
    }
    return p;
  }
  static void ell(Gt& f, const Ell& co, const G1A& p) {
    f = f.mul_by_034(co.c0.mul_fp(p.y), co.c1.mul_fp(p.x), co.c2);
  }
  static Gt miller_loop(const G1A* ps, const G2Prepared* qs, size_t n) {
    std::vector<size_t> live, pos;
    for (size_t k = 0; k < n; k++)
      if (!ps[k].inf && !qs[k].inf) { live.push_back(k); pos.push_back(0); }
    Gt f = Gt::one();
    for (int i = 62; i >= 0; i--) {
      f = f.sqr();
      for (size_t t = 0; t < live.size(); t++) ell(f, qs[live[t]].coeffs[pos[t]++], ps[live[t]]);
      if ((BLS_X >> i) & 1)
        for (size_t t = 0; t < live.size(); t++) ell(f, qs[live[t]].coeffs[pos[t]++], ps[live[t]]);
    }
    return f;  // X positive: no conjugation
  }
  static Gt exp_by_x(const Gt& f) {
    Gt r = Gt::one();
    for (int i = 63; i >= 0; i--) {
      r = r.cyclotomic_square();
      if ((BLS_X >> i) & 1) r = r * f;
    }
    return r;
  }
  static Gt final_exponentiation(const Gt& f) {
    Gt f1 = f.conj();
    Gt f2 = f.inverse();
    Gt r = f1 * f2;
    f2 = r;
    r = r.frob(2) * f2;
    // hard part (ark-ec bls12 final_exponentiation; computes the cube of the reduced pairing)
    Gt y0 = r.cyclotomic_square().conj();
    Gt y5 = exp_by_x(r);
    Gt y1 = y5.cyclotomic_square();
    Gt y3 = y0 * y5;
    y0 = exp_by_x(y3);
    Gt y2 = exp_by_x(y0);
    Gt y4 = exp_by_x(y2);
    y4 = y4 * y1;
    y1 = exp_by_x(y4);
    y3 = y3.conj();
    y1 = y1 * y3;
    y1 = y1 * r;
    y3 = r.conj();
    y0 = y0 * r;
    y0 = y0.frob(3);
    y4 = y4 * y3;
    y4 = y4.frob(1);
    y5 = y5 * y2;
    y5 = y5.frob(2);
    y5 = y5 * y0;
    y5 = y5 * y4;
    y5 = y5 * y1;
    return y5;
  }
  static Gt product_of_pairings(const G1A* ps, const G2A* qs, size_t n) {
    std::vector<G2Prepared> prep(n);
    for (size_t i = 0; i < n; i++) prep[i] = prepare(qs[i]);
    return final_exponentiation(miller_loop(ps, prep.data(), n));
  }
};

// ============================================================ BW6-761
struct Bw6_761 {
  typedef Fq761 Fq;
  typedef Fq6_761 Gt;
  typedef Affine<Fq> G1A;
  typedef Affine<Fq> G2A;
  struct Ell { Fq c0, c1, c2; };
  struct G2Prepared { std::vector<Ell> c1, c2; bool inf; };
  static Fq g1_b() { return -Fq::one(); }
  static Fq g2_b() { return Fq::from_u64(4); }

  // loop 2 digits: signed binary (NAF) expansion of x^3 - x^2 - x, little-endian, derived not recalled
  static const std::vector<int>& loop2() {
    static std::vector<int> d = [] {
      u128 x = BLS_X;
      // n = x^3 - x^2 - x  (190 bits) as 4 x u64
      u64 n[4] = {0, 0, 0, 0};
      auto mul_small = [](u64* a, u64 m) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)a[i] * m; a[i] = (u64)c; c >>= 64; }
      };
      u64 x1[4] = {(u64)x, 0, 0, 0};
      u64 x2[4] = {(u64)x, 0, 0, 0};
      mul_small(x2, (u64)x);
      u64 x3[4];
      memcpy(x3, x2, sizeof x3);
      mul_small(x3, (u64)x);
      big_sub<4>(n, x3, x2);
      big_sub<4>(n, n, x1);
      std::vector<int> out;
      u64 one_[4] = {1, 0, 0, 0};
      while (!big_is_zero<4>(n)) {
        int di = 0;
        if (n[0] & 1) {
          di = 2 - (int)(n[0] & 3);  // 1 or -1
          if (di == 1) big_sub<4>(n, n, one_);
          else big_add<4>(n, n, one_);
        }
        out.push_back(di);
        big_div_small<4>(n, n, 2);
      }
      return out;
    }();
    return d;
  }
  static const u64 LOOP1 = 0x8508c00000000002ULL;  // x + 1

  // ark-ec bw6/g2.rs doubling_step / addition_step (M-twist output order)
  static Ell doubling_step(Fq& rx, Fq& ry, Fq& rz) {
    Fq a = rx * ry;
    Fq b = ry.sqr();
    Fq b4 = b.dbl().dbl();
    Fq c = rz.sqr();
    Fq e = g2_b() * (c.dbl() + c);
    Fq f = e.dbl() + e;
    Fq g = b + f;
    Fq h = (ry + rz).sqr() - (b + c);
    Fq i = e - b;
    Fq j = rx.sqr();
    Fq e2_sq = e.dbl().sqr();
    rx = a.dbl() * (b - f);
    ry = g.sqr() - (e2_sq.dbl() + e2_sq);
    rz = b4 * h;
    return {i, j.dbl() + j, -h};
  }
  static Ell addition_step(Fq& rx, Fq& ry, Fq& rz, const G2A& q) {
    Fq theta = ry - q.y * rz;
    Fq lambda = rx - q.x * rz;
    Fq c = theta.sqr();
    Fq d = lambda.sqr();
    Fq e = lambda * d;
    Fq f = rz * c;
    Fq g = rx * d;
    Fq h = e + f - g.dbl();
    rx = lambda * h;
    ry = theta * (g - h) - e * ry;
    rz = rz * e;
    Fq j = theta * q.x - lambda * q.y;
    return {j, -theta, lambda};
  }
  static G2Prepared prepare(const G2A& q) {
    G2Prepared p;
    p.inf = q.inf;
    if (q.inf) return p;
    Fq rx = q.x, ry = q.y, rz = Fq::one();
    for (int i = 62; i >= 0; i--) {
      p.c1.push_back(doubling_step(rx, ry, rz));
      if ((LOOP1 >> i) & 1) p.c1.push_back(addition_step(rx, ry, rz, q));
    }
    rx = q.x; ry = q.y; rz = Fq::one();
    G2A nq = q.neg();
    const std::vector<int>& d = loop2();
    for (size_t i = d.size() - 1; i >= 1; i--) {
      p.c2.push_back(doubling_step(rx, ry, rz));
      int bit = d[i - 1];
      if (bit == 1) p.c2.push_back(addition_step(rx, ry, rz, q));
      else if (bit == -1) p.c2.push_back(addition_step(rx, ry, rz, nq));
    }
    return p;
  }
  static void ell(Gt& f, const Ell& co, const G1A& p) {
    f = f.mul_by_014(co.c0, co.c1 * p.x, co.c2 * p.y);
  }
  static Gt miller_loop(const G1A* ps, const G2Prepared* qs, size_t n) {
    std::vector<size_t> live;
    for (size_t k = 0; k < n; k++)
      if (!ps[k].inf && !qs[k].inf) live.push_back(k);
    std::vector<size_t> pos(live.size(), 0);
    Gt f1 = Gt::one();
    for (int i = 62; i >= 0; i--) {
      f1 = f1.sqr();
      for (size_t t = 0; t < live.size(); t++) ell(f1, qs[live[t]].c1[pos[t]++], ps[live[t]]);
      if ((LOOP1 >> i) & 1)
        for (size_t t = 0; t < live.size(); t++) ell(f1, qs[live[t]].c1[pos[t]++], ps[live[t]]);
    }
    std::fill(pos.begin(), pos.end(), 0);
    Gt f2 = Gt::one();
    const std::vector<int>& d = loop2();
    for (size_t i = d.size() - 1; i >= 1; i--) {
      if (i != d.size() - 1) f2 = f2.sqr();
      for (size_t t = 0; t < live.size(); t++) ell(f2, qs[live[t]].c2[pos[t]++], ps[live[t]]);
      if (d[i - 1] != 0)
        for (size_t t = 0; t < live.size(); t++) ell(f2, qs[live[t]].c2[pos[t]++], ps[live[t]]);
    }
    return f1 * f2.frob(1);
  }
  // f^(poly in x) helpers for the hard part: exponent given as big integer (little-endian limbs), sign separately
  template <int M> static Gt pow_cyc(const Gt& f, const u64* e, bool neg) {
    Gt r = Gt::one();
    for (int i = big_bits<M>(e) - 1; i >= 0; i--) {
      r = r.sqr();
      if (big_bit(e, i)) r = r * f;
    }
    return neg ? r.conj() : r;  // inverse == conjugate in the cyclotomic subgroup
  }
  // evaluate sum coeff[i] * x^i exactly into (magnitude, sign); 10 limbs are enough for deg 9
  static void poly_at_x(const long* co, int deg, u64* mag, bool& neg) {
    // Horner over signed 640-bit two's complement emulated with magnitude/sign
    u64 acc[10];
    memset(acc, 0, sizeof acc);
    bool aneg = false;
    for (int i = deg; i >= 0; i--) {
      // acc = acc * x
      u128 c = 0;
      for (int k = 0; k < 10; k++) { c += (u128)acc[k] * BLS_X; acc[k] = (u64)c; c >>= 64; }
      // acc += co[i]
      long ci = co[i];
      bool cneg = ci < 0;
      u64 cm[10];
      memset(cm, 0, sizeof cm);
      cm[0] = (u64)(cneg ? -ci : ci);
      if (aneg == cneg) big_add<10>(acc, acc, cm);
      else if (big_cmp<10>(acc, cm) >= 0) big_sub<10>(acc, acc, cm);
      else { u64 t[10]; big_sub<10>(t, cm, acc); memcpy(acc, t, sizeof t); aneg = cneg; }
      if (big_is_zero<10>(acc)) aneg = false;
    }
    memcpy(mag, acc, sizeof acc);
    neg = aneg;
  }
  static Gt final_exponentiation(const Gt& f) {
    // easy part: (q^3 - 1)(q + 1)
    Gt inv = f.inverse();
    Gt a = f.conj() * inv;
    Gt m = a.frob(1) * a;
    // hard part: m^(R0(x)) * (m^q)^(R1(x))   (eprint 2020/351 Alg. 6 polynomials, as in ark-ec bw6)
    static const long R0[8] = {-220, -263, -73, -314, -197, 269, 70, -103};
    static const long R1[10] = {229, 34, -181, 452, -65, -445, 492, 77, -276, 103};
    u64 e0[10], e1[10];
    bool n0, n1;
    poly_at_x(R0, 7, e0, n0);
    poly_at_x(R1, 9, e1, n1);
    Gt p0 = pow_cyc<10>(m, e0, n0);
    Gt p1 = pow_cyc<10>(m.frob(1), e1, n1);
    return p0 * p1;
  }
  static Gt product_of_pairings(const G1A* ps, const G2A* qs, size_t n) {
    std::vector<G2Prepared> prep(n);
    for (size_t i = 0; i < n; i++) prep[i] = prepare(qs[i]);
    return final_exponentiation(miller_loop(ps, prep.data(), n));
  }
};

}  // namespace orc
