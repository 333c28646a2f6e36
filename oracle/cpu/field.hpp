// ORACLE — test infrastructure only.  Nothing under celo-bls-snark-rs_amd/ may
// include, link or call this; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use it (as the checker / the CPU baseline).
//
// CPU restatement of the field arithmetic the reference gets from ark-ff 0.1.0
// (arkworks-rs/algebra#8d76d181, Cargo.lock:138-140 of the reference; source not
// vendored): Montgomery prime fields with 64-bit limbs and R = 2^(64*N) — the
// same radix arkworks uses, so Montgomery-form limbs are interchangeable — and
// the extension towers of ark-bls12-377 / ark-bw6-761
// (arkworks-rs/curves#6ed2450b):
//   BLS12-377: Fq2 = Fq[u]/(u^2+5), Fq6 = Fq2[v]/(v^3-u), Fq12 = Fq6[w]/(w^2-v)
//   BW6-761 :  Fq3 = Fq[u]/(u^3+4), Fq6 = Fq3[v]/(v^2-u)
// All derived constants (R, R^2, -p^-1, Frobenius coefficients) are COMPUTED at
// start-up from the moduli, never recalled; the moduli themselves are pinned
// against the reference's golden vectors (tests/test_oracle_golden.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>

namespace orc {
typedef uint64_t u64;
typedef unsigned __int128 u128;

// ---------------------------------------------------------------- bigint utils
template <int N> static inline int big_cmp(const u64* a, const u64* b) {
  for (int i = N - 1; i >= 0; i--) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}
template <int N> static inline u64 big_add(u64* r, const u64* a, const u64* b) {
  u128 c = 0;
  for (int i = 0; i < N; i++) { c += (u128)a[i] + b[i]; r[i] = (u64)c; c >>= 64; }
  return (u64)c;
}
template <int N> static inline u64 big_sub(u64* r, const u64* a, const u64* b) {
  u64 br = 0;
  for (int i = 0; i < N; i++) {
    u128 t = (u128)a[i] - b[i] - br;
    r[i] = (u64)t;
    br = (u64)(t >> 64) & 1;
  }
  return br;
}
template <int N> static inline bool big_is_zero(const u64* a) {
  u64 o = 0;
  for (int i = 0; i < N; i++) o |= a[i];
  return o == 0;
}
static inline int hexval(char c) {
  if (c >= '0' && c <= '9') return c - '0';
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  if (c >= 'A' && c <= 'F') return c - 'A' + 10;
  return -1;
}
template <int N> static inline void big_from_hex(u64* r, const char* s) {
  memset(r, 0, 8 * N);
  int len = (int)strlen(s);
  for (int i = 0; i < len; i++) {
    int v = hexval(s[len - 1 - i]);
    r[i / 16] |= (u64)v << (4 * (i % 16));
  }
}
// r = a / d (small d), returns remainder
template <int N> static inline u64 big_div_small(u64* r, const u64* a, u64 d) {
  u128 rem = 0;
  for (int i = N - 1; i >= 0; i--) {
    u128 cur = (rem << 64) | a[i];
    r[i] = (u64)(cur / d);
    rem = cur % d;
  }
  return (u64)rem;
}
template <int N> static inline int big_bits(const u64* a) {
  for (int i = N - 1; i >= 0; i--)
    if (a[i]) return 64 * i + 64 - __builtin_clzll(a[i]);
  return 0;
}
static inline bool big_bit(const u64* a, int i) { return (a[i >> 6] >> (i & 63)) & 1; }

// ---------------------------------------------------------------- prime field
template <int N_> struct FpConsts {
  u64 p[N_], r[N_], r2[N_], inv;  // inv = -p^{-1} mod 2^64
  u64 pm1_half[N_];               // (p-1)/2 canonical, for lexicographic "y > -y"
  int bits;
};

template <class Tag> struct Fp {
  static constexpr int N = Tag::N;
  u64 v[N];
  static FpConsts<N>& C() {
    static FpConsts<N> c = make_consts();
    return c;
  }
  static FpConsts<N> make_consts() {
    FpConsts<N> c;
    big_from_hex<N>(c.p, Tag::hex());
    c.bits = big_bits<N>(c.p);
    u64 x = 1;  // Newton: x = p^-1 mod 2^64
    for (int i = 0; i < 6; i++) x *= 2 - c.p[0] * x;
    c.inv = (u64)0 - x;
    // r = 2^(64N) mod p by doubling 1; r2 = 2^(128N) mod p
    u64 t[N];
    memset(t, 0, sizeof t);
    t[0] = 1;
    for (int i = 0; i < 128 * N; i++) {
      u64 carry = big_add<N>(t, t, t);
      if (carry || big_cmp<N>(t, c.p) >= 0) big_sub<N>(t, t, c.p);
      if (i == 64 * N - 1) memcpy(c.r, t, sizeof t);
    }
    memcpy(c.r2, t, sizeof t);
    u64 pm1[N];
    memcpy(pm1, c.p, sizeof pm1);
    pm1[0] -= 1;
    big_div_small<N>(c.pm1_half, pm1, 2);
    return c;
  }
  static Fp zero() { Fp r; memset(r.v, 0, sizeof r.v); return r; }
  static Fp one() { Fp r; memcpy(r.v, C().r, sizeof r.v); return r; }
  bool is_zero() const { return big_is_zero<N>(v); }
  bool operator==(const Fp& o) const { return memcmp(v, o.v, sizeof v) == 0; }
  bool operator!=(const Fp& o) const { return !(*this == o); }

  Fp operator+(const Fp& o) const {
    Fp r;
    u64 c = big_add<N>(r.v, v, o.v);
    if (c || big_cmp<N>(r.v, C().p) >= 0) big_sub<N>(r.v, r.v, C().p);
    return r;
  }
  Fp operator-(const Fp& o) const {
    Fp r;
    if (big_sub<N>(r.v, v, o.v)) big_add<N>(r.v, r.v, C().p);
    return r;
  }
  Fp operator-() const {
    if (is_zero()) return *this;
    Fp r;
    big_sub<N>(r.v, C().p, v);
    return r;
  }
  Fp dbl() const { return *this + *this; }
  // Montgomery CIOS multiplication
  Fp operator*(const Fp& o) const {
    const u64* p = C().p;
    const u64 inv = C().inv;
    u64 t[N + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < N; i++) {
      u128 c = 0;
      for (int j = 0; j < N; j++) {
        c += (u128)v[j] * o.v[i] + t[j];
        t[j] = (u64)c;
        c >>= 64;
      }
      c += t[N];
      t[N] = (u64)c;
      t[N + 1] = (u64)(c >> 64);
      u64 m = t[0] * inv;
      c = (u128)m * p[0] + t[0];
      c >>= 64;
      for (int j = 1; j < N; j++) {
        c += (u128)m * p[j] + t[j];
        t[j - 1] = (u64)c;
        c >>= 64;
      }
      c += t[N];
      t[N - 1] = (u64)c;
      t[N] = t[N + 1] + (u64)(c >> 64);
    }
    Fp r;
    if (t[N] || big_cmp<N>(t, p) >= 0) big_sub<N>(r.v, t, p);
    else memcpy(r.v, t, sizeof r.v);
    return r;
  }
  Fp sqr() const { return *this * *this; }
  Fp& operator+=(const Fp& o) { return *this = *this + o; }
  Fp& operator-=(const Fp& o) { return *this = *this - o; }
  Fp& operator*=(const Fp& o) { return *this = *this * o; }

  // canonical <-> Montgomery
  static Fp from_canonical(const u64* a) {
    Fp x, r2;
    memcpy(x.v, a, sizeof x.v);
    memcpy(r2.v, C().r2, sizeof r2.v);
    return x * r2;
  }
  void to_canonical(u64* out) const {
    Fp o;
    memset(o.v, 0, sizeof o.v);
    o.v[0] = 1;
    Fp r = *this * o;
    memcpy(out, r.v, sizeof r.v);
  }
  static Fp from_u64(u64 a) {
    u64 t[N];
    memset(t, 0, sizeof t);
    t[0] = a;
    return from_canonical(t);
  }
  static Fp from_int(long a) { return a >= 0 ? from_u64((u64)a) : -from_u64((u64)(-a)); }
  template <int M> Fp pow(const u64* e) const {
    Fp r = one();
    for (int i = big_bits<M>(e) - 1; i >= 0; i--) {
      r = r.sqr();
      if (big_bit(e, i)) r = r * *this;
    }
    return r;
  }
  Fp inverse() const {  // Fermat
    u64 e[N];
    memcpy(e, C().p, sizeof e);
    e[0] -= 2;  // p odd and > 2, no borrow
    return pow<N>(e);
  }
  // arkworks' "is y lexicographically larger than -y": canonical(y) > (p-1)/2
  bool lex_largest() const {
    u64 c[N];
    to_canonical(c);
    return big_cmp<N>(c, C().pm1_half) > 0;
  }
  // Tonelli-Shanks (handles p = 3 mod 4 as the s = 1 case); returns false if non-residue
  bool sqrt(Fp& out) const {
    if (is_zero()) { out = *this; return true; }
    u64 t[N], e[N];
    memcpy(t, C().p, sizeof t);
    t[0] -= 1;
    int s = 0;
    while (!(t[0] & 1)) { big_div_small<N>(t, t, 2); s++; }
    // legendre
    Fp leg = pow<N>(C().pm1_half);
    if (leg != one()) return false;
    // non-residue z
    Fp z = from_u64(2);
    while (z.template pow<N>(C().pm1_half) == one()) z = z + one();
    Fp c = z.template pow<N>(t);
    // x = a^((t+1)/2), b = a^t
    u64 one_[N];
    memset(one_, 0, sizeof one_);
    one_[0] = 1;
    big_add<N>(e, t, one_);
    big_div_small<N>(e, e, 2);
    Fp x = pow<N>(e);
    Fp b = pow<N>(t);
    int m = s;
    while (b != one()) {
      int i = 0;
      Fp b2 = b;
      while (b2 != one()) { b2 = b2.sqr(); i++; }
      Fp g = c;
      for (int k = 0; k < m - i - 1; k++) g = g.sqr();
      x = x * g;
      c = g.sqr();
      b = b * c;
      m = i;
    }
    out = x;
    return true;
  }
};

struct Tag377 {
  static constexpr int N = 6;
  static const char* hex() {
    return "01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001";
  }
};
struct Tag761 {
  static constexpr int N = 12;
  static const char* hex() {
    return "0122e824fb83ce0ad187c94004faff3eb926186a81d14688528275ef8087be41707ba638e584e91903cebaff25b423048689c8ed12f9fd9071dcd3dc73ebff2e98a116c25667a8f8160cf8aeeaf0a437e6913e6870000082f49d00000000008b";
  }
};
// Fr of BLS12-377 (253 bits, 2-adicity 47): the field of the hash-helper proof's witness map
struct Tag253 {
  static constexpr int N = 4;
  static const char* hex() { return "12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001"; }
};
typedef Fp<Tag377> Fq377;
typedef Fp<Tag761> Fq761;
typedef Fp<Tag253> Fr253;

// ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2 - NR), NR a small int
template <class F, int NR> struct Fp2T {
  typedef F Base;
  F c0, c1;
  static F mul_nr(const F& a) {  // a * NR for small negative/positive NR
    int k = NR < 0 ? -NR : NR;
    F acc = F::zero(), cur = a;
    while (k) { if (k & 1) acc = acc + cur; cur = cur.dbl(); k >>= 1; }
    return NR < 0 ? -acc : acc;
  }
  static Fp2T zero() { return {F::zero(), F::zero()}; }
  static Fp2T one() { return {F::one(), F::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fp2T& o) const { return c0 == o.c0 && c1 == o.c1; }
  bool operator!=(const Fp2T& o) const { return !(*this == o); }
  Fp2T operator+(const Fp2T& o) const { return {c0 + o.c0, c1 + o.c1}; }
  Fp2T operator-(const Fp2T& o) const { return {c0 - o.c0, c1 - o.c1}; }
  Fp2T operator-() const { return {-c0, -c1}; }
  Fp2T dbl() const { return {c0.dbl(), c1.dbl()}; }
  Fp2T operator*(const Fp2T& o) const {  // Karatsuba
    F v0 = c0 * o.c0, v1 = c1 * o.c1;
    F s = (c0 + c1) * (o.c0 + o.c1);
    return {v0 + mul_nr(v1), s - v0 - v1};
  }
  Fp2T sqr() const {
    F ab = c0 * c1;
    F t = (c0 + c1) * (c0 + mul_nr(c1));
    return {t - ab - mul_nr(ab), ab.dbl()};
  }
  Fp2T mul_fp(const F& k) const { return {c0 * k, c1 * k}; }
  Fp2T conj() const { return {c0, -c1}; }
  Fp2T inverse() const {
    F n = c0.sqr() - mul_nr(c1.sqr());
    F ni = n.inverse();
    return {c0 * ni, -(c1 * ni)};
  }
  Fp2T frob(int i) const { return (i & 1) ? conj() : *this; }
  Fp2T& operator+=(const Fp2T& o) { return *this = *this + o; }
  Fp2T& operator-=(const Fp2T& o) { return *this = *this - o; }
  Fp2T& operator*=(const Fp2T& o) { return *this = *this * o; }
  template <int M> Fp2T pow(const u64* e) const {
    Fp2T r = one();
    for (int i = big_bits<M>(e) - 1; i >= 0; i--) {
      r = r.sqr();
      if (big_bit(e, i)) r = r * *this;
    }
    return r;
  }
  // arkworks ordering for the sign flag: compare c1 first, then c0
  // (reference: crates/epoch-snark/src/encoding.rs:32-33 mirrors it)
  bool lex_largest() const {
    if (!c1.is_zero()) return c1.lex_largest();
    return c0.lex_largest();
  }
  bool sqrt(Fp2T& out) const {
    if (is_zero()) { out = *this; return true; }
    F nrF = F::from_int(NR);
    if (c1.is_zero()) {
      F s;
      if (c0.sqrt(s)) { out = {s, F::zero()}; return true; }
      F t = c0 * nrF.inverse();
      if (!t.sqrt(s)) return false;
      out = {F::zero(), s};
      return true;
    }
    F n = c0.sqr() - mul_nr(c1.sqr());
    F al;
    if (!n.sqrt(al)) return false;
    F i2 = F::from_u64(2).inverse();
    F d = (c0 + al) * i2, x0;
    if (!d.sqrt(x0)) {
      d = (c0 - al) * i2;
      if (!d.sqrt(x0)) return false;
    }
    F x1 = c1 * x0.dbl().inverse();
    out = {x0, x1};
    return out.sqr() == *this;
  }
};
typedef Fp2T<Fq377, -5> Fq2_377;

// ---------------------------------------------------------------- BLS12-377 Fq6 = Fq2[v]/(v^3 - u)
struct Fq6_377 {
  typedef Fq2_377 F2;
  F2 c0, c1, c2;
  static F2 mul_xi(const F2& a) {  // (a0 + a1 u) * u = NR*a1 + a0 u
    return {F2::mul_nr(a.c1), a.c0};
  }
  static Fq6_377 zero() { return {F2::zero(), F2::zero(), F2::zero()}; }
  static Fq6_377 one() { return {F2::one(), F2::zero(), F2::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
  bool operator==(const Fq6_377& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
  Fq6_377 operator+(const Fq6_377& o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
  Fq6_377 operator-(const Fq6_377& o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
  Fq6_377 operator-() const { return {-c0, -c1, -c2}; }
  Fq6_377 operator*(const Fq6_377& o) const {
    F2 v0 = c0 * o.c0, v1 = c1 * o.c1, v2 = c2 * o.c2;
    F2 t0 = mul_xi((c1 + c2) * (o.c1 + o.c2) - v1 - v2) + v0;
    F2 t1 = (c0 + c1) * (o.c0 + o.c1) - v0 - v1 + mul_xi(v2);
    F2 t2 = (c0 + c2) * (o.c0 + o.c2) - v0 - v2 + v1;
    return {t0, t1, t2};
  }
  Fq6_377 sqr() const { return *this * *this; }
  Fq6_377 mul_by_v() const { return {mul_xi(c2), c0, c1}; }  // * v
  Fq6_377 inverse() const {
    F2 t0 = c0.sqr() - mul_xi(c1 * c2);
    F2 t1 = mul_xi(c2.sqr()) - c0 * c1;
    F2 t2 = c1.sqr() - c0 * c2;
    F2 d = (c0 * t0 + mul_xi(c2 * t1) + mul_xi(c1 * t2)).inverse();
    return {t0 * d, t1 * d, t2 * d};
  }
};

// Frobenius coefficients for the BLS12-377 tower, computed at start-up:
//   g[i] = xi^((q^i - 1)/6), i = 0..11  (xi = u);  Fq6 uses g[i]^2 and g[i]^4.
struct Frob377 {
  Fq2_377 g[12];
  static Frob377& get() {
    static Frob377 f = make();
    return f;
  }
  static Frob377 make() {
    Frob377 f;
    u64 e[6], pm1[6];
    memcpy(pm1, Fq377::C().p, sizeof pm1);
    pm1[0] -= 1;
    u64 rem = big_div_small<6>(e, pm1, 6);
    if (rem) { fprintf(stderr, "q-1 not divisible by 6\n"); abort(); }
    Fq2_377 xi = {Fq377::zero(), Fq377::one()};
    f.g[0] = Fq2_377::one();
    f.g[1] = xi.pow<6>(e);
    for (int i = 2; i < 12; i++) f.g[i] = f.g[1] * f.g[i - 1].conj();  // g[i] = g1 * g[i-1]^q
    return f;
  }
};

struct Fq12_377 {
  typedef Fq2_377 F2;
  typedef Fq6_377 F6;
  F6 c0, c1;
  static Fq12_377 one() { return {F6::one(), F6::zero()}; }
  bool operator==(const Fq12_377& o) const { return c0 == o.c0 && c1 == o.c1; }
  bool is_one() const { return *this == one(); }
  Fq12_377 operator*(const Fq12_377& o) const {
    F6 v0 = c0 * o.c0, v1 = c1 * o.c1;
    F6 t = (c0 + c1) * (o.c0 + o.c1) - v0 - v1;
    return {v0 + v1.mul_by_v(), t};
  }
  Fq12_377 sqr() const {
    F6 ab = c0 * c1;
    F6 t = (c0 + c1) * (c0 + c1.mul_by_v()) - ab - ab.mul_by_v();
    return {t, ab + ab};
  }
  Fq12_377 conj() const { return {c0, -c1}; }
  Fq12_377 inverse() const {
    F6 d = (c0.sqr() - c1.sqr().mul_by_v()).inverse();
    return {c0 * d, -(c1 * d)};
  }
  Fq12_377 frob(int i) const {
    const Frob377& fr = Frob377::get();
    F2 g1 = fr.g[i % 12];
    F2 g2 = g1.sqr(), g3 = g2 * g1, g4 = g2.sqr(), g5 = g4 * g1;
    Fq12_377 r;
    r.c0.c0 = c0.c0.frob(i);
    r.c0.c1 = c0.c1.frob(i) * g2;
    r.c0.c2 = c0.c2.frob(i) * g4;
    r.c1.c0 = c1.c0.frob(i) * g1;
    r.c1.c1 = c1.c1.frob(i) * g3;
    r.c1.c2 = c1.c2.frob(i) * g5;
    return r;
  }
  // sparse multiplication by c0 + (d0 + d1 v) w   (ark-ff Fp12::mul_by_034)
  Fq12_377 mul_by_034(const F2& s0, const F2& s3, const F2& s4) const {
    F6 a = {c0.c0 * s0, c0.c1 * s0, c0.c2 * s0};
    // b = c1 * (s3 + s4 v)
    F6 b = mul6_by_01(c1, s3, s4);
    F2 d0 = s0 + s3;
    F6 e = mul6_by_01(c0 + c1, d0, s4);
    return {b.mul_by_v() + a, e - (a + b)};
  }
  static F6 mul6_by_01(const F6& x, const F2& b0, const F2& b1) {
    // (x0 + x1 v + x2 v^2)(b0 + b1 v)
    F2 t0 = x.c0 * b0 + F6::mul_xi(x.c2 * b1);
    F2 t1 = x.c0 * b1 + x.c1 * b0;
    F2 t2 = x.c1 * b1 + x.c2 * b0;
    return {t0, t1, t2};
  }
  // Granger-Scott squaring, valid in the cyclotomic subgroup (ark-ff Fp12::cyclotomic_square)
  Fq12_377 cyclotomic_square() const {
    auto fp4sq = [](const F2& a, const F2& b, F2& o0, F2& o1) {
      F2 tmp = a * b;
      o0 = (a + b) * (F6::mul_xi(b) + a) - tmp - F6::mul_xi(tmp);
      o1 = tmp.dbl();
    };
    const F2 &r0 = c0.c0, &r4 = c0.c1, &r3 = c0.c2, &r2 = c1.c0, &r1 = c1.c1, &r5 = c1.c2;
    F2 t0, t1, t2, t3, t4, t5;
    fp4sq(r0, r1, t0, t1);
    fp4sq(r2, r3, t2, t3);
    fp4sq(r4, r5, t4, t5);
    Fq12_377 z;
    z.c0.c0 = (t0 - r0).dbl() + t0;
    z.c1.c1 = (t1 + r1).dbl() + t1;
    F2 tmp = F6::mul_xi(t5);
    z.c1.c0 = (tmp + r2).dbl() + tmp;
    z.c0.c2 = (t4 - r3).dbl() + t4;
    z.c0.c1 = (t2 - r4).dbl() + t2;
    z.c1.c2 = (t3 + r5).dbl() + t3;
    return z;
  }
};

// ---------------------------------------------------------------- BW6-761 Fq3 = Fq[u]/(u^3 + 4), Fq6 = Fq3[v]/(v^2 - u)
struct Fq3_761 {
  typedef Fq761 F;
  F c0, c1, c2;
  static F mul_nr(const F& a) { return -(a.dbl().dbl()); }  // * (-4)
  static Fq3_761 zero() { return {F::zero(), F::zero(), F::zero()}; }
  static Fq3_761 one() { return {F::one(), F::zero(), F::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
  bool operator==(const Fq3_761& o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
  Fq3_761 operator+(const Fq3_761& o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
  Fq3_761 operator-(const Fq3_761& o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
  Fq3_761 operator-() const { return {-c0, -c1, -c2}; }
  Fq3_761 operator*(const Fq3_761& o) const {
    F v0 = c0 * o.c0, v1 = c1 * o.c1, v2 = c2 * o.c2;
    F t0 = mul_nr((c1 + c2) * (o.c1 + o.c2) - v1 - v2) + v0;
    F t1 = (c0 + c1) * (o.c0 + o.c1) - v0 - v1 + mul_nr(v2);
    F t2 = (c0 + c2) * (o.c0 + o.c2) - v0 - v2 + v1;
    return {t0, t1, t2};
  }
  Fq3_761 sqr() const { return *this * *this; }
  Fq3_761 mul_by_u() const { return {mul_nr(c2), c0, c1}; }
  Fq3_761 mul_fp(const F& k) const { return {c0 * k, c1 * k, c2 * k}; }
  Fq3_761 inverse() const {
    F t0 = c0.sqr() - mul_nr(c1 * c2);
    F t1 = mul_nr(c2.sqr()) - c0 * c1;
    F t2 = c1.sqr() - c0 * c2;
    F d = (c0 * t0 + mul_nr(c2 * t1) + mul_nr(c1 * t2)).inverse();
    return {t0 * d, t1 * d, t2 * d};
  }
};
// Frobenius constants for BW6-761: h[i] = (-4)^((q^i-1)/6), i = 0..5 (all in Fq since 6 | q-1)
struct Frob761 {
  Fq761 h[6];
  static Frob761& get() {
    static Frob761 f = make();
    return f;
  }
  static Frob761 make() {
    Frob761 f;
    u64 e[12], pm1[12];
    memcpy(pm1, Fq761::C().p, sizeof pm1);
    pm1[0] -= 1;
    u64 rem = big_div_small<12>(e, pm1, 6);
    if (rem) { fprintf(stderr, "q761-1 not divisible by 6\n"); abort(); }
    Fq761 nr = Fq761::from_int(-4);
    f.h[0] = Fq761::one();
    f.h[1] = nr.pow<12>(e);
    for (int i = 2; i < 6; i++) f.h[i] = f.h[i - 1] * f.h[1];  // h1 in Fq: h1^q = h1
    return f;
  }
};
struct Fq6_761 {
  typedef Fq761 F;
  typedef Fq3_761 F3;
  F3 c0, c1;
  static Fq6_761 one() { return {F3::one(), F3::zero()}; }
  bool operator==(const Fq6_761& o) const { return c0 == o.c0 && c1 == o.c1; }
  Fq6_761 operator*(const Fq6_761& o) const {
    F3 v0 = c0 * o.c0, v1 = c1 * o.c1;
    F3 t = (c0 + c1) * (o.c0 + o.c1) - v0 - v1;
    return {v0 + v1.mul_by_u(), t};
  }
  Fq6_761 sqr() const { return *this * *this; }
  Fq6_761 conj() const { return {c0, -c1}; }
  Fq6_761 inverse() const {
    F3 d = (c0.sqr() - c1.sqr().mul_by_u()).inverse();
    return {c0 * d, -(c1 * d)};
  }
  // x^(q^i): basis element u^a v^b = w^(2a+b) (w = v, w^6 = -4) maps to itself times h[i]^(2a+b)
  Fq6_761 frob(int i) const {
    const Frob761& fr = Frob761::get();
    F h1 = fr.h[i % 6], h2 = h1 * h1, h3 = h2 * h1, h4 = h2 * h2, h5 = h4 * h1;
    return {{c0.c0, c0.c1 * h2, c0.c2 * h4}, {c1.c0 * h1, c1.c1 * h3, c1.c2 * h5}};
  }
  // sparse: multiply by (s0 + s1 u) + (s4 u) v     (ark-ff Fp6_2over3::mul_by_014)
  Fq6_761 mul_by_014(const F& s0, const F& s1, const F& s4) const {
    Fq6_761 o = {{s0, s1, F::zero()}, {F::zero(), s4, F::zero()}};
    return *this * o;
  }
};

}  // namespace orc
