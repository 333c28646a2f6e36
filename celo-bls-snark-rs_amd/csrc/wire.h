// Compressed-point decoding for BLS12-377 G1 / G2 (arkworks 0.1 CanonicalDeserialize: x little-endian, flags in the two top
// bits of the last byte) as host+device templates: the square root (Fq: a discrete log in the 2-Sylow subgroup, 2-adicity
// 46; the norm method over Fq2 = Fq[u]/(u^2+5)), the choice of y by the "lexicographically largest" flag, and the
// prime-order subgroup check r*P = O that GroupAffine::deserialize performs.  SURVEY.md section 8f row f2: what the
// reference does once per key in PublicKey::deserialize / Signature::deserialize (crates/bls-crypto/src/bls/public.rs:123-149,
// signature.rs:31-57) and per validator in EpochBlock::from FFI bytes (crates/bls-snark-sys/src/snark/epoch_block.rs:187-196).
//
// One source for both sides: seam_a.hip calls these functions on the host for single keys (a 1 ms latency path), the
// k_decompress kernels (unit_wire.hip) run them one point per lane for bulk wire data.  A square root is unique up to
// sign and the sign is fixed by the flag, so every correct implementation decodes to the same canonical coordinates.
#pragma once
#include <cstdint>
#include "curve.h"
#include "fp2.h"
#include "lanes.h"

namespace celo {

#if defined(__HIPCC__)
#define WIRE_FN __host__ __device__ inline __attribute__((noinline))
#else
#define WIRE_FN inline
#endif

struct WireTables {         // for the discrete logarithm in the 2^46-th roots of unity (wire_fq_sqrt); 36 KB, host copy + device copy
  Fq t8[256];               // reduce(h^j), h = z^(2^38): the 256-th roots of unity by exponent (the 64-th ones are every 4th entry)
  Fq bt[12][16];            // z^(-j 2^(4m)): removes nibble m of the logarithm from b
  Fq ht[12][16];            // z^(-j 2^(4m) / 2): the matching correction of the root (m = 0: even j only)
  uint16_t slot[1024];      // open-addressed index of t8 by the low limb of the reduced value (linear probing, 0xffff = empty)
};
struct WireConsts {         // built once on the host (wire_consts()), handed to the kernels by value
  uint64_t tm1_half[6];     // (t - 1) / 2 where q - 1 = 2^46 t, t odd
  uint64_t r_order[4];      // r, the prime subgroup order
  Fq z;                     // c^t for the smallest quadratic non-residue c: a generator of the 2^46-th roots of unity
  Fq inv2, inv5;
  Fq m5_x, m5_b;            // (-5)^((t+1)/2), (-5)^t: the root parts of the fixed non-residue -5 (wire_fq2_sqrt)
  Fq beta;                  // the cube root of unity with (beta x, y) = -[x^2](x, y) on G1 (wire_in_subgroup)
  Fq psi_x, psi_y;          // (-5)^((q-1)/6), (-5)^((q-1)/4): the twisted Frobenius of G2, psi(x, y) = (psi_x conj(x), psi_y conj(y))
  const WireTables* tab;    // in the memory space of whoever runs the functions below (unit_wire.hip: wire_consts_device())
};

enum WireStatus : uint8_t { WIRE_OK = 0, WIRE_INFINITY = 1, WIRE_INVALID = 2, WIRE_NOT_IN_SUBGROUP = 3 };

HD bool wire_eq(const Fq& a, const Fq& b) { return Fq::eq_mod_p(Fq::norm(a), Fq::norm(b)); }
HD bool wire_is_one(const Fq& a) { return wire_eq(a, Fq::one()); }
HD Fq wire_neg(const Fq& a) { return Fq::wred(Fq::norm(Fq::neg<64, 1>(Fq::norm(a)))); }   // weak-reduced: vb <= 3 like a decoded coordinate
HD int wire_cmp(const uint64_t* a, const uint64_t* b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  }
  return 0;
}
// canonical little-endian bytes -> field element; false when the integer is not below q (Fp::read's from_repr failure)
HD bool wire_fq_from_bytes(const uint8_t* in, Fq& out) {
  uint64_t w[6];
  for (int i = 0; i < 6; i++) {
    uint64_t v = 0;
    for (int b = 7; b >= 0; b--) v = (v << 8) | in[8 * i + b];
    w[i] = v;
  }
  if (wire_cmp(w, P377::P64, 6) >= 0) return false;
  out = Fq::from_canonical(w);
  return true;
}
HD bool wire_lex_largest(const Fq& a) {   // canonical(a) > (q - 1) / 2
  uint64_t w[6];
  a.to_canonical(w);
  return wire_cmp(w, P377::PM1_HALF64, 6) > 0;
}
HD bool wire_lex_largest(const Fq2& y) {  // arkworks orders Fq2 by c1 first, then c0
  if (!y.c1.is_zero_mod_p()) return wire_lex_largest(y.c1);
  return wire_lex_largest(y.c0);
}

// a^((t-1)/2), the exponentiation of the square root (q - 1 = 2^46 t).  The exponent is a constant, so its sliding-window program
// (fp_consts.h SqrtChain377: 328 squarings, 83 products by a, a^3, a^5 or a^7) is the same in every lane: the four table entries
// stay in registers and every branch is scalar.  (Round 2 kept a 16-entry table of fixed 4-bit windows in private memory, indexed
// at run time: 896 of the 2812 B/lane of scratch of k_decompress<true>, re-read 83 times per root.)
WIRE_FN Fq wire_pow_root_exponent(const Fq& a_) {
  const Fq t1 = Fq::norm(a_), a2 = Fq::sqr(t1);
  const Fq t3 = Fq::mul(t1, a2), t5 = Fq::mul(t3, a2), t7 = Fq::mul(t5, a2);
  static_assert(SqrtChain377::FIRST == 1 || SqrtChain377::FIRST == 3 || SqrtChain377::FIRST == 5 || SqrtChain377::FIRST == 7, "odd leading window");
  Fq r = SqrtChain377::FIRST == 1 ? t1 : SqrtChain377::FIRST == 3 ? t3 : SqrtChain377::FIRST == 5 ? t5 : t7;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 0; i < SqrtChain377::LEN; i++) {
    r = Fq::sqr(r);
    const int d = SqrtChain377::STEP[i];
    if (d == 1) r = Fq::mul(r, t1);
    else if (d == 3) r = Fq::mul(r, t3);
    else if (d == 5) r = Fq::mul(r, t5);
    else if (d == 7) r = Fq::mul(r, t7);
  }
  return r;
}

// Square root in Fq (q - 1 = 2^46 t).  One exponentiation w = a^((t-1)/2) gives x = a w = a^((t+1)/2) and b = x w = a^t, a
// 2^46-th root of unity: b = z^e, and a is a square exactly when e is even, with root x z^(-e/2).  e is found digit by
// digit (Pohlig-Hellman, five 8-bit digits and one of 6 bits, least significant first): b^(2^(46 - s - w)) is a 2^w-th root
// of unity whose exponent is the next digit - read off a table of the 256-th roots - and the digit is then divided out of b
// while x collects the half power.  110 squarings and <= 24 products after the exponentiation, the same for every input:
// the classic Tonelli-Shanks order search it replaces takes ~550 squarings on average and ~1000 for the slowest lane of a
// wave.  An odd first digit means a non-residue.
HD Fq wire_inv(const Fq& a) { return Fq::inv(a); }                 // division steps (modinv.h), not a^(q - 2)

struct WireRootParts { Fq x, b; };                                  // x = a^((t+1)/2), b = a^t for the a last tried
// the digit-by-digit part: x^2 = a b with b = z^e a 2^46-th root of unity -> x z^(-e/2) (false: e odd, a non-residue)
WIRE_FN bool wire_root_finish(Fq x, Fq b, const WireConsts& k, Fq& out) {
  const WireTables& T = *k.tab;
  for (int i = 0; i < 6; i++) {
    const int s = 8 * i, wd = i == 5 ? 6 : 8;
    Fq u = b;
    for (int q = 0; q < 46 - s - wd; q++) u = Fq::sqr(u);
    u = Fq::reduce(u);
    int e = -1;
    uint32_t sl = u.l[0] & 1023u;                                   // two dependent loads per digit instead of a scan of the table
    for (int probe = 0; probe < 1024; probe++) {
      const uint32_t j = T.slot[sl];
      if (j == 0xffffu) break;
      const Fq& c = T.t8[j];
      bool same = true;
      for (int q = 0; q < P377::L; q++) same = same && c.l[q] == u.l[q];
      if (same) { e = (int)j; break; }
      sl = (sl + 1) & 1023u;
    }
    if (i == 5 && e >= 0) e = (e & 3) ? -1 : e >> 2;               // the last digit has 6 bits: a 64-th root of unity
    if (e < 0) return false;                 // cannot happen: b is a 2^46-th root of unity
    if (i == 0 && (e & 1)) return false;     // odd logarithm: a is a non-residue
    const int lo = e & 15, hi = e >> 4;
    if (lo) { b = Fq::mul(b, T.bt[2 * i][lo]); x = Fq::mul(x, T.ht[2 * i][lo]); }
    if (hi) { b = Fq::mul(b, T.bt[2 * i + 1][hi]); x = Fq::mul(x, T.ht[2 * i + 1][hi]); }
  }
  out = x;
  return true;
}
WIRE_FN bool wire_fq_sqrt_keep(const Fq& a_, const WireConsts& k, Fq& out, WireRootParts& parts) {
  const Fq a = Fq::norm(a_);
  if (a.is_zero_mod_p()) { out = Fq::zero(); parts.x = Fq::zero(); parts.b = Fq::one(); return true; }
  const Fq w = wire_pow_root_exponent(a);
  parts.x = Fq::mul(a, w);
  parts.b = Fq::mul(parts.x, w);
  return wire_root_finish(parts.x, parts.b, k, out);
}
HD bool wire_fq_sqrt(const Fq& a, const WireConsts& k, Fq& out) {
  WireRootParts parts;
  return wire_fq_sqrt_keep(a, k, out, parts);
}
// The textbook Tonelli-Shanks loop (order search): kept as the cross-check of the table-driven root (tests/test_host_field.py)
WIRE_FN bool wire_fq_sqrt_ts(const Fq& a_, const WireConsts& k, Fq& out) {
  const Fq a = Fq::norm(a_);
  if (a.is_zero_mod_p()) { out = Fq::zero(); return true; }
  const Fq w = Fq::pow64(a, k.tm1_half, 6);
  Fq x = Fq::mul(a, w);
  Fq b = Fq::mul(x, w);
  Fq zz = k.z;
  int m = 46;
  while (!wire_is_one(b)) {
    int i = 0;
    Fq b2 = b;
    while (!wire_is_one(b2)) {
      if (i == m - 1) return false;
      b2 = Fq::sqr(b2);
      i++;
    }
    Fq g = zz;
    for (int j = 0; j < m - i - 1; j++) g = Fq::sqr(g);
    x = Fq::mul(x, g);
    zz = Fq::sqr(g);
    b = Fq::mul(b, zz);
    m = i;
  }
  out = x;
  return true;
}
// sqrt in Fq2 by the norm: for a = a0 + a1 u, alpha = sqrt(a0^2 + 5 a1^2), x0^2 = (a0 +- alpha) / 2, x1 = a1 / (2 x0)
WIRE_FN bool wire_fq2_sqrt(const Fq2& a, const WireConsts& k, Fq2& out) {
  if (a.is_zero_mod_p()) { out = Fq2::zero(); return true; }
  const Fq a0 = Fq::norm(a.c0), a1 = Fq::norm(a.c1);
  Fq s;
  if (a1.is_zero_mod_p()) {
    if (wire_fq_sqrt(a0, k, s)) { out = {s, Fq::zero()}; return true; }
    const Fq tt = wire_neg(Fq::mul(a0, k.inv5));      // a0 = -5 t^2  ->  sqrt = t u
    if (!wire_fq_sqrt(tt, k, s)) return false;
    out = {Fq::zero(), s};
    return true;
  }
  const Fq s1 = Fq::sqr(a1);
  const Fq n = Fq::norm(Fq::add(Fq::sqr(a0), Fq::norm(Fq::add(Fq::dbl(Fq::dbl(s1)), s1))));
  Fq al, x0;
  if (!wire_fq_sqrt(n, k, al)) return false;
  const Fq d = Fq::norm(Fq::mul(Fq::norm(Fq::add(a0, al)), k.inv2));
  WireRootParts pd;
  Fq x1;
  if (wire_fq_sqrt_keep(d, k, x0, pd)) x1 = Fq::mul(a1, Fq::inv(Fq::norm(Fq::dbl(x0))));
  else {
    // d is a non-residue and so is -5, so -5 d is a square whose (t+1)/2-th and t-th powers are products of what the failed
    // attempt already computed and two constants: only the digit part is run again.  The other candidate is
    // d' = (a0 - alpha) / 2 = -5 a1^2 / (4 d) = (-5 d) (a1 / (2 d))^2, so with s = sqrt(-5 d): x0 = s a1 / (2 d), x1 = a1 / (2 x0) = d / s.
    Fq sr;
    if (!wire_root_finish(Fq::mul(k.m5_x, pd.x), Fq::mul(k.m5_b, pd.b), k, sr)) return false;
    const Fq isd = Fq::norm(Fq::inv(Fq::norm(Fq::mul(sr, d))));      // (s d)^-1: one inversion for 1/s and 1/d
    const Fq si = Fq::mul(isd, d), di = Fq::mul(isd, sr);
    x1 = Fq::mul(d, si);
    x0 = Fq::mul(Fq::mul(sr, a1), Fq::mul(di, k.inv2));
  }
  out = {x0, x1};
  const Fq2 chk = Fq2::sqr(out);
  return wire_eq(chk.c0, a0) && wire_eq(chk.c1, a1);
}

// r * P == O, MSB-first double-and-add over the 253 bits of r: the reference's is_in_correct_subgroup_assuming_on_curve
// (ark-ec 0.1 GroupAffine) as written.  Kept as the definition the two tests below are checked against
// (tests/test_host_field.py, tests/test_wire_gpu.py); the decoders use the endomorphism forms.
template <class F> WIRE_FN bool wire_in_subgroup_ladder(const Affine<F>& p, const WireConsts& k) {
  Xyzz<F> acc = Xyzz<F>::from_affine(p);
  for (int i = 251; i >= 0; i--) {
    acc = xyzz_dbl(acc);      // inlined: the out-of-line variant saves / restores ~100 callee-saved VGPRs per call on gfx950 (252 calls per point)
    if ((k.r_order[i >> 6] >> (i & 63)) & 1) xyzz_madd(acc, p);
  }
  return acc.is_identity() || acc.ZZ.is_zero_mod_p();
}

// The same predicate through the curves' efficient endomorphisms (x = 0x8508c00000000001 is the BLS12 parameter,
// r = x^4 - x^2 + 1, q + 1 - t = h1 r with t = x + 1): a 64-bit ladder (or two) instead of a 253-bit one.
//
// G1.  phi(x, y) = (beta x, y) satisfies phi^2 + phi + 1 = 0 on all of E(Fq).  If phi(P) = -[x^2]P then
//      0 = (phi^2 + phi + 1)P = [x^4 - x^2 + 1]P = [r]P; conversely phi acts on the order-r points as one of the two roots
//      of l^2 + l + 1 mod r, which are -x^2 and x^2 - 1, and beta is the cube root of unity that gives -x^2.  So
//      "phi(P) == -[x^2]P" holds exactly when r P = O: no cofactor condition is involved.
// G2.  psi = twist^-1 o Frobenius o twist satisfies psi^2 - t psi + q = 0 on all of E'(Fq2).  If psi(P) = [x]P then
//      0 = [x^2 - (x + 1) x + q]P = [q - x]P = [h1 r]P, and P also has order dividing #E'(Fq2) = h2 r; gcd(h1, h2) = 1 for
//      BLS12-377 (checked numerically in tests/test_oracle_golden.py), so r P = O.  Conversely psi acts on G2 as
//      multiplication by q = t - 1 = x mod r.  So "psi(P) == [x]P" holds exactly when r P = O.
// (The tests of Scott, eprint 2021/1130, with the cofactor condition of El Housni-Guillevic-Piellard, eprint 2022/352;
// later arkworks and gnark releases use them for this curve.)  Identical verdicts, 2.5x / 5x fewer field products.
static constexpr uint64_t WIRE_X = 0x8508c00000000001ULL;
template <class F> HD Xyzz<F> wire_mul_x(const Affine<F>& p) {     // [x]P, p finite
  Xyzz<F> acc = Xyzz<F>::from_affine(p);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 62; i >= 0; i--) {
    acc = xyzz_dbl(acc);
    if ((WIRE_X >> i) & 1) xyzz_madd(acc, p);
  }
  return acc;
}
WIRE_FN bool wire_in_subgroup(const Affine<Fq>& p, const WireConsts& k) {
  const Xyzz<Fq> a = wire_mul_x(p);                                  // [x]P
  Xyzz<Fq> acc = a;                                                  // [x]([x]P): the same ladder with full additions
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 62; i >= 0; i--) {
    acc = xyzz_dbl(acc);
    if ((WIRE_X >> i) & 1) xyzz_add(acc, a);
  }
  if (acc.is_identity() || acc.ZZ.is_zero_mod_p()) return false;     // phi(P) is a finite point
  // -[x^2]P == (beta x, y):  X == beta x ZZ  and  -Y == y ZZZ
  return wire_eq(acc.X, Fq::mul(Fq::mul(k.beta, p.x), acc.ZZ)) && wire_eq(wire_neg(acc.Y), Fq::mul(p.y, acc.ZZZ));
}
WIRE_FN bool wire_in_subgroup(const Affine<Fq2>& p, const WireConsts& k) {
  const Xyzz<Fq2> a = wire_mul_x(p);                                 // [x]P
  if (a.is_identity() || a.ZZ.is_zero_mod_p()) return false;
  const Fq2 px = {Fq::mul(p.x.c0, k.psi_x), Fq::mul(wire_neg(p.x.c1), k.psi_x)};
  const Fq2 py = {Fq::mul(p.y.c0, k.psi_y), Fq::mul(wire_neg(p.y.c1), k.psi_y)};
  const Fq2 lx = Fq2::mul(Fq2::norm(px), a.ZZ), ly = Fq2::mul(Fq2::norm(py), a.ZZZ);
  return wire_eq(a.X.c0, lx.c0) && wire_eq(a.X.c1, lx.c1) && wire_eq(a.Y.c0, ly.c0) && wire_eq(a.Y.c1, ly.c1);
}

// 48 bytes -> affine G1 point (y^2 = x^3 + 1)
HD WireStatus wire_decode_g1(const uint8_t* in, const WireConsts& k, bool check_subgroup, Affine<Fq>& p) {
  uint8_t buf[48];
  for (int i = 0; i < 48; i++) buf[i] = in[i];
  const uint8_t flags = buf[47] & 0xC0;
  buf[47] &= 0x3F;
  if (flags == 0xC0) return WIRE_INVALID;       // ark-serialize SWFlags::from_u8: (sign, infinity) both set is no encoding at all
  if (flags & 0x40) return WIRE_INFINITY;
  Fq x, y;
  if (!wire_fq_from_bytes(buf, x)) return WIRE_INVALID;
  const Fq rhs = Fq::norm(Fq::add(Fq::mul(Fq::sqr(x), x), Fq::one()));
  if (!wire_fq_sqrt(rhs, k, y)) return WIRE_INVALID;
  if (wire_lex_largest(y) != ((flags & 0x80) != 0)) y = wire_neg(y);
  p = {Fq::norm(x), Fq::norm(y)};
  if (check_subgroup && !wire_in_subgroup(p, k)) return WIRE_NOT_IN_SUBGROUP;
  return WIRE_OK;
}
// 96 bytes (x = c0 || c1, flags on c1's last byte) -> affine G2 point (y^2 = x^3 + B', B' = 1/u = -u/5)
HD WireStatus wire_decode_g2(const uint8_t* in, const WireConsts& k, bool check_subgroup, Affine<Fq2>& p) {
  uint8_t buf[96];
  for (int i = 0; i < 96; i++) buf[i] = in[i];
  const uint8_t flags = buf[95] & 0xC0;
  buf[95] &= 0x3F;
  if (flags == 0xC0) return WIRE_INVALID;
  if (flags & 0x40) return WIRE_INFINITY;
  Fq2 x, y;
  if (!wire_fq_from_bytes(buf, x.c0) || !wire_fq_from_bytes(buf + 48, x.c1)) return WIRE_INVALID;
  const Fq2 tb = {Fq::zero(), wire_neg(k.inv5)};
  const Fq2 rhs = Fq2::norm(Fq2::add(Fq2::mul(Fq2::sqr(x), x), tb));
  if (!wire_fq2_sqrt(rhs, k, y)) return WIRE_INVALID;
  if (wire_lex_largest(y) != ((flags & 0x80) != 0)) y = {wire_neg(y.c0), wire_neg(y.c1)};
  p = {Fq2::norm(x), Fq2::norm(y)};
  if (check_subgroup && !wire_in_subgroup(p, k)) return WIRE_NOT_IN_SUBGROUP;
  return WIRE_OK;
}

// ---- host: the constants (one search for the smallest non-residue, three exponentiations; first use only)
inline WireConsts wire_consts_build() {
  WireConsts k;
  uint64_t pm1[6], t[6];
  for (int i = 0; i < 6; i++) pm1[i] = P377::P64[i];
  pm1[0] -= 1;
  for (int i = 0; i < 6; i++) t[i] = (pm1[i] >> 46) | (i + 1 < 6 ? pm1[i + 1] << 18 : 0);
  uint64_t tm1[6];
  for (int i = 0; i < 6; i++) tm1[i] = t[i];
  tm1[0] -= 1;   // t is odd
  for (int i = 0; i < 6; i++) k.tm1_half[i] = (tm1[i] >> 1) | (i + 1 < 6 ? tm1[i + 1] << 63 : 0);
  const uint64_t r[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};
  for (int i = 0; i < 4; i++) k.r_order[i] = r[i];
  for (uint64_t g = 2;; g++) {
    const uint64_t gw[6] = {g, 0, 0, 0, 0, 0};
    const Fq gf = Fq::from_canonical(gw);
    if (!wire_is_one(Fq::pow64(gf, P377::PM1_HALF64, 6))) { k.z = Fq::pow64(gf, t, 6); break; }
  }
  const uint64_t two[6] = {2, 0, 0, 0, 0, 0}, five[6] = {5, 0, 0, 0, 0, 0};
  k.inv2 = Fq::inv(Fq::from_canonical(two));
  k.inv5 = Fq::inv(Fq::from_canonical(five));
  {
    const Fq m5 = wire_neg(Fq::from_canonical(five));
    const Fq w5 = Fq::pow64(m5, k.tm1_half, 6);
    const Fq x5 = Fq::mul(m5, w5);
    k.m5_x = Fq::wred(Fq::norm(x5));
    k.m5_b = Fq::wred(Fq::norm(Fq::mul(x5, w5)));
  }
  {  // the endomorphism constants of wire_in_subgroup: (-5)^((q-1)/6), (-5)^((q-1)/4) and beta = psi_x^4
    auto div_small = [&](uint64_t d, uint64_t* o) {
      unsigned __int128 rem = 0;
      for (int i = 5; i >= 0; i--) {
        const unsigned __int128 cur = (rem << 64) | pm1[i];
        o[i] = (uint64_t)(cur / d);
        rem = cur % d;
      }
    };
    uint64_t e6[6], e4[6];
    div_small(6, e6);
    div_small(4, e4);
    const Fq m5 = wire_neg(Fq::from_canonical(five));
    k.psi_x = Fq::wred(Fq::norm(Fq::pow64(m5, e6, 6)));
    k.psi_y = Fq::wred(Fq::norm(Fq::pow64(m5, e4, 6)));
    k.beta = Fq::wred(Fq::norm(Fq::sqr(Fq::sqr(k.psi_x))));
  }
  // the discrete-log tables of wire_fq_sqrt
  static WireTables T;
  Fq h = k.z;
  for (int i = 0; i < 38; i++) h = Fq::sqr(h);                       // z^(2^38): a primitive 256-th root of unity
  Fq p = Fq::one();
  for (int j = 0; j < 256; j++) { T.t8[j] = Fq::reduce(p); p = Fq::mul(p, h); }
  for (int j = 0; j < 1024; j++) T.slot[j] = 0xffff;
  for (int j = 0; j < 256; j++) {
    uint32_t sl = T.t8[j].l[0] & 1023u;
    while (T.slot[sl] != 0xffff) sl = (sl + 1) & 1023u;
    T.slot[sl] = (uint16_t)j;
  }
  const Fq zi = Fq::inv(k.z);
  Fq base = zi;                                                       // z^-(2^(4m)) for m = 0, 1, ...
  Fq half = zi;                                                       // z^-(2^(4m - 1)) for m >= 1 (m = 0 handled below)
  for (int m = 0; m < 12; m++) {
    Fq pb = Fq::one(), ph = Fq::one();
    for (int j = 0; j < 16; j++) {
      T.bt[m][j] = Fq::wred(Fq::norm(pb));
      pb = Fq::mul(pb, base);
      if (m == 0) { T.ht[0][j] = Fq::wred(Fq::norm(ph)); if (j & 1) ph = Fq::mul(ph, zi); }   // entry j (even) = z^-(j/2); odd entries unused
      else { T.ht[m][j] = Fq::wred(Fq::norm(ph)); ph = Fq::mul(ph, half); }
    }
    // next m: base <- base^16; half <- z^-(2^(4(m+1) - 1)) = base^8
    Fq b8 = base;
    for (int q = 0; q < 3; q++) b8 = Fq::sqr(b8);
    half = b8;
    base = Fq::sqr(b8);
  }
  k.tab = &T;
  return k;
}
inline const WireConsts& wire_consts() {
  static const WireConsts k = wire_consts_build();
  return k;
}

}  // namespace celo
