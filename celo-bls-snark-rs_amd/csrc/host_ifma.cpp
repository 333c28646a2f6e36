// Host epilogue of the big MSM on AVX-512 IFMA: the Horner recombination of the per-window results that msm.h's run_device
// does after the last kernel (~250 dependent XYZZ doublings + ~250 additions for a 253-bit scalar; it replaces nothing in the
// reference - arkworks' VariableBaseMSM ends with the same window recombination).  One dependent chain
// cannot be spread over cores, but the field products INSIDE one point operation are independent in rounds of up to six
// (add-2008-s: 4 rounds, dbl-2008-s-1: 3 rounds), and vpmadd52{l,h}uq multiplies eight 52-bit-limb operands side by side: a
// round of <= 8 products costs what ONE 64-bit-limb Montgomery product costs in host64.h.  Compiled as a plain C++ unit with
// the IFMA target flags (the rest of the library is not); msm.h calls it only when the CPU reports avx512ifma and falls back
// to host64.h otherwise - and whenever this code meets a special case of the group law (equal or opposite operands: the
// difference of two x coordinates is 0 mod p), which it only detects and never handles (return value 1).
//
// Representation: radix 2^52, NL limbs (8 for the 377-bit field, 15 for the 761-bit one), Montgomery form with R' = 2^(52 NL);
// eight elements per V (limb j of all eight in one zmm register).  Values are lazy: products come out < 2p with limbs < 2^52,
// sums and differences are carried (signed) but not reduced; every product input stays < 16 p << 2^19 p (a b < R' p).
#include <immintrin.h>
#include <cstdint>
#include <cstring>
#include "fp_consts.h"

namespace {

typedef unsigned __int128 u128;
constexpr uint64_t M52 = (uint64_t(1) << 52) - 1;

template <int NL> struct V { __m512i l[NL]; };

// ---- scalar big-number helpers (set-up only): little-endian 64-bit words
template <int N> static bool geq(const uint64_t* a, const uint64_t* b) {
  for (int i = N - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i];
  return true;
}
template <int N> static void sub_in_place(uint64_t* a, const uint64_t* b) {
  u128 br = 0;
  for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - b[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
// x <- 2 x mod p (x < p < 2^(64 N - 1))
template <int N> static void dbl_mod(uint64_t* x, const uint64_t* p) {
  uint64_t c = 0;
  for (int i = 0; i < N; i++) { uint64_t n = (x[i] << 1) | c; c = x[i] >> 63; x[i] = n; }
  if (geq<N>(x, p)) sub_in_place<N>(x, p);
}
template <int N> static void pow2_mod(int e, const uint64_t* p, uint64_t* out) {
  memset(out, 0, 8 * N);
  out[0] = 1;
  for (int i = 0; i < e; i++) dbl_mod<N>(out, p);
}
template <int N, int NL> static void to52(const uint64_t* w, uint64_t* l) {   // N 64-bit words -> NL 52-bit limbs
  for (int j = 0; j < NL; j++) {
    const int bit = 52 * j, wi = bit >> 6, off = bit & 63;
    uint64_t v = wi < N ? w[wi] >> off : 0;
    if (off > 12 && wi + 1 < N) v |= w[wi + 1] << (64 - off);
    l[j] = v & M52;
  }
}
template <int N, int NL> static void from52(const uint64_t* l, uint64_t* w) {   // canonical limbs (< 2^52) -> words
  memset(w, 0, 8 * N);
  for (int j = 0; j < NL; j++) {
    const int bit = 52 * j, wi = bit >> 6, off = bit & 63;
    if (wi < N) w[wi] |= l[j] << off;
    if (off > 12 && wi + 1 < N) w[wi + 1] |= l[j] >> (64 - off);
  }
}

template <class P, int NL_> struct Ifma {
  static constexpr int N = P::N64, NL = NL_;
  typedef V<NL> Vn;

  struct Consts {
    uint64_t p[NL];        // modulus, radix 2^52
    uint64_t kp[4][NL];    // 2p, 4p, 8p, 16p (limbs may exceed 52 bits: used in carried subtractions only)
    uint64_t inv;          // -p^-1 mod 2^52
    uint64_t c_in[NL];     // 2^(52 NL + (52 NL - 64 N)) mod p: MontMul by it turns x 2^(64 N) into x 2^(52 NL)
    uint64_t c_out[NL];    // 2^(64 N) mod p:                    MontMul by it turns x 2^(52 NL) into x 2^(64 N)
    uint64_t p0k[17];      // low limbs of k p, k = 0 .. 16 (zero filter)
    Consts() {
      to52<N, NL>(P::P64, p);
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < NL; j++) kp[k][j] = p[j] << (k + 1);
      uint64_t x = 1;
      for (int i = 0; i < 6; i++) x *= 2 - P::P64[0] * x;
      inv = ((uint64_t)0 - x) & M52;
      uint64_t t[N];
      pow2_mod<N>(2 * 52 * NL - 64 * N, P::P64, t);
      to52<N, NL>(t, c_in);
      pow2_mod<N>(64 * N, P::P64, t);
      to52<N, NL>(t, c_out);
      for (int k = 0; k <= 16; k++) p0k[k] = (p[0] * (uint64_t)k) & M52;
    }
  };
  static const Consts& C() { static const Consts c; return c; }

  static inline Vn bcast(const uint64_t* limbs) {
    Vn r;
    for (int j = 0; j < NL; j++) r.l[j] = _mm512_set1_epi64((long long)limbs[j]);
    return r;
  }
  // signed carry propagation: limbs in (-2^63, 2^63), value >= 0 and < 2^(52 NL) -> limbs in [0, 2^52)
  static inline void carry(Vn& a) {
    const __m512i m = _mm512_set1_epi64((long long)M52);
    for (int j = 0; j + 1 < NL; j++) {
      const __m512i c = _mm512_srai_epi64(a.l[j], 52);
      a.l[j] = _mm512_and_si512(a.l[j], m);
      a.l[j + 1] = _mm512_add_epi64(a.l[j + 1], c);
    }
  }
  static inline Vn add(const Vn& a, const Vn& b) {
    Vn r;
    for (int j = 0; j < NL; j++) r.l[j] = _mm512_add_epi64(a.l[j], b.l[j]);
    carry(r);
    return r;
  }
  // a - b + 2^(k+1) p, carried (b < 2^(k+1) p)
  template <int K> static inline Vn sub(const Vn& a, const Vn& b) {
    const Consts& c = C();
    Vn r;
    for (int j = 0; j < NL; j++)
      r.l[j] = _mm512_add_epi64(_mm512_sub_epi64(a.l[j], b.l[j]), _mm512_set1_epi64((long long)c.kp[K][j]));
    carry(r);
    return r;
  }
  // Montgomery product of eight pairs, operand limbs < 2^52, a b < R' p  ->  < 2p, limbs < 2^52
  static inline Vn mul(const Vn& a, const Vn& b) {
    const Consts& c = C();
    const __m512i zero = _mm512_setzero_si512(), inv = _mm512_set1_epi64((long long)c.inv);
    __m512i pl[NL];
    for (int j = 0; j < NL; j++) pl[j] = _mm512_set1_epi64((long long)c.p[j]);
    __m512i t[NL + 1];
    for (int j = 0; j <= NL; j++) t[j] = zero;
#pragma GCC unroll 16
    for (int i = 0; i < NL; i++) {
      const __m512i bi = b.l[i];
#pragma GCC unroll 16
      for (int j = 0; j < NL; j++) {
        t[j] = _mm512_madd52lo_epu64(t[j], a.l[j], bi);
        t[j + 1] = _mm512_madd52hi_epu64(t[j + 1], a.l[j], bi);
      }
      const __m512i m = _mm512_madd52lo_epu64(zero, t[0], inv);
#pragma GCC unroll 16
      for (int j = 0; j < NL; j++) {
        t[j] = _mm512_madd52lo_epu64(t[j], m, pl[j]);
        t[j + 1] = _mm512_madd52hi_epu64(t[j + 1], m, pl[j]);
      }
      // t[0] is now a multiple of 2^52: fold its carry into the next limb and shift the window down
      t[1] = _mm512_add_epi64(t[1], _mm512_srli_epi64(t[0], 52));
#pragma GCC unroll 16
      for (int j = 0; j < NL; j++) t[j] = t[j + 1];
      t[NL] = zero;
    }
    Vn r;
    for (int j = 0; j < NL; j++) r.l[j] = t[j];
    carry(r);
    return r;
  }
  // lanes of the result taken from the 16 lanes of (s0, s1): idx[k] in 0 .. 15
  static inline Vn pick(const Vn& s0, const Vn& s1, __m512i idx) {
    Vn r;
    for (int j = 0; j < NL; j++) r.l[j] = _mm512_permutex2var_epi64(s0.l[j], idx, s1.l[j]);
    return r;
  }
  static inline __m512i IDX(int a, int b = 0, int c = 0, int d = 0, int e = 0, int f = 0, int g = 0, int h = 0) {
    return _mm512_setr_epi64(a, b, c, d, e, f, g, h);
  }
  // bit k of the result: lane k may be 0 mod p (its low limb equals that of some j p, j <= 16); lanes hold values < 16 p
  static inline unsigned maybe_zero(const Vn& a) {
    const Consts& c = C();
    unsigned m = 0;
    for (int k = 0; k <= 16; k++) m |= _mm512_cmpeq_epi64_mask(a.l[0], _mm512_set1_epi64((long long)c.p0k[k]));
    return m;
  }

  // a point: lanes 0..3 of a Vn = X, Y, ZZ, ZZZ (the other lanes carry don't-care values)
  // doubling (dbl-2008-s-1), three product rounds; returns false if Y may be 0 mod p
  static inline bool dbl(Vn& a) {
    const Vn U = add(a, a);                                            // lane 1: U = 2 Y
    if (maybe_zero(a) & 2u) return false;
    const Vn A1 = pick(U, a, IDX(1, 8 + 0));                           // [U, X]
    const Vn T1 = mul(A1, A1);                                         // [V, XX]
    const Vn XX2 = add(T1, T1), M3 = add(XX2, T1);                     // lane 1: M = 3 XX   (< 6p)
    const Vn A2 = pick(U, pick(a, M3, IDX(0, 8 + 1)), IDX(1, 8 + 0, 8 + 1));          // [U, X, M]
    const Vn B2 = pick(T1, M3, IDX(0, 0, 8 + 1));                      // [V, V, M]
    const Vn T2 = mul(A2, B2);                                         // [W, S, M2]
    const Vn S2 = add(T2, T2);                                         // lane 1: 2 S (< 4p)
    const Vn X3 = sub<1>(pick(T2, T2, IDX(2)), pick(S2, S2, IDX(1)));  // lane 0: M2 - 2S + 4p   (< 6p)
    const Vn SmX = sub<2>(pick(T2, T2, IDX(1)), X3);                   // lane 0: S - X3 + 8p     (< 10p)
    // [M, W, V, W] x [S - X3, Y, ZZ, ZZZ]
    const Vn A3 = pick(pick(M3, T2, IDX(1, 8 + 0)), T1, IDX(0, 1, 8 + 0, 1));
    const Vn B3 = pick(SmX, a, IDX(0, 8 + 1, 8 + 2, 8 + 3));
    const Vn T3 = mul(A3, B3);                                         // [Y3a, Y3b, ZZ3, ZZZ3]
    const Vn Y3 = sub<0>(T3, pick(T3, T3, IDX(1)));                    // lane 0: Y3a - Y3b + 2p  (< 4p)
    a = pick(pick(X3, Y3, IDX(0, 8 + 0)), T3, IDX(0, 1, 8 + 2, 8 + 3));
    return true;
  }
  // a += b (add-2008-s), four product rounds; b's lanes start at bo (0 or 4); false if the operands may be equal or opposite
  static inline bool add_pt(Vn& a, const Vn& b, int bo) {
    const Vn A1 = pick(a, b, IDX(0, 8 + bo + 0, 1, 8 + bo + 1, 2, 3));                    // [X1, X2, Y1, Y2, ZZ1, ZZZ1]
    const Vn B1 = pick(a, b, IDX(8 + bo + 2, 2, 8 + bo + 3, 3, 8 + bo + 2, 8 + bo + 3));  // [ZZ2, ZZ1, ZZZ2, ZZZ1, ZZ2, ZZZ2]
    const Vn T1 = mul(A1, B1);                                                            // [U1, U2, S1, S2, ZZp, ZZZp]
    const Vn D = sub<0>(pick(T1, T1, IDX(1, 3)), pick(T1, T1, IDX(0, 2)));                // [Pd, R]      (< 4p)
    if (maybe_zero(D) & 1u) return false;
    const Vn T2 = mul(D, D);                                                              // [PP, R2]
    const Vn A3 = pick(D, T1, IDX(0, 8 + 0, 8 + 4));                                      // [Pd, U1, ZZp]
    const Vn B3 = pick(T2, T2, IDX(0, 0, 0));
    const Vn T3 = mul(A3, B3);                                                            // [PPP, Q, ZZ3]
    const Vn Q2 = add(T3, T3);                                                            // lane 1: 2Q
    const Vn s = add(T3, pick(Q2, Q2, IDX(1)));                                           // lane 0: PPP + 2Q (< 6p)
    const Vn X3 = sub<2>(pick(T2, T2, IDX(1)), s);                                        // lane 0: R2 - s + 8p (< 10p)
    const Vn QmX = sub<3>(pick(T3, T3, IDX(1)), X3);                                      // lane 0: Q - X3 + 16p (< 18p)
    const Vn A4 = pick(D, T1, IDX(1, 8 + 2, 8 + 5));                                      // [R, S1, ZZZp]
    const Vn B4 = pick(QmX, T3, IDX(0, 8 + 0, 8 + 0));                                    // [Q - X3, PPP, PPP]
    const Vn T4 = mul(A4, B4);                                                            // [Y3a, Y3b, ZZZ3]
    const Vn Y3 = sub<0>(T4, pick(T4, T4, IDX(1)));                                       // lane 0: Y3a - Y3b + 2p
    a = pick(pick(X3, Y3, IDX(0, 8 + 0)), pick(T3, T4, IDX(2, 8 + 2)), IDX(0, 1, 8 + 0, 8 + 1));
    return true;
  }

  // pts: slots of `stride` u64 (X, Y, ZZ, ZZZ at multiples of N words, arkworks Montgomery form, ZZ = 0 words: identity);
  // step k: acc = 2 acc, then acc += slot order[k] (-1: no addend; bit 30 set: this step does not double).  out: 4 N words.
  static int horner(const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out, int* out_inf) {
    const Consts& c = C();
    const Vn cin = bcast(c.c_in), cout = bcast(c.c_out);
    Vn acc;
    for (int j = 0; j < NL; j++) acc.l[j] = _mm512_setzero_si512();
    bool inf = true;
    alignas(64) uint64_t lanes[NL][8];
    for (int k = 0; k < steps; k++) {
      const bool nodbl = order[k] >= 0 && (order[k] & 0x40000000);
      if (!inf && !nodbl && !dbl(acc)) return 1;
      if (order[k] < 0) continue;
      const uint64_t* q = pts + (size_t)(order[k] & 0x3FFFFFFF) * stride;
      uint64_t zz = 0;
      for (int i = 0; i < N; i++) zz |= q[2 * N + i];
      if (zz == 0) continue;
      for (int e = 0; e < 4; e++) {
        uint64_t l[NL];
        to52<N, NL>(q + e * N, l);
        for (int j = 0; j < NL; j++) lanes[j][e] = l[j];
      }
      Vn b;
      for (int j = 0; j < NL; j++) b.l[j] = _mm512_load_si512((const void*)lanes[j]);
      b = mul(b, cin);
      if (inf) { acc = b; inf = false; continue; }
      if (!add_pt(acc, b, 0)) return 1;
    }
    *out_inf = inf ? 1 : 0;
    if (inf) { memset(out, 0, 4 * N * 8); return 0; }
    Vn r = mul(acc, cout);                      // < 2p, arkworks' Montgomery form
    for (int j = 0; j < NL; j++) _mm512_store_si512((void*)lanes[j], r.l[j]);
    for (int e = 0; e < 4; e++) {
      uint64_t l[NL], w[N];
      for (int j = 0; j < NL; j++) l[j] = lanes[j][e];
      from52<N, NL>(l, w);
      if (geq<N>(w, P::P64)) sub_in_place<N>(w, P::P64);
      memcpy(out + e * N, w, 8 * N);
    }
    return 0;
  }
};

}  // namespace

// The C entry points msm.h calls: 0 = done (out: X, Y, ZZ, ZZZ; *inf = 1: identity), 1 = special case met, use host64.h.
extern "C" int celo_ifma_horner_377(const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out, int* inf) {
  return Ifma<celo::P377, 8>::horner(pts, stride, order, steps, out, inf);
}
extern "C" int celo_ifma_horner_761(const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out, int* inf) {
  return Ifma<celo::P761, 15>::horner(pts, stride, order, steps, out, inf);
}
