// Modular inversion by Bernstein-Yang "safegcd" division steps (constant control flow: every lane of a wave runs the same
// instruction stream whatever its operand), for the two base fields.
//
// Why: the reference's inversions are ark-ff's Fp::inverse (a binary extended Euclid, data-dependent branches: fine on a CPU core,
// a divergence storm on a SIMD); round 1 used Fermat's little theorem instead - a^(p-2), 377 squarings + ~190 products for
// BLS12-377 (3.1*10^5 instructions: a fifth of every final exponentiation), 761 + ~380 products of 28 limbs for BW6-761
// (1.9*10^6 instructions: 7 ms of a lone Groth16 verification).  62 division steps at a time on the low 64 bits of (f, g) give
// a 2x2 transition matrix that is then applied to the full-length f, g and, modulo p, to the Bezout coefficients d, e
// (the structure of libsecp256k1's modinv64, "half-delta" variant): 15 batches for 377 bits (~5*10^4 instructions), 29 for 761.
// Inputs and outputs are plain integers in [0, p) as 64-bit limbs; Fp::inv (fp.h) wraps it for the lazy Montgomery form.
// Where the reference inverts: GroupAffine normalisation (crates/bls-crypto/src/bls/signature.rs:82, public.rs:58), the easy
// part of both final exponentiations (public.rs:102, crates/epoch-snark/src/api/verifier.rs:35), hash-to-curve's cofactor step.
#pragma once
#include <cstdint>

namespace celo {

template <class P> struct SafeGcd {
  static constexpr int N64 = P::N64;
  static constexpr int top_bits() {
    int b = 0;
    for (uint64_t t = P::P64[N64 - 1]; t; t >>= 1) b++;
    return b;
  }
  static constexpr int BITS = (N64 - 1) * 64 + top_bits();
  static constexpr int LEN = (BITS + 2 + 61) / 62;                                   // signed 62-bit limbs; values in (-2p, p)
  static constexpr int BATCHES = ((45907 * BITS + 26313) / 19929 + 61) / 62;         // divsteps needed (half-delta bound) / 62
  static constexpr uint64_t M62 = (uint64_t(1) << 62) - 1;

  HD static void pack(const uint64_t* x, int64_t* r) {                                // N64 x 64 bits -> LEN x 62 bits
#pragma unroll
    for (int i = 0; i < LEN; i++) {
      const int bit = 62 * i, w = bit >> 6, off = bit & 63;
      uint64_t v = w < N64 ? x[w] >> off : 0;
      if (off > 2 && w + 1 < N64) v |= x[w + 1] << (64 - off);
      r[i] = (int64_t)(v & M62);
    }
  }
  HD static void unpack(const int64_t* r, uint64_t* x) {                              // r in [0, p)
#pragma unroll
    for (int w = 0; w < N64; w++) x[w] = 0;
#pragma unroll
    for (int i = 0; i < LEN; i++) {
      const int bit = 62 * i, w = bit >> 6, off = bit & 63;
      if (w < N64) {
        x[w] |= (uint64_t)r[i] << off;
        if (off > 2 && w + 1 < N64) x[w + 1] |= (uint64_t)r[i] >> (64 - off);
      }
    }
  }
  // out = x^-1 mod p (0 for x = 0); x in [0, p)
  HD static void inv(const uint64_t* x, uint64_t* out) {
    int64_t pm[LEN], f[LEN], g[LEN], d[LEN], e[LEN];
    uint64_t p64[N64];
#pragma unroll
    for (int i = 0; i < N64; i++) p64[i] = P::P64[i];
    pack(p64, pm);
    pack(x, g);
#pragma unroll
    for (int i = 0; i < LEN; i++) { f[i] = pm[i]; d[i] = 0; e[i] = 0; }
    e[0] = 1;
    uint64_t pinv = (uint64_t)pm[0];                                                  // p^-1 mod 2^62 (p odd): Newton, doubling the valid bits
#pragma unroll
    for (int k = 0; k < 6; k++) pinv *= 2 - (uint64_t)pm[0] * pinv;
    pinv &= M62;
    int64_t zeta = -1;                                                                // zeta = -(delta + 1/2), delta = 1/2
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int b = 0; b < BATCHES; b++) {
      // ---- 62 division steps on the low bits; (u, v; q, r) = 2^62 x the transition matrix
      uint64_t u = 1, v = 0, q = 0, r = 1;
      uint64_t ff = (uint64_t)f[0] | ((uint64_t)f[1] << 62), gg = (uint64_t)g[0] | ((uint64_t)g[1] << 62);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 2
#endif
      for (int i = 0; i < 62; i++) {
        uint64_t c1 = (uint64_t)(zeta >> 63);                                         // all ones iff zeta < 0
        const uint64_t c2 = 0 - (gg & 1);                                             // all ones iff g odd
        const uint64_t x0 = (ff ^ c1) - c1, y0 = (u ^ c1) - c1, z0 = (v ^ c1) - c1;   // (f, u, v) negated iff zeta < 0
        gg += x0 & c2; q += y0 & c2; r += z0 & c2;
        c1 &= c2;                                                                     // swap iff zeta < 0 and g odd
        zeta = (int64_t)((uint64_t)zeta ^ c1) - 1;
        ff += gg & c1; u += q & c1; v += r & c1;
        gg >>= 1; u <<= 1; v <<= 1;
      }
      const int64_t su = (int64_t)u, sv = (int64_t)v, sq = (int64_t)q, sr = (int64_t)r;
      // ---- (d, e) <- (u d + v e, q d + r e) / 2^62 mod p, kept in (-2p, p)
      {
        const int64_t sd = d[LEN - 1] >> 63, se = e[LEN - 1] >> 63;
        int64_t md = (su & sd) + (sv & se), me = (sq & sd) + (sr & se);
        __int128 cd = (__int128)su * d[0] + (__int128)sv * e[0];
        __int128 ce = (__int128)sq * d[0] + (__int128)sr * e[0];
        md -= (int64_t)((pinv * (uint64_t)cd + (uint64_t)md) & M62);
        me -= (int64_t)((pinv * (uint64_t)ce + (uint64_t)me) & M62);
        cd += (__int128)pm[0] * md;
        ce += (__int128)pm[0] * me;
        cd >>= 62; ce >>= 62;
#pragma unroll
        for (int i = 1; i < LEN; i++) {
          cd += (__int128)su * d[i] + (__int128)sv * e[i] + (__int128)pm[i] * md;
          ce += (__int128)sq * d[i] + (__int128)sr * e[i] + (__int128)pm[i] * me;
          d[i - 1] = (int64_t)((uint64_t)cd & M62); cd >>= 62;
          e[i - 1] = (int64_t)((uint64_t)ce & M62); ce >>= 62;
        }
        d[LEN - 1] = (int64_t)cd;
        e[LEN - 1] = (int64_t)ce;
      }
      // ---- (f, g) <- (u f + v g, q f + r g) / 2^62 (exact)
      {
        __int128 cf = (__int128)su * f[0] + (__int128)sv * g[0];
        __int128 cg = (__int128)sq * f[0] + (__int128)sr * g[0];
        cf >>= 62; cg >>= 62;
#pragma unroll
        for (int i = 1; i < LEN; i++) {
          cf += (__int128)su * f[i] + (__int128)sv * g[i];
          cg += (__int128)sq * f[i] + (__int128)sr * g[i];
          f[i - 1] = (int64_t)((uint64_t)cf & M62); cf >>= 62;
          g[i - 1] = (int64_t)((uint64_t)cg & M62); cg >>= 62;
        }
        f[LEN - 1] = (int64_t)cf;
        g[LEN - 1] = (int64_t)cg;
      }
    }
    // ---- g = 0 and f = +-gcd now; d = +-x^-1 in (-2p, p): fix the sign by f's, then bring it into [0, p)
    {
      const int64_t neg = f[LEN - 1] >> 63;                                           // all ones iff f < 0
      int64_t add = d[LEN - 1] >> 63;                                                 // d < 0: add p first
#pragma unroll
      for (int i = 0; i < LEN; i++) d[i] = ((d[i] + (pm[i] & add)) ^ neg) - neg;       // (limbs may leave [0, 2^62): carries below)
#pragma unroll
      for (int i = 0; i < LEN - 1; i++) { d[i + 1] += d[i] >> 62; d[i] &= (int64_t)M62; }
      add = d[LEN - 1] >> 63;
#pragma unroll
      for (int i = 0; i < LEN; i++) d[i] += pm[i] & add;
#pragma unroll
      for (int i = 0; i < LEN - 1; i++) { d[i + 1] += d[i] >> 62; d[i] &= (int64_t)M62; }
    }
    unpack(d, out);
  }
};

}  // namespace celo
