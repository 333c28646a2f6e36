// Hash-to-G1 by try-and-increment over the DIRECT hasher (Blake2s CRH + Blake2Xs-style XOF), host+device templates -
// SURVEY.md section 8f row f1.  Follows TryAndIncrement<DirectHasher, G1>::hash_with_attempt
// (crates/bls-crypto/src/hash_to_curve/try_and_increment.rs:87-139) with the deployed `compat` bit logic
// (hash_to_curve/mod.rs:146-158) and DirectHasher::{crh, xof, hash} (crates/bls-crypto/src/hashers/direct.rs:23-80):
//   for c = 0, 1, ...:  h = Blake2s(c || extra || message; personal = domain, node offset = xof length in bits 32..47)
//                       64 XOF bytes = Blake2s(h; fanout = depth = 0, leaf = inner = 32, node offset = block index | length)
//                       first 48 bytes -> x candidate and two flag bits; y from the curve equation (wire.h square root),
//                       the root picked by the flag; multiply by the G1 cofactor; first success wins.
// On the GPU (unit_hash.hip) the attempts of different counters are separate lanes / rounds, the cofactor ladder a second
// kernel; the square root is wire.h's.  Blake2s itself is pinned
// on the reference's vectors through Seam A's host implementation (hashers/direct.rs:88-172); this header is checked
// against that implementation and the oracle (tests/test_hash_gpu.py).
#pragma once
#include "wire.h"
#include "pedersen.h"

namespace celo {

struct B2sTab {
  static constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
  static constexpr uint8_t SIGMA[10][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
};

HD uint32_t b2s_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
// RFC 7693 compression F(h, m, t, last)
WIRE_FN void b2s_compress(uint32_t h[8], const uint32_t m[16], uint64_t t, bool last) {
  uint32_t v[16];
  for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2sTab::IV[i]; }
  v[12] ^= (uint32_t)t;
  v[13] ^= (uint32_t)(t >> 32);
  if (last) v[14] ^= 0xFFFFFFFFu;
#define CELO_B2S_G(a, b, c, d, x, y)                                 \
  v[a] = v[a] + v[b] + (x); v[d] = b2s_rotr(v[d] ^ v[a], 16);        \
  v[c] = v[c] + v[d];       v[b] = b2s_rotr(v[b] ^ v[c], 12);        \
  v[a] = v[a] + v[b] + (y); v[d] = b2s_rotr(v[d] ^ v[a], 8);         \
  v[c] = v[c] + v[d];       v[b] = b2s_rotr(v[b] ^ v[c], 7);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    CELO_B2S_G(0, 4, 8, 12, m[B2sTab::SIGMA[r][0]], m[B2sTab::SIGMA[r][1]])
    CELO_B2S_G(1, 5, 9, 13, m[B2sTab::SIGMA[r][2]], m[B2sTab::SIGMA[r][3]])
    CELO_B2S_G(2, 6, 10, 14, m[B2sTab::SIGMA[r][4]], m[B2sTab::SIGMA[r][5]])
    CELO_B2S_G(3, 7, 11, 15, m[B2sTab::SIGMA[r][6]], m[B2sTab::SIGMA[r][7]])
    CELO_B2S_G(0, 5, 10, 15, m[B2sTab::SIGMA[r][8]], m[B2sTab::SIGMA[r][9]])
    CELO_B2S_G(1, 6, 11, 12, m[B2sTab::SIGMA[r][10]], m[B2sTab::SIGMA[r][11]])
    CELO_B2S_G(2, 7, 8, 13, m[B2sTab::SIGMA[r][12]], m[B2sTab::SIGMA[r][13]])
    CELO_B2S_G(3, 4, 9, 14, m[B2sTab::SIGMA[r][14]], m[B2sTab::SIGMA[r][15]])
  }
#undef CELO_B2S_G
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
// h = IV ^ parameter block (no key, no salt); node_offset is the 48-bit field
HD void b2s_init(uint32_t h[8], uint8_t digest_length, uint8_t fanout, uint8_t depth, uint32_t leaf_length, uint64_t node_offset,
                 uint8_t node_depth, uint8_t inner_length, const uint8_t personal[8]) {
  uint32_t pb[8];
  pb[0] = (uint32_t)digest_length | ((uint32_t)fanout << 16) | ((uint32_t)depth << 24);
  pb[1] = leaf_length;
  pb[2] = (uint32_t)node_offset;
  pb[3] = (uint32_t)((node_offset >> 32) & 0xFFFF) | ((uint32_t)node_depth << 16) | ((uint32_t)inner_length << 24);
  pb[4] = 0;
  pb[5] = 0;
  pb[6] = (uint32_t)personal[0] | ((uint32_t)personal[1] << 8) | ((uint32_t)personal[2] << 16) | ((uint32_t)personal[3] << 24);
  pb[7] = (uint32_t)personal[4] | ((uint32_t)personal[5] << 8) | ((uint32_t)personal[6] << 16) | ((uint32_t)personal[7] << 24);
  for (int i = 0; i < 8; i++) h[i] = B2sTab::IV[i] ^ pb[i];
}
// the byte string counter || extra || message of one try-and-increment attempt
struct TaiBytes {
  uint8_t c;
  const uint8_t* extra;
  size_t elen;
  const uint8_t* msg;
  size_t mlen;
  HD size_t size() const { return 1 + elen + mlen; }
  HD uint8_t operator()(size_t j) const { return j == 0 ? c : (j <= elen ? extra[j - 1] : msg[j - 1 - elen]); }
};
// whole-message Blake2s over a byte source (the final block is the last 1..64 bytes, or an empty one for len = 0)
template <class Src> HD void b2s_stream(uint32_t h[8], const Src& src) {
  const size_t len = src.size();
  size_t off = 0;
  uint64_t t = 0;
  uint32_t m[16];
  while (len - off > 64) {
    for (int w = 0; w < 16; w++) {
      uint32_t v = 0;
      for (int b = 3; b >= 0; b--) v = (v << 8) | src(off + 4 * w + b);
      m[w] = v;
    }
    t += 64;
    b2s_compress(h, m, t, false);
    off += 64;
  }
  for (int w = 0; w < 16; w++) {
    uint32_t v = 0;
    for (int b = 3; b >= 0; b--) {
      const size_t j = off + 4 * w + b;
      v = (v << 8) | (j < len ? src(j) : 0);
    }
    m[w] = v;
  }
  t += len - off;
  b2s_compress(h, m, t, true);
}

HD uint64_t b2x_node_offset(uint64_t i, uint32_t xof_len) { return i | ((uint64_t)(xof_len & 0xFFFF) << 32); }

// the first 48 XOF bytes (twelve little-endian words) -> the curve point they select, before the cofactor: the `compat` flag
// logic of hash_to_curve/mod.rs:146-158 and get_point_from_x.  Shared by every hasher (Seam A's composite path calls it on
// the host, so the reference's compat hash-to-G1 vectors pin it).  false: not a field element, the flagged zero, or no y.
HD bool tai_point_from_xof(const uint32_t w12[12], const WireConsts& k, Affine<Fq>& p) {
  uint32_t b47 = w12[11] >> 24;                                          // byte 47
  if (b47 & 2) b47 |= 0x80; else b47 &= 0x7F;                            // `compat`: the y-sign flag is taken from bit 377
  const uint32_t flags = b47 & 0xC0;
  b47 &= 0x01;                                                           // bits below MODULUS_BITS = 377
  uint64_t w[6];
  for (int i = 0; i < 5; i++) w[i] = (uint64_t)w12[2 * i] | ((uint64_t)w12[2 * i + 1] << 32);
  w[5] = (uint64_t)w12[10] | ((uint64_t)((w12[11] & 0x00FFFFFFu) | (b47 << 24)) << 32);
  if (wire_cmp(w, P377::P64, 6) >= 0) return false;
  const Fq x = Fq::from_canonical(w);
  if (x.is_zero_mod_p() && (flags & 0x40)) return false;                 // the zero point scales to zero
  Fq y;
  if (!wire_fq_sqrt(Fq::norm(Fq::add(Fq::mul(Fq::sqr(x), x), Fq::one())), k, y)) return false;
  if (wire_lex_largest(y) != ((flags & 0x80) != 0)) y = wire_neg(y);    // get_point_from_x(x, greatest)
  p = {Fq::norm(x), Fq::norm(y)};
  return true;
}
// one attempt: counter c -> the curve point (x, y) the candidate bytes select, before the cofactor; false when the
// candidate is not a field element, is the flagged zero, or x^3 + 1 is not a square.
// mode TAI_DIRECT:    TryAndIncrement<DirectHasher>: candidate = xof(crh(c || extra || message)), Blake2s CRH
// mode TAI_XOF_ONLY:  the CIP22 loop (try_and_increment_cip22.rs:81-134): `message` is the inner CRH computed once by the
//                     caller (the composite hasher's 48 bytes), candidate = xof(c || extra || inner) - no CRH per attempt
// mode TAI_COMPOSITE: TryAndIncrement<CompositeHasher> before CIP22: candidate = xof(pedersen_crh(c || extra || message)),
//                     `gens` = the generator table (pedersen.h)
enum TaiMode : int { TAI_DIRECT = 0, TAI_XOF_ONLY = 1, TAI_COMPOSITE = 2 };
HD bool tai_candidate(const uint8_t dom[8], const uint8_t* msg, size_t mlen, const uint8_t* extra, size_t elen, int c, const WireConsts& k,
                      Affine<Fq>& p, int mode = TAI_DIRECT, const EdPoint* gens = nullptr) {
  uint32_t x0[8], x1[8];
  const TaiBytes src = {(uint8_t)c, extra, elen, msg, mlen};
  if (mode == TAI_XOF_ONLY) {
    b2s_init(x0, 32, 0, 0, 32, b2x_node_offset(0, 64), 0, 32, dom);
    b2s_stream(x0, src);
    b2s_init(x1, 32, 0, 0, 32, b2x_node_offset(1, 64), 0, 32, dom);
    b2s_stream(x1, src);
  } else if (mode == TAI_COMPOSITE) {
    uint8_t pre[48];
    pedersen_crh_src(gens, src, pre);
    const PtrBytes ps = {pre, 48};
    b2s_init(x0, 32, 0, 0, 32, b2x_node_offset(0, 64), 0, 32, dom);
    b2s_stream(x0, ps);
    b2s_init(x1, 32, 0, 0, 32, b2x_node_offset(1, 64), 0, 32, dom);
    b2s_stream(x1, ps);
  } else {
    uint32_t h[8], m[16];
    b2s_init(h, 32, 1, 1, 0, b2x_node_offset(0, 64), 0, 0, dom);           // DirectHasher::crh with hash_length(48) = 64
    b2s_stream(h, src);
    for (int i = 0; i < 8; i++) { m[i] = h[i]; m[i + 8] = 0; }
    b2s_init(x0, 32, 0, 0, 32, b2x_node_offset(0, 64), 0, 32, dom);        // DirectHasher::xof, blocks 0 and 1
    b2s_compress(x0, m, 32, true);
    b2s_init(x1, 32, 0, 0, 32, b2x_node_offset(1, 64), 0, 32, dom);
    b2s_compress(x1, m, 32, true);
  }
  uint32_t w12[12];
  for (int i = 0; i < 8; i++) w12[i] = x0[i];
  for (int i = 0; i < 4; i++) w12[8 + i] = x1[i];
  return tai_point_from_xof(w12, k, p);
}
// scale_by_cofactor + affine normalisation; false when the multiple is the identity (the reference then tries the next counter)
HD bool tai_finish(const Affine<Fq>& p, Affine<Fq>& out) {
  const uint64_t cof[2] = {0x0000000000000000ULL, 0x170b5d4430000000ULL};   // (x - 1)^2 / 3, 125 bits, bit 124 set
  Xyzz<Fq> s = Xyzz<Fq>::from_affine(p);
  for (int i = 123; i >= 0; i--) {
    s = xyzz_dbl(s);
    if ((cof[i >> 6] >> (i & 63)) & 1) xyzz_madd(s, p);
  }
  if (s.is_identity() || s.ZZ.is_zero_mod_p()) return false;
  const Fq t = wire_inv(Fq::mul(s.ZZ, s.ZZZ));
  out.x = Fq::norm(Fq::mul(s.X, Fq::mul(t, s.ZZZ)));
  out.y = Fq::norm(Fq::mul(s.Y, Fq::mul(t, s.ZZ)));
  return true;
}
// the serial loop (host single hashes, and the GPU path's fallback from counter c_start): -> affine point of the prime-order
// subgroup and the attempt counter; false when no counter below 255 succeeds (the reference errs there)
HD bool hash_to_g1_direct_tai(const uint8_t dom[8], const uint8_t* msg, size_t mlen, const uint8_t* extra, size_t elen, const WireConsts& k,
                              Affine<Fq>& out, int& attempt, int c_start = 0, int mode = TAI_DIRECT, const EdPoint* gens = nullptr) {
  for (int c = c_start; c < 255; c++) {
    Affine<Fq> p = {Fq::zero(), Fq::zero()};
    if (!tai_candidate(dom, msg, mlen, extra, elen, c, k, p, mode, gens)) continue;
    if (!tai_finish(p, out)) continue;
    attempt = c;
    return true;
  }
  return false;
}

}  // namespace celo
