// Bowe-Hopwood-Pedersen CRH evaluation over the twisted Edwards curve ed-on-BW6-761 (-x^2 + y^2 = 1 + 79743 x^2 y^2 over Fq of
// BLS12-377) - the composite hasher's CRH (crates/bls-crypto/src/hashers/composite.rs:79-86: bowe_hopwood::CRH::evaluate with
// WINDOW_SIZE = 93, NUM_WINDOWS = 560), host+device templates.  The message is cut into 3-bit chunks (LSB-first, zero
// padded); chunk ch uses generator ch of the window-major table (generator j of a window = 16^j * its base) and contributes
// (1 + b0 + 2 b1) * g, negated when b2 is set; the hash is the affine x coordinate of the sum, 48 bytes little-endian.
// The generator table itself is built on the host (seam_a.hip: ChaCha20 stream exactly as the reference consumes it) and
// handed to the device once.  It holds FOUR entries per chunk - g, 2g, 3g, 4g (PEDERSEN_MULTIPLES) - so that a chunk's contribution
// (1 + b0 + 2 b1) g is one table entry and costs ONE point addition; round 2 kept g alone and formed the multiple per chunk (up to two
// additions and a doubling before the one that counts: 2.5 point operations per chunk on average).  Same group element, same hash.  One source for Seam A's host path (hash_crh, the composite hashers) and the bulk GPU kernel
// (unit_hash.hip: k_pedersen_crh, one message per lane).
#pragma once
#include "wire.h"

namespace celo {

struct SF {  // "safe" field element: every result weak-reduced and normalised (the sums dominate; no lazy-bound bookkeeping)
  Fq v;
  HD static SF from(const Fq& x) { return {Fq::wred(Fq::norm(x))}; }
  HD SF operator+(const SF& o) const { return from(Fq::add(v, o.v)); }
  HD SF operator-(const SF& o) const { return from(Fq::sub<4, 1>(v, o.v)); }
  HD SF operator*(const SF& o) const { return from(Fq::mul(v, o.v)); }
  HD SF neg() const { return from(Fq::neg<4, 1>(v)); }
  HD SF dbl() const { return from(Fq::add(v, v)); }
};
struct EdPoint { SF X, Y, Z, T; };  // extended twisted Edwards, a = -1
HD SF sf_small(uint64_t k) { uint64_t w[6] = {k, 0, 0, 0, 0, 0}; return SF::from(Fq::from_canonical(w)); }
WIRE_FN EdPoint ed_add(const EdPoint& p, const EdPoint& q) {  // add-2008-hwcd-3 (a = -1)
  const SF d2 = sf_small(2 * 79743);
  SF A = (p.Y - p.X) * (q.Y - q.X), B = (p.Y + p.X) * (q.Y + q.X), C = p.T * d2 * q.T, D = (p.Z * q.Z).dbl();
  SF E = B - A, F = D - C, G = D + C, H = B + A;
  return {E * F, G * H, F * G, E * H};
}
WIRE_FN EdPoint ed_dbl(const EdPoint& p) {  // dbl-2008-hwcd (a = -1)
  SF A = p.X * p.X, B = p.Y * p.Y, C = (p.Z * p.Z).dbl(), D = A.neg();
  SF E = (p.X + p.Y) * (p.X + p.Y) - A - B, G = D + B, F = G - C, H = D - B;
  return {E * F, G * H, F * G, E * H};
}
HD EdPoint ed_neg(const EdPoint& p) { return {p.X.neg(), p.Y, p.Z, p.T.neg()}; }
HD EdPoint ed_zero() { return {sf_small(0), sf_small(1), sf_small(1), sf_small(0)}; }

constexpr int PEDERSEN_WINDOW_SIZE = 93, PEDERSEN_NUM_WINDOWS = 560, PEDERSEN_MULTIPLES = 4;
constexpr size_t PEDERSEN_MAX_BITS = (size_t)PEDERSEN_WINDOW_SIZE * PEDERSEN_NUM_WINDOWS * 3;

// out48 = x coordinate of sum_ch enc_ch over the bytes src(0 .. src.size()); the caller has checked size * 8 <= PEDERSEN_MAX_BITS
// (the reference panics beyond).  Src: any byte source with size() and operator()(j) (a plain buffer, or the counter || extra ||
// message string of a try-and-increment attempt).
struct PtrBytes {
  const uint8_t* p;
  size_t n;
  HD size_t size() const { return n; }
  HD uint8_t operator()(size_t j) const { return p[j]; }
};
template <class Src> HD void pedersen_crh_src(const EdPoint* gens, const Src& src, uint8_t out48[48]) {
  const size_t len = src.size(), nbits = len * 8, nchunks = (nbits + 2) / 3;
  EdPoint total = ed_zero();
  for (size_t ch = 0; ch < nchunks; ch++) {
    const size_t b = 3 * ch;
    // three message bits, LSB-first within a byte, zero beyond the end
    uint32_t two = src(b >> 3);
    if ((b >> 3) + 1 < len) two |= (uint32_t)src((b >> 3) + 1) << 8;
    const uint32_t bits = (two >> (b & 7)) & 7u;
    EdPoint enc = gens[(size_t)PEDERSEN_MULTIPLES * ch + (bits & 3u)];     // (1 + b0 + 2 b1) * generator ch
    if (bits & 4) enc = ed_neg(enc);
    total = ed_add(total, enc);
  }
  const SF x = total.X * SF::from(wire_inv(total.Z.v));
  uint64_t w[6];
  x.v.to_canonical(w);
  for (int i = 0; i < 48; i++) out48[i] = (uint8_t)(w[i >> 3] >> (8 * (i & 7)));
}
HD void pedersen_crh(const EdPoint* gens, const uint8_t* msg, size_t len, uint8_t out48[48]) { pedersen_crh_src(gens, PtrBytes{msg, len}, out48); }

}  // namespace celo
