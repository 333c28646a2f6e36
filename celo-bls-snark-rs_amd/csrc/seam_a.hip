// Seam A: the bls-snark-sys C ABI, rebuilt on top of the gfx950 hot path.
//
// This file provides the handle / wire-format / aggregation half of `crates/bls-snark-sys/src/{serialization,signatures}.rs`
// (opaque PrivateKey / PublicKey / Signature handles, arkworks CanonicalSerialize encodings, aggregate_*), plus the GPU
// verification cores that the reference's verify_* symbols reduce to once the message has been hashed to G1.
// Not yet exported: the hashers (Blake2Xs try-and-increment, Bowe-Hopwood composite),
// hence verify_signature / verify_pop / batch_verify_signature / batch_verify_strict / sign_* / hash_* under their
// reference names, and the epoch-encoding symbols.  Everything here is host orchestration; group arithmetic that is on
// the hot path (MSM, pairings) goes to the kernels, the rest (one decompression, one subgroup check) is plumbing.
//
// Ownership mirrors the reference: handles come from new/delete behind destroy_*; byte buffers are malloc'd and released
// by free_vec(ptr, len) (crates/bls-snark-sys/src/serialization.rs:120-140, 224-268).  Every entry returns `false`
// instead of unwinding (convert_result_to_bool, crates/bls-snark-sys/src/lib.rs:21-27).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <random>
#include <mutex>
#include <vector>
#include "curve.h"
#include "fp2.h"
#include "../../include/celo_bls_amd.h"
#include "../../include/celo_bls_snark_sys.h"

using namespace celo;

typedef Fp<P377> Fq_;
typedef Fp2<P377> Fq2_;

struct PrivateKey { uint64_t k[4]; };          // Fr, canonical
struct PublicKey { uint64_t xyz[36]; };         // G2 Jacobian, arkworks Montgomery limbs (GroupProjective<g2>)
struct Signature { uint64_t xyz[18]; };         // G1 Jacobian

namespace {
// errors are logged and mapped to `false` like the reference's convert_result_to_bool (log::error! + false)
void log_err(const char* what) { if (getenv("CELO_AMD_LOG")) fprintf(stderr, "[celo-amd] %s\n", what); }
const uint64_t R_ORDER[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};

int cmp_n(const uint64_t* a, const uint64_t* b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}
bool fq_from_bytes(const uint8_t* in, Fq_& out) {  // 48 LE bytes, canonical (< p) required
  uint64_t w[6];
  memcpy(w, in, 48);
  if (cmp_n(w, P377::P64, 6) >= 0) return false;
  out = Fq_::from_canonical(w);
  return true;
}
void fq_to_bytes(const Fq_& a, uint8_t* out) {
  uint64_t w[6];
  a.to_canonical(w);
  memcpy(out, w, 48);
}
bool fq_lex_largest(const Fq_& a) {  // canonical(a) > (p-1)/2
  uint64_t w[6];
  a.to_canonical(w);
  return cmp_n(w, P377::PM1_HALF64, 6) > 0;
}
bool fq_is_zero(const Fq_& a) { return a.is_zero_mod_p(); }
bool fq_eq(const Fq_& a, const Fq_& b) { return Fq_::eq_mod_p(Fq_::norm(a), Fq_::norm(b)); }

// Tonelli-Shanks over Fq (q - 1 = 2^46 * t)
struct SqrtCtx {
  uint64_t t[6], t_plus1_half[6], pm1_half[6];
  Fq_ z;  // (non-residue)^t
  SqrtCtx() {
    uint64_t pm1[6];
    memcpy(pm1, P377::P64, 48);
    pm1[0] -= 1;
    // t = (p-1) >> 46
    for (int i = 0; i < 6; i++) t[i] = (pm1[i] >> 46) | (i + 1 < 6 ? pm1[i + 1] << 18 : 0);
    uint64_t tp1[6];
    memcpy(tp1, t, 48);
    tp1[0] += 1;  // t odd, no carry out of limb 0 unless all ones (not the case)
    for (int i = 0; i < 6; i++) t_plus1_half[i] = (tp1[i] >> 1) | (i + 1 < 6 ? tp1[i + 1] << 63 : 0);
    memcpy(pm1_half, P377::PM1_HALF64, 48);
    for (uint64_t g = 2;; g++) {
      uint64_t gw[6] = {g, 0, 0, 0, 0, 0};
      Fq_ gf = Fq_::from_canonical(gw);
      Fq_ l = Fq_::pow64(gf, pm1_half, 6);
      if (!fq_eq(l, Fq_::one())) { z = Fq_::pow64(gf, t, 6); break; }
    }
  }
};
const SqrtCtx& sqrt_ctx() { static SqrtCtx c; return c; }

bool fq_sqrt(const Fq_& a_, Fq_& out) {
  const Fq_ a = Fq_::norm(a_);
  if (fq_is_zero(a)) { out = Fq_::zero(); return true; }
  const SqrtCtx& c = sqrt_ctx();
  if (!fq_eq(Fq_::pow64(a, c.pm1_half, 6), Fq_::one())) return false;
  Fq_ x = Fq_::pow64(a, c.t_plus1_half, 6);
  Fq_ b = Fq_::pow64(a, c.t, 6);
  Fq_ zz = c.z;
  int m = 46;
  while (!fq_eq(b, Fq_::one())) {
    int i = 0;
    Fq_ b2 = b;
    while (!fq_eq(b2, Fq_::one())) { b2 = Fq_::sqr(b2); i++; }
    Fq_ g = zz;
    for (int k = 0; k < m - i - 1; k++) g = Fq_::sqr(g);
    x = Fq_::mul(x, g);
    zz = Fq_::sqr(g);
    b = Fq_::mul(b, zz);
    m = i;
  }
  out = x;
  return true;
}
Fq_ fq_neg(const Fq_& a) { return Fq_::wred(Fq_::norm(Fq_::neg<64, 1>(Fq_::norm(a)))); }  // weak-reduced: keeps the affine-coordinate bound (vb <= 3)
Fq_ fq_inv_of_small(uint64_t k) {
  uint64_t w[6] = {k, 0, 0, 0, 0, 0};
  return Fq_::inv(Fq_::from_canonical(w));
}
// sqrt in Fq2 = Fq[u]/(u^2+5) (complex method generalised to u^2 = -5)
bool fq2_sqrt(const Fq2_& a, Fq2_& out) {
  if (a.is_zero_mod_p()) { out = Fq2_::zero(); return true; }
  static const Fq_ inv2 = fq_inv_of_small(2);
  static const Fq_ inv5 = fq_inv_of_small(5);
  Fq_ a0 = Fq_::norm(a.c0), a1 = Fq_::norm(a.c1);
  if (fq_is_zero(a1)) {
    Fq_ s;
    if (fq_sqrt(a0, s)) { out = {s, Fq_::zero()}; return true; }
    // a0 = -5 t^2  ->  sqrt = t u
    Fq_ tt = fq_neg(Fq_::mul(a0, inv5));
    if (!fq_sqrt(tt, s)) return false;
    out = {Fq_::zero(), s};
    return true;
  }
  // norm = a0^2 + 5 a1^2
  Fq_ s1 = Fq_::sqr(a1);
  Fq_ n = Fq_::norm(Fq_::add(Fq_::sqr(a0), Fq_::norm(Fq_::add(Fq_::dbl(Fq_::dbl(s1)), s1))));
  Fq_ al;
  if (!fq_sqrt(n, al)) return false;
  Fq_ d = Fq_::mul(Fq_::norm(Fq_::add(a0, al)), inv2), x0;
  if (!fq_sqrt(d, x0)) {
    d = Fq_::mul(Fq_::norm(Fq_::sub<4, 1>(a0, Fq_::norm(al))), inv2);
    if (!fq_sqrt(d, x0)) return false;
  }
  Fq_ x1 = Fq_::mul(a1, Fq_::inv(Fq_::norm(Fq_::dbl(x0))));
  out = {x0, x1};
  Fq2_ chk = Fq2_::sqr(out);
  return fq_eq(chk.c0, a0) && fq_eq(chk.c1, a1);
}
bool fq2_lex_largest(const Fq2_& y) {  // arkworks: compare c1 first, then c0
  if (!y.c1.is_zero_mod_p()) return fq_lex_largest(y.c1);
  return fq_lex_largest(y.c0);
}

// ---- group helpers on the host (plumbing: one decompression / subgroup check / small sums)
template <class F> Xyzz<F> scalar_mul_host(const Affine<F>& p, const uint64_t* k, int nlimbs) {
  Xyzz<F> acc = Xyzz<F>::identity();
  for (int i = nlimbs * 64 - 1; i >= 0; i--) {
    acc = xyzz_dbl(acc);
    if ((k[i >> 6] >> (i & 63)) & 1) xyzz_madd(acc, p);
  }
  return acc;
}
template <class F> bool in_subgroup(const Affine<F>& p) {
  Xyzz<F> r = scalar_mul_host(p, R_ORDER, 4);
  return r.is_identity() || r.ZZ.is_zero_mod_p();
}
template <class F> void affine_to_jac(const Affine<F>& p, uint64_t* out) {
  constexpr int A = F::ARK64;
  p.x.to_ark(out);
  p.y.to_ark(out + A);
  F::one().to_ark(out + 2 * A);
}
template <class F> void identity_jac(uint64_t* out) {
  constexpr int A = F::ARK64;
  F::zero().to_ark(out);
  F::one().to_ark(out + A);
  F::zero().to_ark(out + 2 * A);
}
// Jacobian (ark limbs) -> affine; returns false for the identity
template <class F> bool jac_to_affine(const uint64_t* jac, Affine<F>& out) {
  constexpr int A = F::ARK64;
  F Z = F::from_ark(jac + 2 * A);
  if (Z.is_zero_mod_p()) return false;
  F zi = F::inv(Z);
  F zi2 = F::sqr(zi);
  out.x = F::norm(F::mul(F::from_ark(jac), zi2));
  out.y = F::norm(F::mul(F::from_ark(jac + A), F::mul(zi2, zi)));
  return true;
}
uint8_t* alloc_bytes(size_t n) { return (uint8_t*)malloc(n ? n : 1); }
bool emit(const std::vector<uint8_t>& v, uint8_t** out_bytes, int* out_len) {
  uint8_t* p = alloc_bytes(v.size());
  if (!p) return false;
  memcpy(p, v.data(), v.size());
  *out_bytes = p;
  *out_len = (int)v.size();
  return true;
}

// ---- G1 (48-byte x, flags in the top two bits of the last byte)
bool g1_decompress(const uint8_t* in, Affine<Fq_>& p, bool& inf) {
  uint8_t buf[48];
  memcpy(buf, in, 48);
  uint8_t flags = buf[47] & 0xC0;
  buf[47] &= 0x3F;
  inf = (flags & 0x40) != 0;
  if (inf) return true;
  Fq_ x;
  if (!fq_from_bytes(buf, x)) return false;
  Fq_ rhs = Fq_::norm(Fq_::add(Fq_::mul(Fq_::sqr(x), x), Fq_::one())), y;
  if (!fq_sqrt(rhs, y)) return false;
  if (fq_lex_largest(y) != ((flags & 0x80) != 0)) y = fq_neg(y);
  p = {Fq_::norm(x), Fq_::norm(y)};
  return true;
}
void g1_compress(const Affine<Fq_>& p, bool inf, uint8_t* out) {
  memset(out, 0, 48);
  if (inf) { out[47] |= 0x40; return; }
  fq_to_bytes(p.x, out);
  if (fq_lex_largest(p.y)) out[47] |= 0x80;
}
// ---- G2 (96-byte x = c0 || c1, flags on c1's last byte)
Fq2_ twist_b() {
  static const Fq_ inv5 = fq_inv_of_small(5);
  return {Fq_::zero(), fq_neg(inv5)};
}
bool g2_decompress(const uint8_t* in, Affine<Fq2_>& p, bool& inf) {
  uint8_t buf[96];
  memcpy(buf, in, 96);
  uint8_t flags = buf[95] & 0xC0;
  buf[95] &= 0x3F;
  inf = (flags & 0x40) != 0;
  if (inf) return true;
  Fq2_ x;
  if (!fq_from_bytes(buf, x.c0) || !fq_from_bytes(buf + 48, x.c1)) return false;
  Fq2_ rhs = Fq2_::norm(Fq2_::add(Fq2_::mul(Fq2_::sqr(x), x), twist_b())), y;
  if (!fq2_sqrt(rhs, y)) return false;
  if (fq2_lex_largest(y) != ((flags & 0x80) != 0)) y = {fq_neg(y.c0), fq_neg(y.c1)};
  p = {Fq2_::norm(x), Fq2_::norm(y)};
  return true;
}
void g2_compress(const Affine<Fq2_>& p, bool inf, uint8_t* out) {
  memset(out, 0, 96);
  if (inf) { out[95] |= 0x40; return; }
  fq_to_bytes(p.x.c0, out);
  fq_to_bytes(p.x.c1, out + 48);
  if (fq2_lex_largest(p.y)) out[95] |= 0x80;
}
bool on_curve_g1(const Affine<Fq_>& p) { return fq_eq(Fq_::sqr(p.y), Fq_::norm(Fq_::add(Fq_::mul(Fq_::sqr(p.x), p.x), Fq_::one()))); }
bool on_curve_g2(const Affine<Fq2_>& p) {
  Fq2_ l = Fq2_::sqr(p.y), r = Fq2_::norm(Fq2_::add(Fq2_::mul(Fq2_::sqr(p.x), p.x), twist_b()));
  return fq_eq(l.c0, r.c0) && fq_eq(l.c1, r.c1);
}
}  // namespace

extern "C" {

bool init(void) { return celo_amd_init(0) == 0; }

// ---------------------------------------------------------------- keys (crates/bls-snark-sys/src/signatures.rs:19-42)
bool generate_private_key(PrivateKey** out_private_key) {
  if (!out_private_key) return false;
  std::random_device rd;
  PrivateKey* sk = new PrivateKey;
  for (;;) {
    for (int i = 0; i < 4; i++) sk->k[i] = ((uint64_t)rd() << 32) | rd();
    sk->k[3] &= (1ULL << 61) - 1;  // 253 bits
    if (cmp_n(sk->k, R_ORDER, 4) < 0) break;
  }
  *out_private_key = sk;
  return true;
}
bool celo_amd_g2_generator(uint64_t out_xy[24]);
bool private_key_to_public_key(const PrivateKey* in_private_key, PublicKey** out_public_key) {
  if (!in_private_key || !out_public_key) return false;
  uint64_t gen[24];
  if (!celo_amd_g2_generator(gen)) return false;
  Affine<Fq2_> g = {Fq2_::from_ark(gen), Fq2_::from_ark(gen + 12)};
  Xyzz<Fq2_> r = scalar_mul_host(g, in_private_key->k, 4);
  PublicKey* pk = new PublicKey;
  if (r.is_identity()) identity_jac<Fq2_>(pk->xyz);
  else {
    Fq2_::mul(r.X, r.ZZ).to_ark(pk->xyz);
    Fq2_::mul(r.Y, r.ZZZ).to_ark(pk->xyz + 12);
    r.ZZ.to_ark(pk->xyz + 24);
  }
  *out_public_key = pk;
  return true;
}

// ---------------------------------------------------------------- (de)serialisation (serialization.rs:13-117)
bool deserialize_private_key(const uint8_t* in_bytes, int in_len, PrivateKey** out) {
  if (!in_bytes || !out || in_len < 32) return false;
  PrivateKey* sk = new PrivateKey;
  memcpy(sk->k, in_bytes, 32);
  if (cmp_n(sk->k, R_ORDER, 4) >= 0) { delete sk; return false; }
  *out = sk;
  return true;
}
bool serialize_private_key(const PrivateKey* in, uint8_t** out_bytes, int* out_len) {
  if (!in || !out_bytes || !out_len) return false;
  std::vector<uint8_t> v(32);
  memcpy(v.data(), in->k, 32);
  return emit(v, out_bytes, out_len);
}
bool deserialize_public_key(const uint8_t* in_bytes, int in_len, PublicKey** out) {
  if (!in_bytes || !out || in_len < 96) return false;
  Affine<Fq2_> p;
  bool inf;
  if (!g2_decompress(in_bytes, p, inf)) { log_err("deserialize_public_key: not a valid compressed G2 point"); return false; }
  PublicKey* pk = new PublicKey;
  if (inf) identity_jac<Fq2_>(pk->xyz);
  else {
    if (!in_subgroup(p)) { delete pk; log_err("deserialize_public_key: point not in the prime-order subgroup"); return false; }
    affine_to_jac(p, pk->xyz);
  }
  *out = pk;
  return true;
}
bool deserialize_public_key_cached(const uint8_t* in_bytes, int in_len, PublicKey** out) {
  // the reference memoises decompression in an LRU (serialization.rs:44-61); decoding is a pure function, so no cache is
  // observable through the ABI
  return deserialize_public_key(in_bytes, in_len, out);
}
bool serialize_public_key(const PublicKey* in, uint8_t** out_bytes, int* out_len) {
  if (!in || !out_bytes || !out_len) return false;
  Affine<Fq2_> p;
  bool fin = jac_to_affine<Fq2_>(in->xyz, p);
  std::vector<uint8_t> v(96);
  g2_compress(p, !fin, v.data());
  return emit(v, out_bytes, out_len);
}
bool serialize_public_key_uncompressed(const PublicKey* in, uint8_t** out_bytes, int* out_len) {
  if (!in || !out_bytes || !out_len) return false;
  Affine<Fq2_> p;
  bool fin = jac_to_affine<Fq2_>(in->xyz, p);
  std::vector<uint8_t> v(192, 0);
  if (fin) {
    fq_to_bytes(p.x.c0, v.data()); fq_to_bytes(p.x.c1, v.data() + 48);
    fq_to_bytes(p.y.c0, v.data() + 96); fq_to_bytes(p.y.c1, v.data() + 144);
  } else v[191] |= 0x40;
  return emit(v, out_bytes, out_len);
}
bool deserialize_signature(const uint8_t* in_bytes, int in_len, Signature** out) {
  if (!in_bytes || !out || in_len < 48) return false;
  Affine<Fq_> p;
  bool inf;
  if (!g1_decompress(in_bytes, p, inf)) { log_err("deserialize_signature: not a valid compressed G1 point"); return false; }
  Signature* s = new Signature;
  if (inf) identity_jac<Fq_>(s->xyz);
  else {
    if (!in_subgroup(p)) { delete s; log_err("deserialize_signature: point not in the prime-order subgroup"); return false; }
    affine_to_jac(p, s->xyz);
  }
  *out = s;
  return true;
}
bool serialize_signature(const Signature* in, uint8_t** out_bytes, int* out_len) {
  if (!in || !out_bytes || !out_len) return false;
  Affine<Fq_> p;
  bool fin = jac_to_affine<Fq_>(in->xyz, p);
  std::vector<uint8_t> v(48);
  g1_compress(p, !fin, v.data());
  return emit(v, out_bytes, out_len);
}
bool serialize_signature_uncompressed(const Signature* in, uint8_t** out_bytes, int* out_len) {
  if (!in || !out_bytes || !out_len) return false;
  Affine<Fq_> p;
  bool fin = jac_to_affine<Fq_>(in->xyz, p);
  std::vector<uint8_t> v(96, 0);
  if (fin) { fq_to_bytes(p.x, v.data()); fq_to_bytes(p.y, v.data() + 48); }
  else v[95] |= 0x40;
  return emit(v, out_bytes, out_len);
}
// 96-byte x||y -> 48-byte compressed (serialization.rs:167-189); 192 -> 96 (serialization.rs:192-218)
bool compress_signature(const uint8_t* in, int in_len, uint8_t** out, int* out_len) {
  if (!in || !out || !out_len || in_len < 96) return false;
  Affine<Fq_> p;
  if (!fq_from_bytes(in, p.x) || !fq_from_bytes(in + 48, p.y)) return false;
  std::vector<uint8_t> v(48);
  g1_compress(p, false, v.data());
  return emit(v, out, out_len);
}
bool compress_pubkey(const uint8_t* in, int in_len, uint8_t** out, int* out_len) {
  if (!in || !out || !out_len || in_len < 192) return false;
  Affine<Fq2_> p;
  if (!fq_from_bytes(in, p.x.c0) || !fq_from_bytes(in + 48, p.x.c1) || !fq_from_bytes(in + 96, p.y.c0) || !fq_from_bytes(in + 144, p.y.c1))
    return false;
  std::vector<uint8_t> v(96);
  g2_compress(p, false, v.data());
  return emit(v, out, out_len);
}

// ---------------------------------------------------------------- destructors (serialization.rs:224-268)
bool destroy_private_key(PrivateKey* p) { if (!p) return false; delete p; return true; }
bool destroy_public_key(PublicKey* p) { if (!p) return false; delete p; return true; }
bool destroy_signature(Signature* p) { if (!p) return false; delete p; return true; }
bool free_vec(uint8_t* bytes, int len) { (void)len; if (!bytes) return false; free(bytes); return true; }

// ---------------------------------------------------------------- aggregation (signatures.rs:428-505)
bool aggregate_public_keys(const PublicKey* const* in, int n, PublicKey** out) {
  if (!out || n < 0 || (n > 0 && !in)) return false;
  std::vector<uint64_t> buf((size_t)n * 36);
  for (int i = 0; i < n; i++) { if (!in[i]) return false; memcpy(&buf[(size_t)i * 36], in[i]->xyz, 288); }
  PublicKey* pk = new PublicKey;
  if (celo_amd_sum_jacobian_bls12_377_g2(buf.data(), (size_t)n, pk->xyz) != 0) { delete pk; return false; }
  *out = pk;
  return true;
}
bool aggregate_public_keys_subtract(const PublicKey* agg, const PublicKey* const* in, int n, PublicKey** out) {
  if (!agg || !out || n < 0 || (n > 0 && !in)) return false;
  std::vector<uint64_t> buf((size_t)(n + 1) * 36);
  memcpy(buf.data(), agg->xyz, 288);
  for (int i = 0; i < n; i++) {
    if (!in[i]) return false;
    uint64_t* d = &buf[(size_t)(i + 1) * 36];
    memcpy(d, in[i]->xyz, 288);
    Fq2_ y = Fq2_::from_ark(d + 12);                       // negate: (X, -Y, Z)
    Fq2_ ny = {fq_neg(y.c0), fq_neg(y.c1)};
    ny.to_ark(d + 12);
  }
  PublicKey* pk = new PublicKey;
  if (celo_amd_sum_jacobian_bls12_377_g2(buf.data(), (size_t)n + 1, pk->xyz) != 0) { delete pk; return false; }
  *out = pk;
  return true;
}
bool aggregate_signatures(const Signature* const* in, int n, Signature** out) {
  if (!out || n < 0 || (n > 0 && !in)) return false;
  std::vector<uint64_t> buf((size_t)n * 18);
  for (int i = 0; i < n; i++) { if (!in[i]) return false; memcpy(&buf[(size_t)i * 18], in[i]->xyz, 144); }
  Signature* s = new Signature;
  if (celo_amd_sum_jacobian_bls12_377_g1(buf.data(), (size_t)n, s->xyz) != 0) { delete s; return false; }
  *out = s;
  return true;
}

// ---------------------------------------------------------------- GPU verification cores (the part of verify_* after hashing)
// message_hash_xy: H(m) as an affine G1 point in arkworks limbs (12 u64) — what hash_to_g1.hash(..) returns in
// PublicKey::verify_sig (crates/bls-crypto/src/bls/public.rs:108) after into_affine().
bool celo_amd_g2_generator(uint64_t out_xy[24]) {
  Fq_::from_limbs(T377::G2_GEN_X0).to_ark(out_xy);
  Fq_::from_limbs(T377::G2_GEN_X1).to_ark(out_xy + 6);
  Fq_::from_limbs(T377::G2_GEN_Y0).to_ark(out_xy + 12);
  Fq_::from_limbs(T377::G2_GEN_Y1).to_ark(out_xy + 18);
  return true;
}
bool celo_amd_verify_hash(const PublicKey* pk, const uint64_t* message_hash_xy, const Signature* sig, bool* out_verified) {
  if (!pk || !message_hash_xy || !sig || !out_verified) return false;
  Affine<Fq_> s;
  Affine<Fq2_> p;
  uint8_t inf1[2] = {0, 0}, inf2[2] = {0, 0};
  uint64_t g1[24], g2[48];
  memset(g1, 0, sizeof g1);
  memset(g2, 0, sizeof g2);
  if (jac_to_affine<Fq_>(sig->xyz, s)) { s.x.to_ark(g1); s.y.to_ark(g1 + 6); } else inf1[0] = 1;
  memcpy(g1 + 12, message_hash_xy, 96);
  uint64_t gen[24];
  celo_amd_g2_generator(gen);
  Fq2_ gy = Fq2_::from_ark(gen + 12);
  Fq2_ ngy = {fq_neg(gy.c0), fq_neg(gy.c1)};
  memcpy(g2, gen, 96);
  ngy.to_ark(g2 + 12);
  if (jac_to_affine<Fq2_>(pk->xyz, p)) { p.x.to_ark(g2 + 24); p.y.to_ark(g2 + 36); } else inf2[1] = 1;
  int one = 0;
  if (pairing_product_is_one_bls12_377(g1, inf1, g2, inf2, 2, &one) != 0) return false;
  *out_verified = one != 0;
  return true;
}
}
