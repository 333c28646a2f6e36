// Seam A: the bls-snark-sys C ABI, rebuilt on top of the gfx950 hot path.
//
// This file provides all 36 symbols of `crates/bls-snark-sys/src/{serialization,signatures}.rs` and `src/snark/{mod,epoch_block}.rs`
// under their reference names: opaque PrivateKey / PublicKey / Signature handles, arkworks CanonicalSerialize encodings,
// aggregate_*, both hashers (Blake2Xs try-and-increment, the Bowe-Hopwood composite hasher, before and after CIP22),
// sign_* / verify_* / batch_verify_*, the Groth16 `verify` and the epoch encoders.  Everything here is host orchestration:
// group arithmetic on the hot path (MSM, pairings, bulk hashing from 256 messages up) goes to the kernels through the
// Seam B entry points, the rest (one decompression, one subgroup check, one hash) is latency-path plumbing on the host.
//
// Ownership mirrors the reference: handles come from new/delete behind destroy_*; byte buffers are malloc'd and released
// by free_vec(ptr, len) (crates/bls-snark-sys/src/serialization.rs:120-140, 224-268).  Every entry returns `false`
// instead of unwinding (convert_result_to_bool, crates/bls-snark-sys/src/lib.rs:21-27).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <chrono>
#include <mutex>
#include <vector>
#include <unordered_set>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <list>
#include <string>
#include <unordered_map>
#include <algorithm>
#include <sys/random.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include "curve.h"
#include "fp2.h"
#include "wire.h"
#include "hash_direct.h"
#include "pedersen.h"
#include "host64.h"
#include "runtime.h"
#include "../../include/celo_bls_amd.h"
#include "../../include/celo_bls_snark_sys.h"

using namespace celo;

typedef Fp<P377> Fq_;
typedef Fp2<P377> Fq2_;

struct PrivateKey { uint64_t k[4]; };          // Fr, canonical
// PublicKey / Signature handles live in ARENAS (HandleArena below): a handle is its slot of the arena plus the serial number of the
// allocation that filled the slot.  batch_verify_strict keeps a per-device mirror of the arenas' points in HBM (affine, one entry per
// slot, tagged with the serial it was uploaded for), so a call over handles the device has already seen ships 4-byte slot numbers, not
// 320 bytes per signer: validator keys recur epoch after epoch (the reason the reference memoises their decompression,
// crates/bls-crypto/src/bls/cache.rs:36), and the handle types are opaque to every caller.
struct PublicKey { uint64_t xyz[36]; uint64_t serial; uint32_t slot; };         // G2 Jacobian, arkworks Montgomery limbs (GroupProjective<g2>)
struct Signature { uint64_t xyz[18]; uint64_t serial; uint32_t slot; };         // G1 Jacobian

namespace {
// errors are logged and mapped to `false` like the reference's convert_result_to_bool (log::error! + false)
void log_err(const char* what) { if (getenv("CELO_AMD_LOG")) fprintf(stderr, "[celo-amd] %s\n", what); }
const uint64_t R_ORDER[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};

// Chunked slab with a free list.  A handle's address never moves (chunks are never reallocated or released), its contents are written
// once, by the entry point that creates it, before the caller sees it; release() puts the slot back and the next alloc() of that slot
// carries a new serial, which is what marks the device mirrors' copy of the slot stale.  Serials are 64-bit and never 0.
template <class T> struct HandleArena {
  static constexpr uint32_t CHUNK = 1u << 14;
  std::mutex mu;
  std::vector<T*> chunks;
  std::vector<uint32_t> free_slots;
  uint32_t next = 0;
  uint64_t serial = 0;
  T* alloc() {
    std::lock_guard<std::mutex> lk(mu);
    uint32_t s;
    if (!free_slots.empty()) { s = free_slots.back(); free_slots.pop_back(); }
    else {
      if (next == 0xffffffffu) return nullptr;
      if ((size_t)next == chunks.size() * CHUNK) {
        T* c = (T*)malloc((size_t)CHUNK * sizeof(T));
        if (!c) return nullptr;
        try { chunks.push_back(c); } catch (...) { free(c); return nullptr; }
      }
      s = next++;
    }
    T* p = &chunks[s / CHUNK][s % CHUNK];
    p->slot = s;
    p->serial = ++serial;
    return p;
  }
  bool release(T* p) {
    std::lock_guard<std::mutex> lk(mu);
    if (p->serial == 0) return false;                           // destroyed already: the slot must not enter the free list twice
    p->serial = 0;
    try { free_slots.push_back(p->slot); } catch (...) {}       // out of memory: the slot is lost, nothing else
    return true;
  }
  uint32_t high_water() { std::lock_guard<std::mutex> lk(mu); return next; }
};
// Persistent host workers for the per-call passes over 10^6 handles: creating 64 threads costs 1.5-4 ms per call (measured inside
// batch_verify_strict at config-3 scale), waking 64 sleeping ones ~0.1 ms.  One job at a time; a caller that finds the pool busy (a
// concurrent call on another device) or without threads runs its ranges on threads of its own / inline as before.  The pool object is
// never destroyed and its threads are detached: nothing to join at process exit.
struct HostPool {
  std::mutex mu, busy;
  std::condition_variable cv, cv_done;
  std::function<void(unsigned)> job;
  unsigned n = 0, next = 0, finished = 0, nthreads = 0;
  std::atomic<bool> job_failed{false};          // a range of the current job threw (out of memory): the submitter treats the call as failed
  pid_t owner = 0;                              // a fork()ed child has this object but none of its threads: it must not wait for them
  static HostPool& get() { static HostPool* p = new HostPool; return *p; }
  HostPool() {
    owner = getpid();
    unsigned hw = std::thread::hardware_concurrency();
    if (hw > 64) hw = 64;
    for (unsigned i = 0; i < hw; i++) {
      try { std::thread([this]() { loop(); }).detach(); nthreads++; } catch (...) { break; }
    }
  }
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return next < n; });
      const unsigned i = next++;
      lk.unlock();
      try { job(i); } catch (...) { job_failed = true; }
      lk.lock();
      if (++finished == n) cv_done.notify_all();
    }
  }
  bool try_begin(unsigned count, std::function<void(unsigned)> fn) {
    if (count == 0 || count > nthreads || getpid() != owner || !busy.try_lock()) return false;
    { std::lock_guard<std::mutex> lk(mu); job = std::move(fn); n = count; next = 0; finished = 0; job_failed = false; }
    cv.notify_all();
    return true;
  }
  bool finish() {      // false: a range of the job threw
    bool ok;
    { std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return finished == n; }); n = 0; next = 0; finished = 0; job = nullptr; ok = !job_failed.load(); }
    busy.unlock();
    return ok;
  }
};

HandleArena<PublicKey>& pk_arena() { static HandleArena<PublicKey> a; return a; }
HandleArena<Signature>& sig_arena() { static HandleArena<Signature> a; return a; }
PublicKey* new_public_key() { return pk_arena().alloc(); }
Signature* new_signature() { return sig_arena().alloc(); }
bool drop(PublicKey* p) { return pk_arena().release(p); }
bool drop(Signature* p) { return sig_arena().release(p); }

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

// square roots, decoding and the subgroup check live in wire.h (shared with the bulk GPU kernels of unit_wire.hip)
bool fq_sqrt(const Fq_& a, Fq_& out) { return wire_fq_sqrt(a, wire_consts(), out); }
Fq_ fq_neg(const Fq_& a) { return Fq_::wred(Fq_::norm(Fq_::neg<64, 1>(Fq_::norm(a)))); }  // weak-reduced: keeps the affine-coordinate bound (vb <= 3)
Fq_ fq_inv_of_small(uint64_t k) {
  uint64_t w[6] = {k, 0, 0, 0, 0, 0};
  return Fq_::inv(Fq_::from_canonical(w));
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
template <class F> bool in_subgroup(const Affine<F>& p) { return wire_in_subgroup(p, wire_consts()); }
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
// ... and for callers that must carry the identity the way arkworks does: GroupAffine::zero() is (x, y, infinity) = (0, 1, true), and the
// reference's encode_public_key (crates/epoch-snark/src/encoding.rs:23-47) reads x and y of whatever into_affine() returned - for the identity
// 754 zero bits and a clear sign bit.  (It documents "not the point at infinity" as an assumption and does not check it.)
template <class F> Affine<F> jac_to_affine_or_zero(const uint64_t* jac) {
  Affine<F> out;
  if (!jac_to_affine<F>(jac, out)) out = {F::zero(), F::one()};
  return out;
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
  const WireStatus st = wire_decode_g1(in, wire_consts(), false, p);
  inf = st == WIRE_INFINITY;
  return st != WIRE_INVALID;
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
  const WireStatus st = wire_decode_g2(in, wire_consts(), false, p);
  inf = st == WIRE_INFINITY;
  return st != WIRE_INVALID;
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

// ---------------------------------------------------------------- DirectHasher (crates/bls-crypto/src/hashers/direct.rs:8-80)
// Blake2s (RFC 7693) with an explicit parameter block: the XOF uses fanout = depth = 0, leaf/inner length 32 and the
// BLAKE2X node-offset trick (xof length in bits 32..47 of the 48-bit node offset).  Host plumbing.
const uint32_t B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void b2s_compress(uint32_t h[8], const uint8_t block[64], uint64_t t, bool last) {
  uint32_t m[16], v[16];
  for (int i = 0; i < 16; i++) memcpy(&m[i], block + 4 * i, 4);
  for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2S_IV[i]; }
  v[12] ^= (uint32_t)t;
  v[13] ^= (uint32_t)(t >> 32);
  if (last) v[14] ^= 0xFFFFFFFFu;
#define B2S_G(a, b, c, d, x, y)                                   \
  v[a] = v[a] + v[b] + (x); v[d] = rotr32(v[d] ^ v[a], 16);       \
  v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 12);       \
  v[a] = v[a] + v[b] + (y); v[d] = rotr32(v[d] ^ v[a], 8);        \
  v[c] = v[c] + v[d];       v[b] = rotr32(v[b] ^ v[c], 7);
  for (int r = 0; r < 10; r++) {
    const uint8_t* s = B2S_SIGMA[r];
    B2S_G(0, 4, 8, 12, m[s[0]], m[s[1]]) B2S_G(1, 5, 9, 13, m[s[2]], m[s[3]])
    B2S_G(2, 6, 10, 14, m[s[4]], m[s[5]]) B2S_G(3, 7, 11, 15, m[s[6]], m[s[7]])
    B2S_G(0, 5, 10, 15, m[s[8]], m[s[9]]) B2S_G(1, 6, 11, 12, m[s[10]], m[s[11]])
    B2S_G(2, 7, 8, 13, m[s[12]], m[s[13]]) B2S_G(3, 4, 9, 14, m[s[14]], m[s[15]])
  }
#undef B2S_G
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
struct B2sParams { uint8_t digest_length = 32, fanout = 1, depth = 1, node_depth = 0, inner_length = 0; uint32_t leaf_length = 0; uint64_t node_offset = 0; };
std::vector<uint8_t> blake2s(const uint8_t* data, size_t len, const B2sParams& p, const uint8_t* personal, size_t plen) {
  uint8_t pb[32];
  memset(pb, 0, 32);
  pb[0] = p.digest_length; pb[1] = 0; pb[2] = p.fanout; pb[3] = p.depth;
  memcpy(pb + 4, &p.leaf_length, 4);
  for (int i = 0; i < 6; i++) pb[8 + i] = (uint8_t)(p.node_offset >> (8 * i));
  pb[14] = p.node_depth; pb[15] = p.inner_length;
  memcpy(pb + 24, personal, plen < 8 ? plen : 8);
  uint32_t h[8];
  for (int i = 0; i < 8; i++) { uint32_t w; memcpy(&w, pb + 4 * i, 4); h[i] = B2S_IV[i] ^ w; }
  uint64_t t = 0;
  size_t off = 0;
  while (len - off > 64) { t += 64; b2s_compress(h, data + off, t, false); off += 64; }
  uint8_t lastb[64];
  memset(lastb, 0, 64);
  if (len - off) memcpy(lastb, data + off, len - off);
  t += len - off;
  b2s_compress(h, lastb, t, true);
  std::vector<uint8_t> out(p.digest_length);
  uint8_t full[32];
  memcpy(full, h, 32);
  memcpy(out.data(), full, p.digest_length);
  return out;
}
uint64_t xof_node_offset(uint64_t i, size_t xof_len) { return i | ((uint64_t)(xof_len & 0xFF) << 32) | ((uint64_t)((xof_len >> 8) & 0xFF) << 40); }
std::vector<uint8_t> direct_crh(const uint8_t* dom, size_t dlen, const uint8_t* msg, size_t mlen, size_t xof_len) {
  B2sParams p;
  p.node_offset = xof_node_offset(0, xof_len);
  return blake2s(msg, mlen, p, dom, dlen);
}
std::vector<uint8_t> direct_xof(const uint8_t* dom, size_t dlen, const uint8_t* hashed, size_t hlen, size_t xof_len) {
  size_t n = (xof_len + 31) / 32;
  std::vector<uint8_t> out;
  for (size_t i = 0; i < n; i++) {
    B2sParams p;
    p.digest_length = (uint8_t)((i == n - 1 && (xof_len % 32)) ? xof_len % 32 : 32);
    p.fanout = 0; p.depth = 0; p.leaf_length = 32; p.inner_length = 32;
    p.node_offset = xof_node_offset(i, xof_len);
    std::vector<uint8_t> h = blake2s(hashed, hlen, p, dom, dlen);
    out.insert(out.end(), h.begin(), h.end());
  }
  return out;
}
std::vector<uint8_t> direct_hash(const uint8_t* dom, size_t dlen, const uint8_t* msg, size_t mlen, size_t n) {
  std::vector<uint8_t> c = direct_crh(dom, dlen, msg, mlen, n);
  return direct_xof(dom, dlen, c.data(), c.size(), n);
}
const uint8_t SIG_DOMAIN[8] = {'U', 'L', 'f', 'o', 'r', 'x', 'o', 'f'};  // crates/bls-crypto/src/lib.rs:75
const uint8_t POP_DOMAIN[8] = {'U', 'L', 'f', 'o', 'r', 'p', 'o', 'p'};  // lib.rs:78

// TryAndIncrement<DirectHasher, G1>::hash_with_attempt with the deployed `compat` bit logic
// (crates/bls-crypto/src/hash_to_curve/try_and_increment.rs:87-139, mod.rs:146-158)
bool hash_to_g1_direct(const uint8_t* dom, const uint8_t* msg, size_t mlen, const uint8_t* extra, size_t elen, Affine<Fq_>& out, int& attempt) {
  return hash_to_g1_direct_tai(dom, msg, mlen, extra, elen, wire_consts(), out, attempt);   // hash_direct.h: the source the GPU kernels run
}

// ---------------------------------------------------------------- CompositeHasher (crates/bls-crypto/src/hashers/composite.rs)
// Bowe-Hopwood-Pedersen CRH over the twisted Edwards curve ed-on-BW6-761 (-x^2 + y^2 = 1 + 79743 x^2 y^2 over Fq of
// BLS12-377), WINDOW_SIZE = 93, NUM_WINDOWS = 560, generators drawn from ChaCha20 seeded with
// Blake2s("ULTRALIGHT PRNG SEED", personal "UL_prngs") exactly as rand 0.7 / rand_chacha 0.2 / ark-ff 0.1 consume it, then
// the Blake2Xs XOF.  Host plumbing; pinned on the reference's CRH vector and its 20 compat hash-to-G1
// vectors through the oracle restatement (tests/test_oracle_golden.py, tests/test_seam_a.py).
struct ChaCha20Rng {  // rand_chacha 0.2 behind rand_core 0.5 BlockRng: 64-word buffer (4 blocks), 64-bit block counter
  uint32_t key[8]; uint64_t counter = 0; uint32_t buf[64]; int idx = 64;
  static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  void block(uint32_t* out) {
    uint32_t s[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), 0, 0};
    uint32_t w[16];
    memcpy(w, s, sizeof w);
#define CC_QR(a, b, c, d) w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12); \
                          w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7);
    for (int r = 0; r < 10; r++) {
      CC_QR(0, 4, 8, 12) CC_QR(1, 5, 9, 13) CC_QR(2, 6, 10, 14) CC_QR(3, 7, 11, 15)
      CC_QR(0, 5, 10, 15) CC_QR(1, 6, 11, 12) CC_QR(2, 7, 8, 13) CC_QR(3, 4, 9, 14)
    }
#undef CC_QR
    for (int i = 0; i < 16; i++) out[i] = w[i] + s[i];
    counter++;
  }
  void generate() { for (int b = 0; b < 4; b++) block(buf + 16 * b); idx = 0; }
  uint32_t next_u32() { if (idx >= 64) generate(); return buf[idx++]; }
  uint64_t next_u64() {
    if (idx < 63) { uint64_t v = ((uint64_t)buf[idx + 1] << 32) | buf[idx]; idx += 2; return v; }
    if (idx >= 64) { generate(); uint64_t v = ((uint64_t)buf[1] << 32) | buf[0]; idx = 2; return v; }
    uint64_t lo = buf[63]; generate(); uint64_t hi = buf[0]; idx = 1; return (hi << 32) | lo;
  }
};
// A ChaCha20 stream seeded with 32 bytes from the operating system (getrandom): what rand::thread_rng() is in the reference
// (batch.rs:51 draws the batching exponents from it).  One system call per FFI call instead of one per exponent word.
bool os_seeded_rng(ChaCha20Rng& rng) {
  uint8_t seed[32];
  size_t got = 0;
  while (got < sizeof seed) {
    ssize_t r = getrandom(seed + got, sizeof seed - got, 0);
    if (r <= 0) return false;
    got += (size_t)r;
  }
  memcpy(rng.key, seed, 32);
  return true;
}
struct CompositeParams {
  std::vector<EdPoint> gens;  // [NUM_WINDOWS * WINDOW_SIZE][PEDERSEN_MULTIPLES]: window-major, generator j = 16^j * base, then its multiples 1..4
  static constexpr int WINDOW_SIZE = PEDERSEN_WINDOW_SIZE, NUM_WINDOWS = PEDERSEN_NUM_WINDOWS;
  CompositeParams() {
    static const uint8_t PERS[8] = {'U', 'L', '_', 'p', 'r', 'n', 'g', 's'};
    static const char* SEED_MSG = "ULTRALIGHT PRNG SEED";
    B2sParams p;
    std::vector<uint8_t> seed = blake2s((const uint8_t*)SEED_MSG, strlen(SEED_MSG), p, PERS, 8);
    ChaCha20Rng rng;
    memcpy(rng.key, seed.data(), 32);
    const SF one = sf_small(1), dcoef = sf_small(79743);
    gens.reserve((size_t)WINDOW_SIZE * NUM_WINDOWS * PEDERSEN_MULTIPLES);
    for (int w = 0; w < NUM_WINDOWS; w++) {
      EdPoint base;
      for (;;) {  // TEProjective::rand: x = Fq::rand (raw Montgomery limbs, top 7 bits masked, rejection), greatest = rng.gen::<bool>()
        uint64_t raw[6];
        for (;;) {
          for (int i = 0; i < 6; i++) raw[i] = rng.next_u64();
          raw[5] &= (1ULL << 57) - 1;
          if (cmp_n(raw, P377::P64, 6) < 0) break;
        }
        bool greatest = (rng.next_u32() >> 31) != 0;
        SF x = SF::from(Fq_::from_ark(raw));
        SF x2 = x * x;
        SF num = x2.neg() - one, den = dcoef * x2 - one;      // (a x^2 - 1) / (d x^2 - 1), a = -1
        if (den.v.is_zero_mod_p()) continue;
        SF y2 = num * SF::from(Fq_::inv(den.v));
        Fq_ yv;
        if (!fq_sqrt(y2.v, yv)) continue;
        if (fq_lex_largest(yv) != greatest) yv = fq_neg(yv);   // (y < -y) ^ greatest ? y : -y
        SF y = SF::from(yv);
        base = {x, y, one, x * y};
        for (int k = 0; k < 3; k++) base = ed_dbl(base);       // scale_by_cofactor (8)
        break;
      }
      for (int j = 0; j < WINDOW_SIZE; j++) {
        const EdPoint g2 = ed_dbl(base), g4 = ed_dbl(g2);
        gens.push_back(base); gens.push_back(g2); gens.push_back(ed_add(g2, base)); gens.push_back(g4);   // (1 + b0 + 2 b1) g for the chunk's two low bits
        base = ed_dbl(ed_dbl(g4));                                                                          // 16 g
      }
    }
  }
};
const CompositeParams& composite_params() { static CompositeParams p; return p; }
// The HOST evaluation of ONE message's CRH (round 6; VERDICT r5 item 5: the single-caller path - verify_signature, hash_composite*, hash_crh -
// spent more than half of a call here).  pedersen.h is one host+device source on the device's 28-bit limbs: right for the bulk GPU kernel, ~1.4 us
// per Edwards addition on a host core.  Here the same sum runs on 64-bit limbs (host64.h: arkworks' own Montgomery form, 6 x 6 CIOS) against a
// table of AFFINE entries precomputed for the mixed addition - (y - x, y + x, 2 d x y) per multiple, so an addition is 7 products (madd-2008-hwcd-3)
// and a negated entry is a swap and one negation: ~0.3 us per chunk.  Same group element, so the same 48 bytes (the reference's CRH vector and
// its 20 compat hash-to-G1 points pin both paths: tests/test_seam_a.py, tests/test_hash_gpu.py).  The table is derived from CompositeParams::gens
// (one batched inversion) window by window on first use: messages of the FFI's shapes touch the first two or three of the 560 windows.
struct CompositeHostTable {
  typedef HFp<P377> H;
  struct Entry { H ymx, ypx, t2d; };
  static constexpr size_t PER_WINDOW = (size_t)PEDERSEN_WINDOW_SIZE * PEDERSEN_MULTIPLES;
  std::vector<Entry> tab;
  std::vector<std::atomic<int>> ready;       // per window: 0 = not built, 2 = built (1 = being built: the builder holds `mu`)
  std::mutex mu;
  H two_d, raw_one;
  CompositeHostTable() : tab(PER_WINDOW * PEDERSEN_NUM_WINDOWS), ready(PEDERSEN_NUM_WINDOWS) {
    for (auto& r : ready) r.store(0);
    uint64_t w[6];
    sf_small(2 * 79743).v.to_ark(w);
    two_d = H::load(w);
    memset(raw_one.v, 0, sizeof raw_one.v);
    raw_one.v[0] = 1;                           // a * raw_one = a R^-1: Montgomery form -> the plain integer
  }
  static H from_sf(const SF& a) { uint64_t w[6]; a.v.to_ark(w); return H::load(w); }
  static H inv(const H& a) {                   // a^(p - 2), square-and-multiply from the top: one per built window and one per message
    uint64_t e[6];
    memcpy(e, P377::P64, sizeof e);
    e[0] -= 2;                                  // (p is odd and its low limb is far from 0: no borrow)
    H r = H::one();
    bool started = false;
    for (int i = 6 * 64 - 1; i >= 0; i--) {
      if (started) r = r.sqr();
      if ((e[i >> 6] >> (i & 63)) & 1) { r = started ? r * a : a; started = true; }
    }
    return r;
  }
  void build(int w) {
    std::lock_guard<std::mutex> lk(mu);
    if (ready[(size_t)w].load(std::memory_order_acquire) == 2) return;
    const EdPoint* g = composite_params().gens.data() + (size_t)w * PER_WINDOW;
    std::vector<H> z(PER_WINDOW), pre(PER_WINDOW);
    H acc = H::one();
    for (size_t i = 0; i < PER_WINDOW; i++) { z[i] = from_sf(g[i].Z); pre[i] = acc; acc = acc * z[i]; }
    H iv = inv(acc);
    for (size_t i = PER_WINDOW; i-- > 0;) {
      const H zi = iv * pre[i];
      iv = iv * z[i];
      const H x = from_sf(g[i].X) * zi, y = from_sf(g[i].Y) * zi;
      tab[(size_t)w * PER_WINDOW + i] = {y - x, y + x, x * y * two_d};
    }
    ready[(size_t)w].store(2, std::memory_order_release);
  }
  void eval(const uint8_t* msg, size_t len, uint8_t out48[48]) {
    const size_t nbits = len * 8, nchunks = (nbits + 2) / 3;
    for (size_t w = 0; w * PEDERSEN_WINDOW_SIZE < nchunks; w++)
      if (ready[w].load(std::memory_order_acquire) != 2) build((int)w);
    H X = H::zero(), Y = H::one(), Z = H::one(), T = H::zero();
    for (size_t ch = 0; ch < nchunks; ch++) {
      const size_t b = 3 * ch;
      uint32_t two = msg[b >> 3];
      if ((b >> 3) + 1 < len) two |= (uint32_t)msg[(b >> 3) + 1] << 8;
      const uint32_t bits = (two >> (b & 7)) & 7u;
      const Entry& e = tab[(size_t)PEDERSEN_MULTIPLES * ch + (bits & 3u)];
      const bool ng = (bits & 4) != 0;           // -(x, y) = (-x, y): the two sums swap, the T term changes sign
      const H A = (Y - X) * (ng ? e.ypx : e.ymx), B = (Y + X) * (ng ? e.ymx : e.ypx), Ct = T * e.t2d, D = Z.dbl();
      const H C = ng ? H::zero() - Ct : Ct;
      const H E = B - A, F = D - C, G = D + C, Hh = B + A;
      X = E * F; Y = G * Hh; Z = F * G; T = E * Hh;
    }
    // x = X / Z as a plain integer: Z out of Montgomery form, inverted by division steps (modinv.h works on plain 64-bit limbs: a few us on a
    // host core, where Fermat's 570 products were half of a 64-byte message's time), and X_mont * (1 / Z)_plain = (X / Z)_plain
    const H zc = Z * raw_one;
    H zi;
    SafeGcd<P377>::inv(zc.v, zi.v);
    const H x = X * zi;
    for (int i = 0; i < 48; i++) out48[i] = (uint8_t)(x.v[i >> 3] >> (8 * (i & 7)));
  }
  static CompositeHostTable& get() { static CompositeHostTable* t = new CompositeHostTable; return *t; }
};
bool composite_crh(const uint8_t* msg, size_t len, std::vector<uint8_t>& out) {  // bowe_hopwood::CRH::evaluate -> affine x, 48 bytes
  if (len * 8 > PEDERSEN_MAX_BITS) return false;  // the reference panics
  out.assign(48, 0);
  static const bool slow = getenv("CELO_CRH_DEVICE_LIMBS") != nullptr;      // A/B and cross-check switch: pedersen.h's source on the host, as rounds 2-5 ran it
  if (slow) pedersen_crh(composite_params().gens.data(), msg, len, out.data());
  else CompositeHostTable::get().eval(msg, len, out.data());
  return true;
}
// scale_by_cofactor + affine normalisation of hash_direct.h's tai_finish on the host's 64-bit limbs (host64.h): 124 doublings and 17 additions
// are ~1 700 products - 0.3 ms of every single hash on the device's 28-bit limbs, a third of that here.  Same group element.
static bool tai_finish_host64(const Affine<Fq_>& p, Affine<Fq_>& out) {
  typedef HFp<P377> H;
  const uint64_t cof[2] = {0x0000000000000000ULL, 0x170b5d4430000000ULL};   // (x - 1)^2 / 3, 125 bits, bit 124 set
  uint64_t w[6];
  HXyzz<H> base;
  p.x.to_ark(w); base.X = H::load(w);
  p.y.to_ark(w); base.Y = H::load(w);
  base.ZZ = H::one(); base.ZZZ = H::one();
  HXyzz<H> s = base;
  for (int i = 123; i >= 0; i--) {
    s = hxyzz_dbl(s);
    if ((cof[i >> 6] >> (i & 63)) & 1) hxyzz_add(s, base);
  }
  if (s.is_identity()) return false;
  H raw_one;
  memset(raw_one.v, 0, sizeof raw_one.v);
  raw_one.v[0] = 1;
  const H zc = (s.ZZ * s.ZZZ) * raw_one;     // plain integer
  H t;
  SafeGcd<P377>::inv(zc.v, t.v);             // 1 / (ZZ ZZZ) as a plain integer: (a_mont * b_plain) is a b plain, times c_mont is a b c plain
  const H x = (s.X * t) * s.ZZZ, y = (s.Y * t) * s.ZZ;
  out.x = Fq_::norm(Fq_::from_canonical(x.v));
  out.y = Fq_::norm(Fq_::from_canonical(y.v));
  return true;
}

// generic try-and-increment over {direct, composite} x {plain, cip22} with the `compat` bit logic
// pre_cofactor (optional): the curve point BEFORE scale_by_cofactor (from_random_bytes' point), for the callers that restate arkworks' own multiple
bool hash_to_g1(bool composite, bool cip22, const uint8_t* dom, const uint8_t* msg, size_t mlen, const uint8_t* extra, size_t elen,
                Affine<Fq_>& out, int& attempt, Affine<Fq_>* pre_cofactor = nullptr) {
  auto crh = [&](const uint8_t* m, size_t l, std::vector<uint8_t>& o) -> bool {
    if (composite) return composite_crh(m, l, o);
    o = direct_crh(dom, 8, m, l, 64);
    return true;
  };
  std::vector<uint8_t> inner, buf, pre;
  if (cip22 && !crh(msg, mlen, inner)) return false;
  const uint8_t* tail = cip22 ? inner.data() : msg;
  const size_t tlen = cip22 ? inner.size() : mlen;
  buf.resize(1 + elen + tlen);
  if (elen) memcpy(buf.data() + 1, extra, elen);
  if (tlen) memcpy(buf.data() + 1 + elen, tail, tlen);
  for (int c = 0; c < 255; c++) {
    buf[0] = (uint8_t)c;
    std::vector<uint8_t> cand;
    if (cip22) cand = direct_xof(dom, 8, buf.data(), buf.size(), 64);
    else { if (!crh(buf.data(), buf.size(), pre)) return false; cand = direct_xof(dom, 8, pre.data(), pre.size(), 64); }
    uint32_t w12[12];
    memcpy(w12, cand.data(), 48);
    Affine<Fq_> p = {Fq_::zero(), Fq_::zero()};
    if (!tai_point_from_xof(w12, wire_consts(), p)) continue;            // hash_direct.h: compat flags, get_point_from_x
    if (!tai_finish_host64(p, out)) continue;                            // scale_by_cofactor
    attempt = c;
    if (pre_cofactor) *pre_cofactor = p;
    return true;
  }
  return false;
}

// hash many messages on all host cores (hash-to-curve is host plumbing for now; one attempt costs a
// Blake2Xs call, a 377-bit square root and a 125-bit cofactor multiplication)
struct HashJob { const uint8_t* msg; size_t mlen; const uint8_t* extra; size_t elen; uint64_t* out_xy; };
// failed (optional, one flag per job): a message that cannot be hashed marks its own job and the call goes on (the caller
// decides what one bad message means); without it the first failure fails the call.
bool hash_many(bool composite, bool cip22, const uint8_t* dom, std::vector<HashJob>& jobs, std::vector<uint8_t>* failed = nullptr) {
  (void)wire_consts();
  if (failed) failed->assign(jobs.size(), 0);
  bool gpu_done = false;
  // many messages: hashing runs on the GPU (a lone wave of 64 needs ~4 ms, so the host cores keep the small calls), for
  // every hasher: hash_direct.h's try-and-increment rounds over the Blake2s CRH, over precomputed Pedersen CRHs (CIP22), or
  // with a Pedersen CRH per attempt (composite before CIP22)
  if (jobs.size() >= 256) {
    const size_t n = jobs.size();
    std::vector<uint64_t> moff(n + 1, 0), eoff(n + 1, 0);
    for (size_t i = 0; i < n; i++) { moff[i + 1] = moff[i] + jobs[i].mlen; eoff[i + 1] = eoff[i] + jobs[i].elen; }
    std::vector<uint8_t> mb(moff[n] + 1), eb(eoff[n] + 1), att(n);
    for (size_t i = 0; i < n; i++) {
      if (jobs[i].mlen) memcpy(&mb[moff[i]], jobs[i].msg, jobs[i].mlen);
      if (jobs[i].elen) memcpy(&eb[eoff[i]], jobs[i].extra, jobs[i].elen);
    }
    std::vector<uint64_t> xy(n * 12);
    const int rc = composite ? hash_to_g1_composite_bls12_377(dom, mb.data(), moff.data(), eb.data(), eoff.data(), n, cip22 ? 1 : 0, xy.data(), att.data())
                             : hash_to_g1_direct_bls12_377(dom, mb.data(), moff.data(), eb.data(), eoff.data(), n, xy.data(), att.data());
    if (rc != 0 && !failed) return false;
    if (rc == 0) {
      for (size_t i = 0; i < n; i++) {
        if (att[i] == 255) { if (!failed) return false; (*failed)[i] = 1; continue; }
        memcpy(jobs[i].out_xy, &xy[i * 12], 96);
      }
      gpu_done = true;
    }
    // rc != 0 with per-job flags wanted (e.g. ONE oversized composite message): the host loop below sorts out which
  }
  if (gpu_done) return true;
  if (composite) (void)composite_params();  // initialise shared constants before the threads start
  std::atomic<size_t> next(0);
  std::atomic<bool> ok(true);
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 64) nt = 64;
  if (nt > jobs.size() / 4 + 1) nt = (unsigned)(jobs.size() / 4 + 1);    // at least ~4 hashes (~1 ms) per thread
  auto work = [&]() {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= jobs.size()) break;
      Affine<Fq_> h; int c;
      if (!hash_to_g1(composite, cip22, dom, jobs[i].msg, jobs[i].mlen, jobs[i].extra, jobs[i].elen, h, c)) {
        if (failed) (*failed)[i] = 1; else ok = false;
        continue;
      }
      h.x.to_ark(jobs[i].out_xy); h.y.to_ark(jobs[i].out_xy + 6);
    }
  };
  if (nt <= 1) work();
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work);
    for (auto& t : th) t.join();
  }
  return ok;
}
// Jacobian (ark limbs, stride 3*A u64) -> affine xy (ark limbs); inf[i] = 1 for the identity.  Handles that came from the wire
// (deserialize_*) carry Z = 1 and are copied without arithmetic; the rest share one inversion per chunk (Montgomery's
// trick); large inputs are cut into chunks across the host cores (at BASELINE config 3's scale - 10^6 keys and signatures
// per call - a serial pass here would cost 30x the GPU work it feeds).
template <class F> void batch_to_affine_range(const uint64_t* jac, size_t n, uint64_t* xy, uint8_t* inf) {
  constexpr int A = F::ARK64;
  uint64_t one_ark[A];
  F::one().to_ark(one_ark);
  std::vector<uint32_t> todo;
  for (size_t i = 0; i < n; i++) {
    const uint64_t* zp = jac + i * 3 * A + 2 * A;
    if (memcmp(zp, one_ark, A * 8) == 0) { memcpy(xy + i * 2 * A, jac + i * 3 * A, 2 * A * 8); inf[i] = 0; continue; }
    bool zero = true;
    for (int k = 0; k < A; k++) zero = zero && zp[k] == 0;
    if (zero) { memset(xy + i * 2 * A, 0, 2 * A * 8); inf[i] = 1; continue; }
    todo.push_back((uint32_t)i);
  }
  if (todo.empty()) return;
  std::vector<F> z(todo.size()), pre(todo.size());
  F acc = F::one();
  for (size_t t = 0; t < todo.size(); t++) {
    z[t] = F::norm(F::from_ark(jac + (size_t)todo[t] * 3 * A + 2 * A));
    inf[todo[t]] = z[t].is_zero_mod_p() ? 1 : 0;     // a non-canonical zero cannot come from this library; handled anyway
    pre[t] = acc;
    if (!inf[todo[t]]) acc = F::mul(acc, z[t]);
  }
  F ai = F::inv(acc);
  for (size_t t = todo.size(); t-- > 0;) {
    const size_t i = todo[t];
    uint64_t* o = xy + i * 2 * A;
    if (inf[i]) { memset(o, 0, 2 * A * 8); continue; }
    F zi = F::mul(ai, pre[t]);
    ai = F::mul(ai, z[t]);
    F zi2 = F::sqr(zi);
    F::mul(F::from_ark(jac + i * 3 * A), zi2).to_ark(o);
    F::mul(F::from_ark(jac + i * 3 * A + A), F::mul(zi2, zi)).to_ark(o + A);
  }
}
template <class F> void batch_to_affine(const uint64_t* jac, size_t n, uint64_t* xy, uint8_t* inf) {
  constexpr int A = F::ARK64;
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 64) nt = 64;
  if (n < 4096 || nt < 2) { batch_to_affine_range<F>(jac, n, xy, inf); return; }
  const size_t chunk = (n + nt - 1) / nt;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) {
    const size_t lo = (size_t)t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    if (lo >= hi) break;
    th.emplace_back([=]() { batch_to_affine_range<F>(jac + lo * 3 * A, hi - lo, xy + lo * 2 * A, inf + lo); });
  }
  for (auto& x : th) x.join();
}

// ---------------------------------------------------------------- BW6-761 wire format (Groth16 VerifyingKey / Proof points)
typedef Fp<P761> Fw_;
bool fw_from_bytes(const uint8_t* in, Fw_& out) {
  uint64_t w[12];
  memcpy(w, in, 96);
  if (cmp_n(w, P761::P64, 12) >= 0) return false;
  out = Fw_::from_canonical(w);
  return true;
}
bool fw_eq(const Fw_& a, const Fw_& b) { return Fw_::eq_mod_p(Fw_::norm(a), Fw_::norm(b)); }
bool fw_lex_largest(const Fw_& a) {
  uint64_t w[12];
  a.to_canonical(w);
  return cmp_n(w, P761::PM1_HALF64, 12) > 0;
}
Fw_ fw_neg(const Fw_& a) { return Fw_::wred(Fw_::norm(Fw_::neg<64, 1>(Fw_::norm(a)))); }
bool fw_sqrt(const Fw_& a, Fw_& out) {  // q = 3 (mod 4): a^((q+1)/4)
  uint64_t e[12];
  memcpy(e, P761::P64, 96);
  e[0] += 1;  // no carry: low limb of q is ...8b
  for (int i = 0; i < 12; i++) e[i] = (e[i] >> 2) | (i + 1 < 12 ? e[i + 1] << 62 : 0);
  Fw_ y = Fw_::pow64(Fw_::norm(a), e, 12);
  if (!fw_eq(Fw_::sqr(y), a)) return false;
  out = y;
  return true;
}
// b = -1 for G1 (y^2 = x^3 - 1), +4 for G2 (M-twist, coordinates in Fq)
bool bw6_decompress(const uint8_t* in, bool g2, Affine<Fw_>& p, bool& inf) {
  uint8_t buf[96];
  memcpy(buf, in, 96);
  uint8_t flags = buf[95] & 0xC0;
  buf[95] &= 0x3F;
  inf = (flags & 0x40) != 0;
  if (inf) return true;
  Fw_ x;
  if (!fw_from_bytes(buf, x)) return false;
  Fw_ x3 = Fw_::mul(Fw_::sqr(x), x);
  Fw_ rhs;
  if (g2) { Fw_ one = Fw_::one(); rhs = Fw_::norm(Fw_::add(x3, Fw_::norm(Fw_::dbl(Fw_::dbl(one))))); }
  else rhs = Fw_::norm(Fw_::sub<4, 1>(x3, Fw_::one()));
  Fw_ y;
  if (!fw_sqrt(rhs, y)) return false;
  if (fw_lex_largest(y) != ((flags & 0x80) != 0)) y = fw_neg(y);
  p = {Fw_::norm(x), Fw_::norm(y)};
  // arkworks' GroupAffine::deserialize checks the prime-order subgroup (r_BW6 = q_BLS12-377)
  Xyzz<Fw_> r = scalar_mul_host(p, P377::P64, 6);
  return r.is_identity() || r.ZZ.is_zero_mod_p();
}
void bw6_store_xy(const Affine<Fw_>& p, uint64_t* out) { p.x.to_ark(out); p.y.to_ark(out + 12); }

// ---------------------------------------------------------------- epoch encoding (crates/epoch-snark/src/{encoding,epoch_block}.rs,
// crates/bls-gadgets/src/utils.rs:2-56, crates/epoch-snark/src/gadgets/mod.rs:75-83) — byte/bit plumbing
typedef std::vector<uint8_t> Bits;
void bits_append_le(Bits& b, const uint8_t* bytes, size_t nbytes, size_t take) {  // bytes_le_to_bits_le
  for (size_t i = 0; i < take; i++) b.push_back(i / 8 < nbytes ? (bytes[i / 8] >> (i % 8)) & 1 : 0);
}
void bits_append_be(Bits& b, const uint8_t* bytes, size_t nbytes, size_t take) {  // bytes_le_to_bits_be: first `take` LE bits, reversed
  for (size_t i = take; i-- > 0;) b.push_back(i / 8 < nbytes ? (bytes[i / 8] >> (i % 8)) & 1 : 0);
}
std::vector<uint8_t> bits_be_to_bytes_le(const Bits& bits) {
  std::vector<uint8_t> out;
  size_t n = bits.size();
  for (size_t i = 0; i < n; i += 8) {
    uint8_t byte = 0;
    for (size_t k = 0; k < 8 && i + k < n; k++) byte |= (uint8_t)(bits[n - 1 - (i + k)] << k);
    out.push_back(byte);
  }
  return out;
}
void encode_uint(Bits& b, uint64_t v, size_t nbytes) {
  uint8_t le[8];
  for (size_t i = 0; i < 8; i++) le[i] = (uint8_t)(v >> (8 * i));
  bits_append_le(b, le, nbytes, 8 * nbytes);
}
void encode_public_key_bits(Bits& b, const Affine<Fq2_>& pk) {  // encoding.rs:23-47
  uint8_t x0[48], x1[48];
  fq_to_bytes(pk.x.c0, x0);
  fq_to_bytes(pk.x.c1, x1);
  bits_append_be(b, x0, 48, 377);
  bits_append_be(b, x1, 48, 377);
  bool over_half = fq_lex_largest(pk.y.c1) || (pk.y.c1.is_zero_mod_p() && fq_lex_largest(pk.y.c0));
  b.push_back(over_half ? 1 : 0);
}
void encode_entropy_bits(Bits& b, const uint8_t* entropy) {  // epoch_block.rs:140-148 (None -> zero bits)
  uint8_t zero[16];
  memset(zero, 0, 16);
  bits_append_le(b, entropy ? entropy : zero, 16, 128);
}
struct EpochBlockHost {
  uint16_t index; uint8_t round; const uint8_t* epoch_entropy; const uint8_t* parent_entropy;
  uint32_t maximum_non_signers; size_t maximum_validators; std::vector<Affine<Fq2_>> pubkeys; std::vector<uint64_t> pubkeys_jac;
};
Affine<Fq2_> g2_generator_affine() {
  return {{Fq_::from_limbs(T377::G2_GEN_X0), Fq_::from_limbs(T377::G2_GEN_X1)}, {Fq_::from_limbs(T377::G2_GEN_Y0), Fq_::from_limbs(T377::G2_GEN_Y1)}};
}
void epoch_bits_cip22(const EpochBlockHost& e, bool first, Bits& b) {  // epoch_block.rs:118-138
  encode_uint(b, e.index, 2);
  encode_entropy_bits(b, first ? e.parent_entropy : e.epoch_entropy);
  encode_uint(b, e.maximum_non_signers, 4);
  for (const auto& pk : e.pubkeys) encode_public_key_bits(b, pk);
  for (size_t i = e.pubkeys.size(); i < e.maximum_validators; i++) encode_public_key_bits(b, g2_generator_affine());
}
std::vector<uint8_t> blake2s_out_domain(const std::vector<uint8_t>& data) {  // epoch_block.rs:226-236, OUT_DOMAIN = "ULforout"
  static const uint8_t OUT_DOMAIN[8] = {'U', 'L', 'f', 'o', 'r', 'o', 'u', 't'};
  B2sParams p;
  return blake2s(data.data(), data.size(), p, OUT_DOMAIN, 8);
}
bool epoch_from_ffi(const EpochBlockFFI& src, EpochBlockHost& e) {  // snark/epoch_block.rs:129-146
  e.index = src.index; e.round = src.round; e.epoch_entropy = src.epoch_entropy; e.parent_entropy = src.parent_entropy;
  e.maximum_non_signers = src.maximum_non_signers; e.maximum_validators = src.maximum_validators;
  e.pubkeys.resize(src.pubkeys_num);
  e.pubkeys_jac.resize(src.pubkeys_num * 36);
  // one square root in Fq2 + one subgroup check per key (~0.5 ms each): spread over the host cores for real validator sets
  std::atomic<bool> ok(true);
  auto decode = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi && ok; i++) {
      bool inf;
      if (!g2_decompress(src.pubkeys + 96 * i, e.pubkeys[i], inf)) { ok = false; return; }
      if (inf) {      // read_pubkeys (snark/epoch_block.rs:187-196) takes G2Affine::deserialize's zero() as it comes: (0, 1), the identity in the sum
        e.pubkeys[i] = {Fq2_::zero(), Fq2_::one()};
        identity_jac<Fq2_>(&e.pubkeys_jac[i * 36]);
        continue;
      }
      if (!in_subgroup(e.pubkeys[i])) { ok = false; return; }
      affine_to_jac(e.pubkeys[i], &e.pubkeys_jac[i * 36]);
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 32) nt = 32;
  if (nt > src.pubkeys_num / 2) nt = (unsigned)(src.pubkeys_num / 2);
  if (nt < 2) decode(0, src.pubkeys_num);
  else {
    (void)wire_consts();   // shared constants before the threads start
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(decode, src.pubkeys_num * t / nt, src.pubkeys_num * (t + 1) / nt);
    for (auto& x : th) x.join();
  }
  return ok;
}
void neg_g2_generator(uint64_t out_xy[24]) {
  Fq_::from_limbs(T377::G2_GEN_X0).to_ark(out_xy);
  Fq_::from_limbs(T377::G2_GEN_X1).to_ark(out_xy + 6);
  fq_neg(Fq_::from_limbs(T377::G2_GEN_Y0)).to_ark(out_xy + 12);
  fq_neg(Fq_::from_limbs(T377::G2_GEN_Y1)).to_ark(out_xy + 18);
}
}  // namespace

namespace celo {
struct BvJob { BatchRun keys, sigs; };      // unit_batchverify.hip: the chained Batch::verify in three steps
int bv_begin_keys(BvJob*, const void*, const void*, const void*, int, const uint32_t*, size_t);
int bv_begin_sigs(BvJob*, const void*, const void*, const void*, int, const uint32_t*, size_t);
int bv_finish(BvJob*, int, const void*, const void*, int, const uint64_t*, size_t, uint8_t*);
int bv_mirror_scatter(int, const uint64_t*, const uint8_t*, const uint32_t*, uint64_t*, uint8_t*, size_t, hipStream_t);
int bv_mirror_gather(int, const uint64_t*, const uint8_t*, const uint32_t*, uint64_t*, uint8_t*, size_t, hipStream_t);
int bv_draw_exponents(const uint32_t*, const uint32_t*, size_t, size_t, uint64_t*, hipStream_t);
// the composite hasher's generator table for the bulk GPU kernel (unit_hash.hip: k_pedersen_crh)
const EdPoint* celo_composite_gens(size_t* count) {
  const CompositeParams& cp = composite_params();
  *count = cp.gens.size();
  return cp.gens.data();
}
}  // namespace celo

extern "C" {

bool init(void) {  // lib.rs:28-36: force both lazy hashers (the Bowe-Hopwood generator table) and bring the device up
  (void)composite_params();
  (void)wire_consts();
  return celo_amd_init(0) == 0;
}

// ---------------------------------------------------------------- keys (crates/bls-snark-sys/src/signatures.rs:19-42)
bool generate_private_key(PrivateKey** out_private_key) {
  if (!out_private_key) return false;
  ChaCha20Rng rng;
  if (!os_seeded_rng(rng)) return false;
  PrivateKey* sk = new PrivateKey;
  for (;;) {
    for (int i = 0; i < 4; i++) sk->k[i] = rng.next_u64();
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
  PublicKey* pk = new_public_key();
  if (!pk) return false;
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
  PublicKey* pk = new_public_key();
  if (!pk) return false;
  if (inf) identity_jac<Fq2_>(pk->xyz);
  else {
    if (!in_subgroup(p)) { drop(pk); log_err("deserialize_public_key: point not in the prime-order subgroup"); return false; }
    affine_to_jac(p, pk->xyz);
  }
  *out = pk;
  return true;
}
// The reference memoises decompression in a 512-entry LRU keyed by the serialized bytes (serialization.rs:44-61,
// crates/bls-crypto/src/bls/cache.rs:36,49-65): validator keys recur epoch after epoch, and a hit replaces a square root and a
// subgroup check (~0.5 ms) by a 288-byte copy.  Decoding is a pure function, so the cache is not observable through the ABI.
bool deserialize_public_key_cached(const uint8_t* in_bytes, int in_len, PublicKey** out) {
  if (!in_bytes || in_len != 96 || !out) return deserialize_public_key(in_bytes, in_len, out);
  static std::mutex mu;
  struct Limbs { uint64_t xyz[36]; };
  static std::list<std::pair<std::string, Limbs>> lru;                                       // front = most recent
  static std::unordered_map<std::string, std::list<std::pair<std::string, Limbs>>::iterator> index;
  const std::string key((const char*)in_bytes, 96);
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = index.find(key);
    if (it != index.end()) {
      lru.splice(lru.begin(), lru, it->second);
      PublicKey* pk = new_public_key();
      if (!pk) return false;
      memcpy(pk->xyz, it->second->second.xyz, 288);
      *out = pk;
      return true;
    }
  }
  if (!deserialize_public_key(in_bytes, in_len, out)) return false;
  std::lock_guard<std::mutex> lk(mu);
  if (index.find(key) == index.end()) {
    Limbs v;
    memcpy(v.xyz, (*out)->xyz, 288);
    lru.emplace_front(key, v);
    index[key] = lru.begin();
    if (lru.size() > 512) { index.erase(lru.back().first); lru.pop_back(); }
  }
  return true;
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
  Signature* s = new_signature();
  if (!s) return false;
  if (inf) identity_jac<Fq_>(s->xyz);
  else {
    if (!in_subgroup(p)) { drop(s); log_err("deserialize_signature: point not in the prime-order subgroup"); return false; }
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
bool destroy_public_key(PublicKey* p) { return p && drop(p); }   // false for a handle destroyed before (its slot stays out of the free list)
bool destroy_signature(Signature* p) { return p && drop(p); }
bool free_vec(uint8_t* bytes, int len) { if (!bytes || len < 0) return false; free(bytes); return true; }   // buffers are malloc blocks: the length is not needed to release one

// ---------------------------------------------------------------- aggregation (signatures.rs:428-505)
// aggregate_public_keys and aggregate_public_keys_subtract route their list through PublicKeyCache::aggregate
// (crates/bls-crypto/src/bls/cache.rs:65-87), which collects the keys into a HashSet with BYTE-LEVEL equality of the Jacobian
// (x, y, z) limbs (cache.rs:95-104): a handle listed twice - or two handles holding the same limbs - counts once, whereas two
// different Jacobian representatives of one point count twice.  The cache's incremental update (subtract the keys that left,
// add the new ones) is an optimisation of "sum of the set"; only the set semantics is observable.  aggregate_signatures is a
// plain sum (Signature::aggregate, signature.rs:61-67).
static bool unique_key_limbs(const PublicKey* const* in, int n, std::vector<uint64_t>& buf, size_t first) {
  struct Ref { const uint64_t* p; };
  struct H { size_t operator()(const Ref& r) const { uint64_t h = 0xcbf29ce484222325ull; for (int i = 12; i < 24; i++) h = (h ^ r.p[i]) * 0x100000001b3ull; return (size_t)h; } };
  struct E { bool operator()(const Ref& a, const Ref& b) const { return memcmp(a.p, b.p, 288) == 0; } };
  std::unordered_set<Ref, H, E> seen;
  seen.reserve((size_t)n * 2 + 1);
  buf.resize(first * 36);
  for (int i = 0; i < n; i++) {
    if (!in[i]) return false;
    if (!seen.insert(Ref{in[i]->xyz}).second) continue;
    buf.insert(buf.end(), in[i]->xyz, in[i]->xyz + 36);
  }
  return true;
}
bool aggregate_public_keys(const PublicKey* const* in, int n, PublicKey** out) {
  if (!out || n < 0 || (n > 0 && !in)) return false;
  std::vector<uint64_t> buf;
  if (!unique_key_limbs(in, n, buf, 0)) return false;
  PublicKey* pk = new_public_key();
  if (!pk) return false;
  if (celo_amd_sum_jacobian_bls12_377_g2(buf.data(), buf.size() / 36, pk->xyz) != 0) { drop(pk); return false; }
  *out = pk;
  return true;
}
bool aggregate_public_keys_subtract(const PublicKey* agg, const PublicKey* const* in, int n, PublicKey** out) {
  if (!agg || !out || n < 0 || (n > 0 && !in)) return false;
  std::vector<uint64_t> buf;
  if (!unique_key_limbs(in, n, buf, 1)) return false;
  memcpy(buf.data(), agg->xyz, 288);
  for (size_t i = 1; i < buf.size() / 36; i++) {
    uint64_t* d = &buf[i * 36];
    Fq2_ y = Fq2_::from_ark(d + 12);                       // negate: (X, -Y, Z)
    Fq2_ ny = {fq_neg(y.c0), fq_neg(y.c1)};
    ny.to_ark(d + 12);
  }
  PublicKey* pk = new_public_key();
  if (!pk) return false;
  if (celo_amd_sum_jacobian_bls12_377_g2(buf.data(), buf.size() / 36, pk->xyz) != 0) { drop(pk); return false; }
  *out = pk;
  return true;
}
bool aggregate_signatures(const Signature* const* in, int n, Signature** out) {
  if (!out || n < 0 || (n > 0 && !in)) return false;
  std::vector<uint64_t> buf((size_t)n * 18);
  for (int i = 0; i < n; i++) { if (!in[i]) return false; memcpy(&buf[(size_t)i * 18], in[i]->xyz, 144); }
  Signature* s = new_signature();
  if (!s) return false;
  if (celo_amd_sum_jacobian_bls12_377_g1(buf.data(), (size_t)n, s->xyz) != 0) { drop(s); return false; }
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
struct PhaseLog {  // CELO_AMD_LOG=1: wall time of the host / device phases of one FFI call
  const char* fn; bool on; std::chrono::steady_clock::time_point t;
  explicit PhaseLog(const char* f) : fn(f), on(getenv("CELO_AMD_LOG") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[celo-amd] %s: %-28s %9.3f ms\n", fn, what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
bool celo_amd_verify_hash(const PublicKey* pk, const uint64_t* message_hash_xy, const Signature* sig, bool* out_verified) {
  if (!pk || !message_hash_xy || !sig || !out_verified) return false;
  PhaseLog ph("verify_hash");
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
  ph.mark("handles -> affine pairs (host)");
  int one = 0;
  if (pairing_product_is_one_bls12_377(g1, inf1, g2, inf2, 2, &one) != 0) return false;
  ph.mark("2-pair product check (GPU, latency path)");
  *out_verified = one != 0;
  return true;
}

// ---------------------------------------------------------------- hashing / signing with the DIRECT hasher
// (composite = Bowe-Hopwood over Edwards-BW6-761 is not built yet: those flag combinations return false, logged)
static bool hash_flags_supported(bool composite, bool cip22) {
  if (!composite && cip22) { log_err("(composite=false, cip22=true) is rejected by the reference too (signatures.rs:61)"); return false; }
  return true;
}
static bool emit_affine_tobytes(const Affine<Fq_>& p, uint8_t** out, int* out_len) {  // ToBytes: x || y || infinity byte
  std::vector<uint8_t> v(97, 0);
  fq_to_bytes(p.x, v.data());
  fq_to_bytes(p.y, v.data() + 48);
  return emit(v, out, out_len);
}
bool hash_direct(const uint8_t* msg, int len, uint8_t** out_hash, int* out_len, bool use_pop) {                /* signatures.rs:93 */
  if ((!msg && len) || !out_hash || !out_len || len < 0) return false;
  Affine<Fq_> h; int c;
  if (!hash_to_g1_direct(use_pop ? POP_DOMAIN : SIG_DOMAIN, msg, (size_t)len, nullptr, 0, h, c)) return false;
  return emit_affine_tobytes(h, out_hash, out_len);
}
bool hash_direct_with_attempt(const uint8_t* msg, int len, uint8_t** out_hash, int* out_len, int* out_attempt, bool use_pop) { /* :117 */
  if ((!msg && len) || !out_hash || !out_len || !out_attempt || len < 0) return false;
  Affine<Fq_> h; int c;
  if (!hash_to_g1_direct(use_pop ? POP_DOMAIN : SIG_DOMAIN, msg, (size_t)len, nullptr, 0, h, c)) return false;
  *out_attempt = c;
  return emit_affine_tobytes(h, out_hash, out_len);
}
bool hash_direct_first_step(const uint8_t* msg, int len, int hash_bytes, uint8_t** out_hash, int* out_len) {    /* signatures.rs:192 */
  if ((!msg && len) || !out_hash || !out_len || len < 0 || hash_bytes < 0 || hash_bytes > 65535) return false;
  return emit(direct_hash(SIG_DOMAIN, 8, msg, (size_t)len, (size_t)hash_bytes), out_hash, out_len);
}
// hash_composite / hash_composite_cip22 return ToBytes of a G1Projective (x || y || z, 144 bytes): the Jacobian representative that
// arkworks' scale_by_cofactor leaves (crates/bls-crypto/src/hash_to_curve/try_and_increment.rs:130, try_and_increment_cip22.rs:125 return
// `scaled` as it is, signatures.rs:143,215 write it).  Round 5 restates that schedule so that the BYTES agree and not only
// the point: GroupAffine::mul_bits over BitIteratorBE(COFACTOR) - res = zero; per bit, MSB first: res.double_in_place(); if bit:
// res.add_assign_mixed(p) - with ark-ec's a = 0 doubling (dbl-2009-l) and mixed addition (madd-2007-bl) on GroupProjective (SURVEY.md
// Appendix B.6; pinned revision arkworks-rs/algebra@8d76d181, source not on disk: restated from the published formulas).  The formulas are
// exact arithmetic mod q, so the representative is a function of (p, COFACTOR) alone; it is computed here on 64-bit Montgomery limbs
// (host64.h).  No reference vector pins these 144 bytes: tests check that into_affine() of them is the hash point the
// compat vectors pin, and the schedule against an independent big-integer restatement (oracle/py).
static void ark_scale_by_cofactor_tobytes(const Affine<Fq_>& p, uint8_t out[144]) {
  typedef HFp<P377> H;
  uint64_t w[6];
  p.x.to_ark(w); const H px = H::load(w);
  p.y.to_ark(w); const H py = H::load(w);
  H X = H::zero(), Y = H::one(), Z = H::zero();                          // GroupProjective::zero() = (0, 1, 0)
  const uint64_t cof[2] = {0x0000000000000000ULL, 0x170b5d4430000000ULL};  // G1 COFACTOR (ark-bls12-377 g1.rs), little-endian limbs
  auto dbl = [&]() {
    if (Z.is_zero()) return;
    const H a = X.sqr(), b = Y.sqr(), c = b.sqr();
    const H d = ((X + b).sqr() - a - c).dbl();
    const H e = a + a.dbl(), f = e.sqr();
    Z = (Z * Y).dbl();
    X = f - d - d;
    Y = (d - X) * e - c.dbl().dbl().dbl();
  };
  auto madd = [&]() {
    if (Z.is_zero()) { X = px; Y = py; Z = H::one(); return; }
    const H z1z1 = Z.sqr(), u2 = px * z1z1, s2 = (py * Z) * z1z1;
    if (memcmp(X.v, u2.v, sizeof X.v) == 0 && memcmp(Y.v, s2.v, sizeof Y.v) == 0) { dbl(); return; }
    const H h = u2 - X, hh = h.sqr(), i = hh.dbl().dbl();
    H j = h * i;
    const H r = (s2 - Y).dbl(), v = X * i;
    X = r.sqr() - j - v - v;
    j = (j * Y).dbl();
    Y = (v - X) * r - j;
    Z = (Z + h).sqr() - z1z1 - hh;
  };
  for (int i = 127; i >= 0; i--) {
    dbl();
    if ((cof[i >> 6] >> (i & 63)) & 1) madd();
  }
  uint64_t raw1[6] = {1, 0, 0, 0, 0, 0};
  const H unmont = H::load(raw1);                                        // x * (1 as a raw residue) = x R^-1: out of Montgomery form
  const H c[3] = {X * unmont, Y * unmont, Z * unmont};
  for (int k = 0; k < 3; k++) memcpy(out + 48 * k, c[k].v, 48);
}
static bool emit_projective_tobytes(const Affine<Fq_>& pre, uint8_t** out, int* out_len) {
  std::vector<uint8_t> v(144, 0);
  ark_scale_by_cofactor_tobytes(pre, v.data());
  return emit(v, out, out_len);
}
bool hash_composite(const uint8_t* msg, int mlen, const uint8_t* extra, int elen, uint8_t** out_hash, int* out_len) {   /* signatures.rs:143 */
  if ((!msg && mlen) || (!extra && elen) || !out_hash || !out_len || mlen < 0 || elen < 0) return false;
  Affine<Fq_> h, pre; int c;
  if (!hash_to_g1(true, false, SIG_DOMAIN, msg, (size_t)mlen, extra, (size_t)elen, h, c, &pre)) return false;
  return emit_projective_tobytes(pre, out_hash, out_len);
}
bool hash_composite_cip22(const uint8_t* msg, int mlen, const uint8_t* extra, int elen, uint8_t** out_hash, int* out_len,
                          uint8_t* attempt_counter) {                                                                   /* signatures.rs:215 */
  if ((!msg && mlen) || (!extra && elen) || !out_hash || !out_len || !attempt_counter || mlen < 0 || elen < 0) return false;
  Affine<Fq_> h, pre; int c;
  if (!hash_to_g1(true, true, SIG_DOMAIN, msg, (size_t)mlen, extra, (size_t)elen, h, c, &pre)) return false;
  *attempt_counter = (uint8_t)c;
  return emit_projective_tobytes(pre, out_hash, out_len);
}
bool hash_crh(const uint8_t* msg, int mlen, int hash_bytes, uint8_t** out_hash, int* out_len) {                         /* signatures.rs:169 */
  (void)hash_bytes;  // the Bowe-Hopwood CRH ignores the domain and the output length (composite.rs:79-86)
  if ((!msg && mlen) || !out_hash || !out_len || mlen < 0) return false;
  std::vector<uint8_t> o;
  if (!composite_crh(msg, (size_t)mlen, o)) return false;
  return emit(o, out_hash, out_len);
}
/* test hook: DirectHasher with an arbitrary domain (<= 8 bytes, may be empty).  what: 0 = crh (32 bytes), 1 = xof of `msg`,
   2 = hash = xof(domain, crh(domain, msg))  (hashers/direct.rs:20-78) */
bool celo_amd_direct_hasher(int what, const uint8_t* domain, int dlen, const uint8_t* msg, int mlen, int out_bytes, uint8_t* out) {
  if (!out || dlen < 0 || dlen > 8 || mlen < 0 || out_bytes < 0 || (!domain && dlen) || (!msg && mlen)) return false;
  std::vector<uint8_t> r = what == 0 ? direct_crh(domain, (size_t)dlen, msg, (size_t)mlen, (size_t)out_bytes)
                         : what == 1 ? direct_xof(domain, (size_t)dlen, msg, (size_t)mlen, (size_t)out_bytes)
                                     : direct_hash(domain, (size_t)dlen, msg, (size_t)mlen, (size_t)out_bytes);
  memcpy(out, r.data(), r.size());
  return true;
}
/* test hook: CompositeHasher::hash = xof(domain, crh(message), out_bytes) (hashers/mod.rs Hasher::hash) */
bool celo_amd_composite_hash(const uint8_t* domain8, const uint8_t* msg, int mlen, int out_bytes, uint8_t* out) {
  if (!domain8 || !out || mlen < 0 || out_bytes < 0) return false;
  std::vector<uint8_t> c;
  if (!composite_crh(msg, (size_t)mlen, c)) return false;
  std::vector<uint8_t> x = direct_xof(domain8, 8, c.data(), c.size(), (size_t)out_bytes);
  memcpy(out, x.data(), (size_t)out_bytes);
  return true;
}
/* test hook: hash-to-G1 with an arbitrary 8-byte domain (the reference's golden vectors use random domains), compressed 48-byte output */
bool celo_amd_hash_to_g1(bool composite, bool cip22, const uint8_t* domain8, const uint8_t* msg, int mlen, const uint8_t* extra, int elen,
                         uint8_t* out48, int* out_attempt) {
  if (!domain8 || !out48 || mlen < 0 || elen < 0 || (!composite && cip22)) return false;
  Affine<Fq_> h; int c;
  if (!hash_to_g1(composite, cip22, domain8, msg, (size_t)mlen, extra, (size_t)elen, h, c)) return false;
  g1_compress(h, false, out48);
  if (out_attempt) *out_attempt = c;
  return true;
}
static bool sign_with(const PrivateKey* sk, bool composite, bool cip22, const uint8_t* dom, const uint8_t* msg, int mlen, const uint8_t* extra,
                      int elen, Signature** out) {
  if (!sk || !out || mlen < 0 || elen < 0) return false;
  Affine<Fq_> h; int c;
  if (!hash_to_g1(composite, cip22, dom, msg, (size_t)mlen, extra, (size_t)elen, h, c)) return false;
  Xyzz<Fq_> r = scalar_mul_host(h, sk->k, 4);                 // PrivateKey::sign_raw (crates/bls-crypto/src/bls/secret.rs:65)
  Signature* s = new_signature();
  if (!s) return false;
  if (r.is_identity() || r.ZZ.is_zero_mod_p()) identity_jac<Fq_>(s->xyz);
  else { Fq_::mul(r.X, r.ZZ).to_ark(s->xyz); Fq_::mul(r.Y, r.ZZZ).to_ark(s->xyz + 6); r.ZZ.to_ark(s->xyz + 12); }
  *out = s;
  return true;
}
bool sign_message(const PrivateKey* sk, const uint8_t* msg, int mlen, const uint8_t* extra, int elen, bool composite, bool cip22,
                  Signature** out) {                                                                                  /* signatures.rs:44 */
  if (!hash_flags_supported(composite, cip22)) return false;
  return sign_with(sk, composite, cip22, SIG_DOMAIN, msg, mlen, extra, elen, out);
}
bool sign_pop(const PrivateKey* sk, const uint8_t* msg, int mlen, Signature** out) {                                  /* signatures.rs:74 */
  return sign_with(sk, false, false, POP_DOMAIN, msg, mlen, nullptr, 0, out);
}

// ---------------------------------------------------------------- verification (hash on the host, pairings / MSMs on the GPU)
static bool verify_with(const PublicKey* pk, bool composite, bool cip22, const uint8_t* dom, const uint8_t* msg, int mlen, const uint8_t* extra,
                        int elen, const Signature* sig, bool* out_verified) {
  if (!pk || !sig || !out_verified || mlen < 0 || elen < 0) return false;
  PhaseLog ph("verify_with");
  Affine<Fq_> h; int c;
  if (!hash_to_g1(composite, cip22, dom, msg, (size_t)mlen, extra, (size_t)elen, h, c)) return false;
  uint64_t hxy[12];
  h.x.to_ark(hxy); h.y.to_ark(hxy + 6);
  ph.mark("message -> G1 (host)");
  return celo_amd_verify_hash(pk, hxy, sig, out_verified);
}
bool verify_signature(const PublicKey* pk, const uint8_t* msg, int mlen, const uint8_t* extra, int elen, const Signature* sig,
                      bool composite, bool cip22, bool* out_verified) {                                               /* signatures.rs:244 */
  if (!hash_flags_supported(composite, cip22)) return false;
  return verify_with(pk, composite, cip22, SIG_DOMAIN, msg, mlen, extra, elen, sig, out_verified);
}
bool verify_pop(const PublicKey* pk, const uint8_t* msg, int mlen, const Signature* sig, bool* out_verified) {        /* signatures.rs:407 */
  return verify_with(pk, false, false, POP_DOMAIN, msg, mlen, nullptr, 0, sig, out_verified);
}
// Signature::batch_verify (crates/bls-crypto/src/bls/signature.rs:101-155): aggregate the signatures, hash every message,
// ONE (n+1)-pair product on the GPU
bool batch_verify_signature(const MessageFFI* messages, size_t n, bool composite, bool cip22, bool* verified) {       /* signatures.rs:290 */
  if (!hash_flags_supported(composite, cip22) || (!messages && n) || !verified) return false;
  std::vector<uint64_t> sigs(n * 18), pkj(n * 36);
  for (size_t i = 0; i < n; i++) {
    if (!messages[i].public_key || !messages[i].sig) return false;
    memcpy(&sigs[i * 18], messages[i].sig->xyz, 144);
    memcpy(&pkj[i * 36], messages[i].public_key->xyz, 288);
  }
  uint64_t asig[18];
  if (celo_amd_sum_jacobian_bls12_377_g1(sigs.data(), n, asig) != 0) return false;
  std::vector<uint64_t> g1((n + 1) * 12), g2((n + 1) * 24);
  std::vector<uint8_t> i1(n + 1, 0), i2(n + 1, 0);
  batch_to_affine<Fq_>(asig, 1, g1.data(), i1.data());
  neg_g2_generator(g2.data());
  batch_to_affine<Fq2_>(pkj.data(), n, g2.data() + 24, i2.data() + 1);
  std::vector<HashJob> jobs(n);
  for (size_t i = 0; i < n; i++) jobs[i] = {messages[i].data.ptr, messages[i].data.len, messages[i].extra.ptr, messages[i].extra.len, &g1[(i + 1) * 12]};
  if (n && !hash_many(composite, cip22, SIG_DOMAIN, jobs)) { *verified = false; return true; }   // a message that does not hash: not verified, not an error
  int one = 0;
  if (pairing_product_is_one_bls12_377(g1.data(), i1.data(), g2.data(), i2.data(), n + 1, &one) != 0) return false;
  *verified = one != 0;
  return true;
}
// Batch::verify per batch (crates/bls-crypto/src/bls/batch.rs:44-84), all batches at once: random exponents from the OS
// RNG, all G2 MSMs in one call, all G1 MSMs in one call, all 2-pair checks in one call.  out_results is always filled;
// the return value is false if any batch fails (signatures.rs:392-400).
bool batch_verify_strict(const BatchMessageFFI* batches, size_t m, bool composite, bool cip22, bool* out_results) {   /* signatures.rs:343 */
  if ((!batches && m) || !out_results) return false;
  PhaseLog ph("batch_verify_strict");
  for (size_t i = 0; i < m; i++) out_results[i] = false;      // always filled: every early return below leaves "not verified"
  if (!composite && cip22) return false;                      // per-batch false (signatures.rs:387)
  if (m == 0) return true;
  std::vector<uint32_t> offs(m + 1, 0);
  std::vector<size_t> blen(m);
  for (size_t b = 0; b < m; b++) {
    // Batch::verify zips keys with signatures (batch.rs:60-64): a longer side is truncated to the shorter
    blen[b] = batches[b].public_keys_len < batches[b].signatures_len ? batches[b].public_keys_len : batches[b].signatures_len;
    if (blen[b] && (!batches[b].public_keys || !batches[b].signatures)) return false;    // a length without its array
    offs[b + 1] = offs[b] + (uint32_t)blen[b];
  }
  const size_t tot = offs[m];
  // Per device (a caller bound to another device with celo_amd_use_device must not run its engines against device-0 memory and a
  // device-0 stream): one grow-only PINNED host staging buffer, its device twin, one copy stream, and the MIRRORS of the two handle
  // arenas - for every arena slot the affine point (192 / 96 bytes) and an identity byte in HBM, tagged on the host with the serial of
  // the allocation they were uploaded for.  A call then moves, per signer, two 4-byte slot numbers (the exponents are drawn on the device); only
  // handles the device has not seen in their present state (new since the last call, or a reused slot) are normalised and uploaded,
  // and the dense point arrays the batch MSMs read are gathered on the GPU (unit_batchverify.hip: k_mirror_scatter / k_mirror_gather).
  // Round 3 gathered and shipped 320 bytes per signer on every call: 8 ms before the G2 MSM - the longest leg - could start.
  if (api_enter() != 0) return false;
  struct Mirror {
    uint64_t* d_xy = nullptr; uint8_t* d_inf = nullptr; uint32_t cap = 0;
    std::unique_ptr<std::atomic<uint64_t>[]> seen;                      // serial of the allocation whose point sits in the slot's row (0: none)
    bool grow(uint32_t need, int words) {
      if (need <= cap) return true;
      uint64_t ncap64 = ((uint64_t)need + 0xffffu) & ~(uint64_t)0xffffu;
      if (ncap64 < 2 * (uint64_t)cap) ncap64 = 2 * (uint64_t)cap;
      if (ncap64 > 0xffffffffu) ncap64 = 0xffffffffu;
      const uint32_t ncap = (uint32_t)ncap64;
      uint64_t* nx = nullptr; uint8_t* ni = nullptr;
      std::unique_ptr<std::atomic<uint64_t>[]> ns(new (std::nothrow) std::atomic<uint64_t>[ncap]);
      if (!ns || hipMalloc((void**)&nx, (size_t)ncap * words * 8) != hipSuccess) return false;
      if (hipMalloc((void**)&ni, ncap) != hipSuccess) { (void)hipFree(nx); return false; }
      if (cap && (hipMemcpy(nx, d_xy, (size_t)cap * words * 8, hipMemcpyDeviceToDevice) != hipSuccess ||
                  hipMemcpy(ni, d_inf, cap, hipMemcpyDeviceToDevice) != hipSuccess)) { (void)hipFree(nx); (void)hipFree(ni); return false; }
      for (uint32_t i = 0; i < ncap; i++) ns[i].store(i < cap ? seen[i].load(std::memory_order_relaxed) : 0, std::memory_order_relaxed);
      if (d_xy) (void)hipFree(d_xy);
      if (d_inf) (void)hipFree(d_inf);
      d_xy = nx; d_inf = ni; cap = ncap; seen = std::move(ns);
      return true;
    }
    void forget() { for (uint32_t i = 0; i < cap; i++) seen[i].store(0, std::memory_order_relaxed); }
  };
  // Per device: the mirrors, the pinned host staging and the upload buffer are shared by the calls and used under `mu` - but only for the
  // MIRROR PHASE of a call (host pass over the handles, new rows across, the dense arrays gathered: a few ms); the gathered arrays, the
  // exponents and the hash points live in a per-call lease from `dpool`, so the lock is released before the call's long part - two batch
  // MSMs and the pairing checks on pooled, lock-free engines - and two host threads verifying different epochs overlap there (round 5,
  // the reference's batch_verify_strict is re-entrant, crates/bls-snark-sys/src/signatures.rs:343; rounds 3-4 held the
  // lock for the whole call).
  struct DStage { uint8_t* p = nullptr; size_t cap = 0; bool busy = false; };
  struct DevStage {
    std::mutex mu; uint8_t* stage = nullptr; size_t stage_cap = 0;
    uint8_t* d_up = nullptr; size_t d_up_cap = 0; hipStream_t copy_stream = nullptr; Mirror keys, sigs;
    std::mutex pool_mu; std::condition_variable pool_cv; DStage dpool[8];
  };
  static DevStage dev_stage[MAX_DEVICES];
  DevStage& DS = dev_stage[api_device()];
  struct DLease {
    DevStage& ds; DStage* slot = nullptr;
    explicit DLease(DevStage& d) : ds(d) {}
    uint8_t* take(size_t need) {          // a free buffer of the pool, grown if it is too small; waits while all eight are out
      std::unique_lock<std::mutex> lk(ds.pool_mu);
      for (;;) {
        DStage *fit = nullptr, *big = nullptr;          // the smallest free buffer that holds the call, else the largest free one (to be grown)
        for (DStage& q : ds.dpool) {
          if (q.busy) continue;
          if (q.cap >= need && (!fit || q.cap < fit->cap)) fit = &q;
          if (!big || q.cap > big->cap) big = &q;
        }
        DStage* best = fit ? fit : big;
        if (best) { best->busy = true; slot = best; break; }
        ds.pool_cv.wait(lk);
      }
      lk.unlock();
      if (slot->cap < need) {
        if (slot->p) (void)hipFree(slot->p);
        slot->p = nullptr; slot->cap = 0;
        if (hipMalloc((void**)&slot->p, need + need / 4) != hipSuccess) { slot->p = nullptr; return nullptr; }
        slot->cap = need + need / 4;
      }
      return slot->p;
    }
    ~DLease() {
      if (!slot) return;
      { std::lock_guard<std::mutex> lk(ds.pool_mu); slot->busy = false; }
      ds.pool_cv.notify_one();
    }
  } dlease(DS);                            // (declared before the lock: released after everything below, when bv_finish has drained the GPU work)
  std::unique_lock<std::mutex> stage_lk(DS.mu);
  uint8_t*& stage = DS.stage;
  size_t& stage_cap = DS.stage_cap;
  // host staging: rows of the handles to upload (worst case: every signer's) and the slot numbers
  const size_t need = tot * (24 + 12) * 8 + tot * 16 + 2 * tot + 4096;
  if (need > stage_cap) {
    if (stage) (void)hipHostFree(stage);
    stage = nullptr; stage_cap = 0;
    if (hipHostMalloc((void**)&stage, need + need / 4, hipHostMallocDefault) != hipSuccess) { log_err("batch_verify_strict: pinned staging allocation failed"); return false; }
    stage_cap = need + need / 4;
  }
  uint64_t* up_pk_xy = (uint64_t*)stage;
  uint64_t* up_sg_xy = up_pk_xy + tot * 24;
  uint32_t* idx_pk = (uint32_t*)(up_sg_xy + tot * 12);
  uint32_t* idx_sg = idx_pk + tot;
  uint32_t* up_pk_slot = idx_sg + tot;
  uint32_t* up_sg_slot = up_pk_slot + tot;
  uint8_t* up_pk_inf = (uint8_t*)(up_sg_slot + tot);
  uint8_t* up_sg_inf = up_pk_inf + tot;
  // device staging: the gathered point arrays, the exponents, the slot numbers; m hash points and flags behind them
  hipStream_t& copy_stream = DS.copy_stream;
  const size_t d_body = tot * (24 + 12 + 4) * 8 + tot * 8 + 2 * tot;
  const size_t d_need = d_body + 256 + m * 97 + 256 + (m + 1) * 4 + 4096;
  uint8_t* const d_stage = dlease.take(d_need);
  if (!d_stage) { log_err("batch_verify_strict: device staging allocation failed"); return false; }
  if (!copy_stream && hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) { copy_stream = nullptr; return false; }
  uint64_t* d_pk_xy = (uint64_t*)d_stage;
  uint64_t* d_sg_xy = d_pk_xy + tot * 24;
  uint64_t* d_sc = d_sg_xy + tot * 12;
  uint32_t* d_idx_pk = (uint32_t*)(d_sc + tot * 4);
  uint32_t* d_idx_sg = d_idx_pk + tot;
  uint8_t* d_pk_inf = (uint8_t*)(d_idx_sg + tot);
  uint8_t* d_sg_inf = d_pk_inf + tot;
  uint64_t* d_hxy = (uint64_t*)(d_stage + ((d_body + 255) & ~size_t(255)));
  uint8_t* d_hinf = (uint8_t*)(d_hxy + m * 12);
  uint32_t* d_offs = (uint32_t*)(((uintptr_t)(d_hinf + m) + 255) & ~uintptr_t(255));
  // the mirrors cover every slot THIS call's lists name (not every slot either arena has ever handed out - a process that holds
  // millions of live handles and verifies a few would pay 192 + 96 B of HBM per handle it never shows the device).  The largest slot of
  // the call is looked up only when a mirror is smaller than its arena - new handles since the last call, when rows are about to be
  // normalised and uploaded anyway; in the steady state (same validator set, epoch after epoch) the check is two comparisons.
  {
    uint32_t need_k = 0, need_s = 0;
    const uint32_t hw_k = pk_arena().high_water(), hw_s = sig_arena().high_water();
    if (hw_k > DS.keys.cap || hw_s > DS.sigs.cap) {
      // (a destroyed or foreign handle carries serial 0 or a garbage slot - neither may size an allocation; slots the arenas never
      // handed out are ignored here and the workers below fail the call with bad_handle as before)
      for (size_t b = 0; b < m; b++)
        for (size_t i = 0; i < blen[b]; i++) {
          const PublicKey* pk = batches[b].public_keys[i];
          const Signature* sg = batches[b].signatures[i];
          if (pk && pk->serial != 0 && pk->slot < hw_k && pk->slot >= need_k) need_k = pk->slot + 1;       // (null handles: the workers below fail the call)
          if (sg && sg->serial != 0 && sg->slot < hw_s && sg->slot >= need_s) need_s = sg->slot + 1;
        }
    }
    if (!DS.keys.grow(need_k, 24) || !DS.sigs.grow(need_s, 12)) { log_err("batch_verify_strict: device mirror allocation failed"); return false; }
  }
  Mirror& MK = DS.keys;
  Mirror& MS = DS.sigs;
  // `seen` is advanced by the workers BEFORE the rows are on the device: any failure between the first claim and the last scatter drops both
  // tag sets - under the lock, at the end of the mirror phase below (nothing between here and there returns)
  ChaCha20Rng master;
  if (!os_seeded_rng(master)) { log_err("batch_verify_strict: no OS randomness"); return false; }
  ph.mark("validate + allocate");
  // the message hashes depend on nothing computed here: they run on the host cores while the GPU does the two MSMs
  std::vector<uint64_t> hxy(m * 12);
  std::vector<HashJob> jobs(m);
  for (size_t b = 0; b < m; b++) jobs[b] = {batches[b].data.ptr, batches[b].data.len, batches[b].extra.ptr, batches[b].extra.len, &hxy[b * 12]};
  bool hash_ok = false;
  std::vector<uint8_t> hash_failed(m, 0);
  std::thread hasher([&]() { hash_ok = hash_many(composite, cip22, SIG_DOMAIN, jobs, &hash_failed); });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } hasher_guard{hasher};   // early returns must not leave it running
  // Host pass over the handles, ranges of batches across host threads, keys and then signatures (the exponents are drawn on the device,
  // from a ChaCha20 key taken from the OS-seeded master stream: bv_draw_exponents): a handle contributes its slot number; one whose
  // serial differs from the mirror's tag is claimed by exactly one thread (atomic exchange on the tag) and staged for upload - copied
  // straight if its Z is 1 (everything that came from the wire), otherwise normalised with one shared inversion per thread.
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 64) nt = 64;
  if (nt < 1 || tot < 8192) nt = 1;
  uint32_t exp_key[8];                                                  // this call's exponent stream (unit_batchverify.hip k_draw_exponents)
  for (int k = 0; k < 8; k++) exp_key[k] = master.next_u32();
  uint64_t one2[12], one1[6];
  Fq2_::one().to_ark(one2);
  Fq_::one().to_ark(one1);
  // Each pass runs in PHASES phases over the batch range that all workers share, and a phase stages its new rows in a region of its own
  // (the rows of the positions before it are the worst case before it): the calling thread ships a finished phase's rows while the
  // workers stage the next - what matters when many handles are new to the device (every signature of a block is; the first call of a
  // process brings 10^6 keys too)
  constexpr size_t PHASES = 4;
  std::atomic<size_t> up_n[2][PHASES];
  std::atomic<unsigned> phase_done[2][PHASES];
  std::atomic<bool> bad_handle(false);
  std::atomic<bool> work_failed(false);        // a range threw (out of memory in its vectors): the call fails, on whichever thread the range ran
  for (int q = 0; q < 2; q++) for (size_t f = 0; f < PHASES; f++) { up_n[q][f].store(0); phase_done[q][f].store(0); }
  auto phase_lo = [&](size_t f) { return m * f / PHASES; };
  auto work = [&](unsigned t, int pass) {
   for (size_t f = 0; f < PHASES; f++) {
    // a range never lets an exception out - the pool's workers, the fallback threads and the inline calls all run it - and always
    // counts its phase as done: the calling thread waits on that counter
    try {
    const size_t p_lo = phase_lo(f), p_hi = phase_lo(f + 1);
    const size_t b_lo = p_lo + (p_hi - p_lo) * t / nt, b_hi = p_lo + (p_hi - p_lo) * (t + 1) / nt;
    const bool is_pk = pass == 0;
    Mirror& M = is_pk ? MK : MS;
    std::vector<const uint64_t*> todo;          // xyz of the handles this thread uploads (slot: behind the limbs, read again below)
    std::vector<uint32_t> todo_slot;
    for (size_t b = b_lo; b < b_hi; b++) {
      const size_t n = blen[b];
      for (size_t i = 0; i < n; i++) {
        const size_t at = offs[b] + i;
        uint32_t slot; uint64_t serial; const uint64_t* xyz;
        if (is_pk) { const PublicKey* h = batches[b].public_keys[i]; if (h) { slot = h->slot; serial = h->serial; xyz = h->xyz; } else { slot = 0; serial = 0; xyz = nullptr; } }
        else { const Signature* h = batches[b].signatures[i]; if (h) { slot = h->slot; serial = h->serial; xyz = h->xyz; } else { slot = 0; serial = 0; xyz = nullptr; } }
        if (serial == 0 || slot >= M.cap) { bad_handle = true; slot = 0; }      // null, a destroyed handle, or not one of this library's
        else if (M.seen[slot].load(std::memory_order_relaxed) != serial && M.seen[slot].exchange(serial, std::memory_order_relaxed) != serial) {
          todo.push_back(xyz); todo_slot.push_back(slot);
        }
        (is_pk ? idx_pk : idx_sg)[at] = slot;
      }
    }
    if (!todo.empty()) {
      const int A3 = is_pk ? 36 : 18, A2 = is_pk ? 24 : 12;
      const size_t base = offs[p_lo] + up_n[pass][f].fetch_add(todo.size());     // the phase's region starts at its first position
      uint64_t* up_xy = is_pk ? up_pk_xy : up_sg_xy;
      uint32_t* up_slot = is_pk ? up_pk_slot : up_sg_slot;
      uint8_t* up_inf = is_pk ? up_pk_inf : up_sg_inf;
      std::vector<uint32_t> nz;                                         // not affine yet (aggregates, fresh signatures, identities)
      for (size_t k = 0; k < todo.size(); k++) {
        up_slot[base + k] = todo_slot[k];
        const bool affine = is_pk ? memcmp(todo[k] + 24, one2, 96) == 0 : memcmp(todo[k] + 12, one1, 48) == 0;
        if (affine) { memcpy(up_xy + (base + k) * A2, todo[k], (size_t)A2 * 8); up_inf[base + k] = 0; } else nz.push_back((uint32_t)k);
      }
      if (!nz.empty()) {
        std::vector<uint64_t> jac(nz.size() * A3), xy(nz.size() * A2);
        std::vector<uint8_t> inf(nz.size());
        for (size_t k = 0; k < nz.size(); k++) memcpy(&jac[k * A3], todo[nz[k]], (size_t)A3 * 8);
        if (is_pk) batch_to_affine_range<Fq2_>(jac.data(), nz.size(), xy.data(), inf.data());
        else batch_to_affine_range<Fq_>(jac.data(), nz.size(), xy.data(), inf.data());
        for (size_t k = 0; k < nz.size(); k++) {
          memcpy(up_xy + (base + nz[k]) * A2, &xy[k * A2], (size_t)A2 * 8);
          up_inf[base + nz[k]] = inf[k];
        }
      }
    }
    } catch (...) { work_failed = true; }
    phase_done[pass][f].fetch_add(1);
   }
  };
  // calling thread: when a pass is through, the new rows go to their mirror slots, the slot numbers (and, with the keys, the exponents)
  // cross PCIe, the GPU gathers the dense arrays, and that leg's batch MSM starts - the G2 leg (the longest of the chain) first, the
  // signatures' host pass runs under it
  bool copy_failed = false;
  // a finished phase's new rows: staging -> device -> their mirror slots (enqueued on the copy stream; the device buffer is reused
  // phase after phase in stream order)
  auto ship_phase = [&](int pass, size_t f) -> bool {
    const bool is_pk = pass == 0;
    Mirror& M = is_pk ? MK : MS;
    const int W = is_pk ? 24 : 12;
    const size_t k = up_n[pass][f].load(), base = offs[phase_lo(f)];
    if (!k) return true;
    const size_t row = (size_t)W * 8, upb = k * (row + 5) + 256;
    if (upb > DS.d_up_cap) {
      if (hipStreamSynchronize(copy_stream) != hipSuccess) return false;        // an earlier phase may still be reading the old buffer
      if (DS.d_up) (void)hipFree(DS.d_up);
      DS.d_up = nullptr; DS.d_up_cap = 0;
      if (hipMalloc((void**)&DS.d_up, upb + upb / 4) != hipSuccess) return false;
      DS.d_up_cap = upb + upb / 4;
    }
    uint64_t* du_xy = (uint64_t*)DS.d_up;
    uint32_t* du_slot = (uint32_t*)(du_xy + k * W);
    uint8_t* du_inf = (uint8_t*)(du_slot + k);
    return hipMemcpyAsync(du_xy, (is_pk ? up_pk_xy : up_sg_xy) + base * W, k * row, hipMemcpyHostToDevice, copy_stream) == hipSuccess &&
           hipMemcpyAsync(du_slot, (is_pk ? up_pk_slot : up_sg_slot) + base, k * 4, hipMemcpyHostToDevice, copy_stream) == hipSuccess &&
           hipMemcpyAsync(du_inf, (is_pk ? up_pk_inf : up_sg_inf) + base, k, hipMemcpyHostToDevice, copy_stream) == hipSuccess &&
           bv_mirror_scatter(W, du_xy, du_inf, du_slot, M.d_xy, M.d_inf, k, copy_stream) == 0;
  };
  auto stage_pass = [&](int pass) -> bool {
    const bool is_pk = pass == 0;
    Mirror& M = is_pk ? MK : MS;
    const int W = is_pk ? 24 : 12;
    if (tot) {
      if (hipMemcpyAsync(is_pk ? d_idx_pk : d_idx_sg, is_pk ? idx_pk : idx_sg, tot * 4, hipMemcpyHostToDevice, copy_stream) != hipSuccess) return false;
      if (is_pk && (hipMemcpyAsync(d_offs, offs.data(), (m + 1) * 4, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
                    bv_draw_exponents(exp_key, d_offs, m, tot, d_sc, copy_stream) != 0)) return false;
      if (bv_mirror_gather(W, M.d_xy, M.d_inf, is_pk ? d_idx_pk : d_idx_sg, is_pk ? d_pk_xy : d_sg_xy, is_pk ? d_pk_inf : d_sg_inf, tot, copy_stream) != 0) return false;
    }
    return hipStreamSynchronize(copy_stream) == hipSuccess;
  };
  BvJob job;
  int rc_keys = 0, rc_sigs = 0;
  bool pool_failed = false;
  {
    std::vector<std::thread> th;
    struct JoinAll { std::vector<std::thread>& v; ~JoinAll() { for (auto& t : v) if (t.joinable()) t.join(); } } guard{th};
    HostPool& pool = HostPool::get();
    const bool pooled = nt > 1 && pool.try_begin(nt, [&](unsigned t) { work(t, 0); work(t, 1); });
    struct PoolWait { HostPool& p; bool on; bool& failed; ~PoolWait() { if (on && !p.finish()) failed = true; } } pool_wait{pool, pooled, pool_failed};   // the job holds references to this frame
    if (!pooled) {
      if (nt > 1) {
        try { for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t]() { work(t, 0); work(t, 1); }); }
        catch (...) { for (unsigned t = (unsigned)th.size(); t < nt; t++) { work(t, 0); work(t, 1); } }   // no more threads: the rest of the ranges here
      } else { work(0, 0); work(0, 1); }
    }
    for (int pass = 0; pass < 2 && !copy_failed; pass++) {
      for (size_t f = 0; f < PHASES; f++) {
        while (phase_done[pass][f].load() < nt) std::this_thread::yield();
        if (bad_handle || work_failed || !ship_phase(pass, f)) { copy_failed = true; break; }
      }
      if (copy_failed) break;
      ph.mark(pass == 0 ? "  host pass over the key handles" : "  host pass over the signature handles");
      if (bad_handle || !stage_pass(pass)) { copy_failed = true; break; }
      ph.mark("  rows / slots across, gathered on the device");
      if (pass == 0) { rc_keys = bv_begin_keys(&job, d_pk_xy, d_pk_inf, d_sc, 1, offs.data(), m); ph.mark("key slots + exponents across, G2 batch MSM started"); }
      else { rc_sigs = bv_begin_sigs(&job, d_sg_xy, d_sg_inf, d_sc, 1, offs.data(), m); ph.mark("signature slots across, G1 batch MSM started"); }
    }
  }
  if (pool_failed || work_failed) copy_failed = true;                   // a range threw (out of memory): rows may be missing
  if (copy_failed) {
    MK.forget(); MS.forget();
    (void)hipStreamSynchronize(copy_stream);      //  copies and scatters of the phases shipped before the failure may still be queued on the shared
                                                  // stream, reading the shared pinned stage and d_up: they end before the next call on this device may stage
  }
  stage_lk.unlock();                                                    // end of the mirror phase: the next call on this device may stage while this one computes
  if (bad_handle) log_err("batch_verify_strict: a destroyed or foreign handle in the batch lists");
  hasher.join();
  ph.mark("  message hashes joined");
  std::vector<uint8_t> hinf(m, 0), ok(m, 0);
  if (hash_ok) for (size_t b = 0; b < m; b++) if (hash_failed[b]) hinf[b] = 1;   // no H(m): that pair is left out, the verdict is forced below
  uint64_t ng2[24];
  neg_g2_generator(ng2);
  const bool staged = hash_ok && !copy_failed && rc_keys == 0 && rc_sigs == 0 &&
                      hipMemcpyAsync(d_hxy, hxy.data(), m * 96, hipMemcpyHostToDevice, copy_stream) == hipSuccess &&
                      hipMemcpyAsync(d_hinf, hinf.data(), m, hipMemcpyHostToDevice, copy_stream) == hipSuccess &&
                      hipStreamSynchronize(copy_stream) == hipSuccess;
  ph.mark("  hashes across");
  if (bv_finish(&job, staged ? 1 : 0, d_hxy, d_hinf, 1, ng2, m, ok.data()) != 0 || !staged) return false;   // (bv_finish releases the engines either way)
  ph.mark("pairs -> pairing checks (GPU, chained)");
  bool all = true;
  for (size_t b = 0; b < m; b++) { out_results[b] = ok[b] != 0 && !hash_failed[b]; all = all && out_results[b]; }
  ph.mark("results");
  return all;
}

// ---------------------------------------------------------------- Groth16 verification over BW6-761 through the FFI
// crates/bls-snark-sys/src/snark/mod.rs:23-45 -> crates/epoch-snark/src/api/verifier.rs:23-40 -> ark_groth16::verify_proof.
// Decoding, hashing and packing are host plumbing; the two input scalar-muls go through msm_bw6_761_g1 and the check
//   e(A,B) * e(acc,-gamma) * e(C,-delta) * e(-alpha,beta) == 1   through pairing_product_is_one_bw6_761 (GPU).
bool verify(const uint8_t* vk, uint32_t vk_len, const uint8_t* proof, uint32_t proof_len, EpochBlockFFI first_epoch, EpochBlockFFI last_epoch) {
  if (!vk || !proof || proof_len < 288 || vk_len < 392) return false;
  PhaseLog ph("verify");
  // ---- VerifyingKey = alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | u64 len | gamma_abc_g1[len];  Proof = A | B | C
  uint64_t nabc;
  memcpy(&nabc, vk + 384, 8);
  if (nabc != 3 || vk_len < 392 + 96 * nabc) { log_err("verify: vk must carry 2 public inputs"); return false; }
  // ten BW6-761 decompressions (a 761-bit square root each) and the two blocks' validator keys: independent, so they share the
  // host cores instead of queueing on one (they were a fifth of the call)
  EpochBlockHost first, last;
  Affine<Fw_> pts[10];                       // alpha, beta, gamma, delta, abc[0..2], A, B, C
  const uint8_t* src[10] = {vk, vk + 96, vk + 192, vk + 288, vk + 392, vk + 488, vk + 584, proof, proof + 96, proof + 192};
  const bool on_g2[10] = {false, true, true, true, false, false, false, false, true, false};
  bool okp[10], ok_first = false, ok_last = false;
  {
    std::vector<std::thread> th;
    th.emplace_back([&]() { ok_first = epoch_from_ffi(first_epoch, first); });
    th.emplace_back([&]() { ok_last = epoch_from_ffi(last_epoch, last); });
    for (int i = 0; i < 10; i++) th.emplace_back([&, i]() { bool inf = false; okp[i] = bw6_decompress(src[i], on_g2[i], pts[i], inf) && !inf; });
    for (auto& x : th) x.join();
  }
  if (!ok_first || !ok_last) { log_err("verify: bad epoch public keys"); return false; }
  for (int i = 0; i < 10; i++) if (!okp[i]) { log_err(i < 7 ? "verify: bad vk" : "verify: bad proof"); return false; }
  const Affine<Fw_>&alpha = pts[0], &beta = pts[1], &gamma = pts[2], &delta = pts[3], &A = pts[7], &B = pts[8], &Cc = pts[9];
  const Affine<Fw_>* abc = &pts[4];
  // ---- public inputs: Blake2s("ULforout") of the first epoch and of the last epoch + aggregated key, 512 bits, packed 376|136
  Bits fb, lb;
  epoch_bits_cip22(first, true, fb);
  epoch_bits_cip22(last, false, lb);
  uint64_t agg[36];
  if (celo_amd_sum_jacobian_bls12_377_g2(last.pubkeys_jac.data(), last.pubkeys.size(), agg) != 0) return false;
  encode_public_key_bits(lb, jac_to_affine_or_zero<Fq2_>(agg));   // (an aggregate that is the identity encodes as arkworks' zero(), as in the reference)
  std::vector<uint8_t> h1 = blake2s_out_domain(bits_be_to_bytes_le(fb)), h2 = blake2s_out_domain(bits_be_to_bytes_le(lb));
  Bits hb;
  bits_append_le(hb, h1.data(), 32, 256);
  bits_append_le(hb, h2.data(), 32, 256);
  uint64_t scalars[3 * 6];
  memset(scalars, 0, sizeof scalars);
  scalars[0] = 1;  // gamma_abc[0] enters with scalar 1
  for (int chunk = 0; chunk < 2; chunk++) {  // pack::<Fr, CAPACITY = 376>: big-endian bits
    size_t lo = chunk * 376, hi = lo + 376 < hb.size() ? lo + 376 : hb.size();
    uint64_t* sc = scalars + 6 * (chunk + 1);
    size_t nb = hi - lo;
    for (size_t i = 0; i < nb; i++)
      if (hb[lo + i]) { size_t bit = nb - 1 - i; sc[bit >> 6] |= 1ULL << (bit & 63); }
  }
  uint64_t bases[3 * 24], accj[36], accxy[24];
  for (int i = 0; i < 3; i++) bw6_store_xy(abc[i], bases + 24 * i);
  ph.mark("decode keys, vk, proof; public inputs");
  if (msm_bw6_761_g1(bases, nullptr, scalars, 3, accj) != 0) return false;
  ph.mark("3-term input MSM (GPU)");
  uint8_t ainf;
  batch_to_affine<Fw_>(accj, 1, accxy, &ainf);
  // ---- the 4-pair product
  uint64_t g1[4 * 24], g2[4 * 24];
  uint8_t i1[4] = {0, ainf, 0, 0}, i2[4] = {0, 0, 0, 0};
  bw6_store_xy(A, g1); bw6_store_xy(B, g2);
  memcpy(g1 + 24, accxy, 192); bw6_store_xy({gamma.x, fw_neg(gamma.y)}, g2 + 24);
  bw6_store_xy(Cc, g1 + 48); bw6_store_xy({delta.x, fw_neg(delta.y)}, g2 + 48);
  bw6_store_xy({alpha.x, fw_neg(alpha.y)}, g1 + 72); bw6_store_xy(beta, g2 + 72);
  int one = 0;
  if (pairing_product_is_one_bw6_761(g1, i1, g2, i2, 4, &one) != 0) return false;
  ph.mark("4-pair product check (GPU)");
  return one != 0;
}

static bool collect_pubkeys(const PublicKey* const* in, int n, std::vector<Affine<Fq2_>>& out) {
  if (n < 0 || (n > 0 && !in)) return false;
  out.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    if (!in[i]) return false;
    out[(size_t)i] = jac_to_affine_or_zero<Fq2_>(in[i]->xyz);
  }
  return true;
}
bool encode_epoch_block_to_bytes_cip22(unsigned short index, unsigned char round, const uint8_t* epoch_entropy, const uint8_t* parent_entropy,
                                       unsigned int maximum_non_signers, unsigned int maximum_validators, const PublicKey* const* added_public_keys,
                                       int added_public_keys_len, uint8_t** out_bytes, int* out_len, uint8_t** out_extra, int* out_extra_len) {
  if (!out_bytes || !out_len || !out_extra || !out_extra_len) return false;                                  /* snark/epoch_block.rs:17 */
  std::vector<Affine<Fq2_>> pks;
  if (!collect_pubkeys(added_public_keys, added_public_keys_len, pks)) return false;
  Bits eb, xb;                                                                                                /* epoch_block.rs:150-169 */
  encode_uint(xb, index, 2); encode_uint(xb, round, 1); encode_uint(xb, maximum_non_signers, 4);
  encode_entropy_bits(eb, epoch_entropy);
  encode_entropy_bits(eb, parent_entropy);
  for (const auto& pk : pks) encode_public_key_bits(eb, pk);
  for (size_t i = pks.size(); i < maximum_validators; i++) encode_public_key_bits(eb, g2_generator_affine());
  return emit(bits_be_to_bytes_le(eb), out_bytes, out_len) && emit(bits_be_to_bytes_le(xb), out_extra, out_extra_len);
}
bool encode_epoch_block_to_bytes(unsigned short index, unsigned int maximum_non_signers, const PublicKey* const* added_public_keys,
                                 int added_public_keys_len, uint8_t** out_bytes, int* out_len) {              /* snark/epoch_block.rs:69 */
  if (!out_bytes || !out_len) return false;
  std::vector<Affine<Fq2_>> pks;
  if (!collect_pubkeys(added_public_keys, added_public_keys_len, pks)) return false;
  Bits b;                                                                                                     /* epoch_block.rs:106-114 */
  encode_uint(b, index, 2); encode_uint(b, maximum_non_signers, 4);
  for (const auto& pk : pks) encode_public_key_bits(b, pk);
  return emit(bits_be_to_bytes_le(b), out_bytes, out_len);
}
}
