// ORACLE — test infrastructure only (see field.hpp header).
// C ABI over the CPU restatement so that tests/ and bench.py's cpu_baseline leg can
// drive it through ctypes.  Conventions (identical to the product's Seam B so the
// same buffers can be handed to both): limbs little-endian u64; field elements in
// Montgomery form, R = 2^384 (BLS12-377 Fq) / 2^768 (BW6-761 Fq); scalars canonical;
// affine points x||y with a separate infinity byte array (NULL = none infinite);
// Jacobian results X||Y||Z with Z == 0 for the identity.
#include "pairing.hpp"
#include <chrono>

using namespace orc;

namespace {
template <class F> constexpr int limbs_of();
template <> constexpr int limbs_of<Fq377>() { return 6; }
template <> constexpr int limbs_of<Fq761>() { return 12; }
template <> constexpr int limbs_of<Fq2_377>() { return 12; }

template <class F> void load_f(F& f, const u64* p);
template <> void load_f<Fq377>(Fq377& f, const u64* p) { memcpy(f.v, p, 48); }
template <> void load_f<Fq761>(Fq761& f, const u64* p) { memcpy(f.v, p, 96); }
template <> void load_f<Fq2_377>(Fq2_377& f, const u64* p) { memcpy(f.c0.v, p, 48); memcpy(f.c1.v, p + 6, 48); }
template <class F> void store_f(const F& f, u64* p);
template <> void store_f<Fq377>(const Fq377& f, u64* p) { memcpy(p, f.v, 48); }
template <> void store_f<Fq761>(const Fq761& f, u64* p) { memcpy(p, f.v, 96); }
template <> void store_f<Fq2_377>(const Fq2_377& f, u64* p) { memcpy(p, f.c0.v, 48); memcpy(p + 6, f.c1.v, 48); }

template <class F> std::vector<Affine<F>> load_affine(const u64* xy, const uint8_t* inf, size_t n) {
  constexpr int L = limbs_of<F>();
  std::vector<Affine<F>> v(n);
  for (size_t i = 0; i < n; i++) {
    load_f(v[i].x, xy + i * 2 * L);
    load_f(v[i].y, xy + i * 2 * L + L);
    v[i].inf = inf ? inf[i] != 0 : false;
  }
  return v;
}
template <class F> void store_jac(const Jac<F>& j, u64* out) {
  constexpr int L = limbs_of<F>();
  store_f(j.x, out);
  store_f(j.y, out + L);
  store_f(j.z, out + 2 * L);
}
template <class F> Jac<F> load_jac(const u64* in) {
  constexpr int L = limbs_of<F>();
  Jac<F> j;
  load_f(j.x, in);
  load_f(j.y, in + L);
  load_f(j.z, in + 2 * L);
  return j;
}

template <class F, int SL>
int msm_impl(const u64* xy, const uint8_t* inf, const u64* sc, size_t n, int bits, int threads, int naive, u64* out) {
  auto b = load_affine<F>(xy, inf, n);
  Jac<F> r = naive ? msm_naive<F, SL>(b.data(), sc, n) : msm_pippenger<F, SL>(b.data(), sc, n, bits, threads);
  store_jac(r, out);
  return 0;
}
template <class F, int SL> int mul_impl(const u64* xy, const u64* k, u64* out) {
  auto b = load_affine<F>(xy, nullptr, 1);
  store_jac(Jac<F>::from_affine(b[0]).template mul<SL>(k), out);
  return 0;
}
template <class F> int normalize_impl(const u64* jac, size_t n, u64* xy, uint8_t* inf) {
  constexpr int L = limbs_of<F>();
  std::vector<Jac<F>> in(n);
  std::vector<Affine<F>> o(n);
  for (size_t i = 0; i < n; i++) in[i] = load_jac<F>(jac + i * 3 * L);
  batch_normalize(in.data(), o.data(), n);
  for (size_t i = 0; i < n; i++) {
    store_f(o[i].x, xy + i * 2 * L);
    store_f(o[i].y, xy + i * 2 * L + L);
    inf[i] = o[i].inf;
  }
  return 0;
}
template <class F> int sum_impl(const u64* jac, size_t n, u64* out) {  // Signature::aggregate / PublicKey::aggregate
  constexpr int L = limbs_of<F>();
  Jac<F> acc = Jac<F>::identity();
  for (size_t i = 0; i < n; i++) acc = acc.add(load_jac<F>(jac + i * 3 * L));
  store_jac(acc, out);
  return 0;
}
void store_gt377(const Fq12_377& g, u64* out) {
  const Fq2_377* c[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
  for (int i = 0; i < 6; i++) store_f(*c[i], out + 12 * i);
}
Fq12_377 load_gt377(const u64* in) {
  Fq12_377 g;
  Fq2_377* c[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
  for (int i = 0; i < 6; i++) load_f(*c[i], in + 12 * i);
  return g;
}
void store_gt761(const Fq6_761& g, u64* out) {
  const Fq761* c[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
  for (int i = 0; i < 6; i++) store_f(*c[i], out + 12 * i);
}
}  // namespace

// radix-2 NTT, a template over the field: see the comment at orc_ntt_fq377 below
template <class F> static int ntt_impl(u64* data, unsigned log_n, const u64* omega6, const u64* coset6, int coset_after, const u64* scale6) {
  constexpr int A = F::N;
  const size_t n = size_t(1) << log_n;
  std::vector<F> a(n);
  for (size_t i = 0; i < n; i++) memcpy(a[i].v, data + A * i, 8 * A);
  F w, g, sc;
  memcpy(w.v, omega6, 8 * A);
  if (coset6) memcpy(g.v, coset6, 8 * A);
  if (scale6) memcpy(sc.v, scale6, 8 * A);
  if (coset6 && !coset_after) { F p = F::one(); for (size_t i = 0; i < n; i++) { a[i] = a[i] * p; p = p * g; } }
  for (size_t i = 0; i < n; i++) {  // bit reversal
    size_t r = 0;
    for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
    if (r > i) std::swap(a[i], a[r]);
  }
  for (unsigned s = 1; s <= log_n; s++) {
    const size_t m = size_t(1) << s;
    F wm = w;                                   // omega^(n/m)
    for (unsigned k = s; k < log_n; k++) wm = wm * wm;
    for (size_t k = 0; k < n; k += m) {
      F t = F::one();
      for (size_t j = 0; j < m / 2; j++) {
        F u = a[k + j], v = a[k + j + m / 2] * t;
        a[k + j] = u + v;
        a[k + j + m / 2] = u - v;
        t = t * wm;
      }
    }
  }
  if (coset6 && coset_after) { F p = F::one(); for (size_t i = 0; i < n; i++) { a[i] = a[i] * p; p = p * g; } }
  if (scale6) for (size_t i = 0; i < n; i++) a[i] = a[i] * sc;
  for (size_t i = 0; i < n; i++) memcpy(data + A * i, a[i].v, 8 * A);
  return 0;
}

extern "C" {

// ---- Montgomery conversion (count field elements)
int orc_to_mont_377(const u64* in, u64* out, size_t count) {
  for (size_t i = 0; i < count; i++) { Fq377 f = Fq377::from_canonical(in + 6 * i); memcpy(out + 6 * i, f.v, 48); }
  return 0;
}
int orc_from_mont_377(const u64* in, u64* out, size_t count) {
  for (size_t i = 0; i < count; i++) { Fq377 f; memcpy(f.v, in + 6 * i, 48); f.to_canonical(out + 6 * i); }
  return 0;
}
int orc_to_mont_761(const u64* in, u64* out, size_t count) {
  for (size_t i = 0; i < count; i++) { Fq761 f = Fq761::from_canonical(in + 12 * i); memcpy(out + 12 * i, f.v, 96); }
  return 0;
}
int orc_from_mont_761(const u64* in, u64* out, size_t count) {
  for (size_t i = 0; i < count; i++) { Fq761 f; memcpy(f.v, in + 12 * i, 96); f.to_canonical(out + 12 * i); }
  return 0;
}

// ---- radix-2 NTT over Fr(BW6-761) = Fq(BLS12-377): restatement of ark-poly 0.1 Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}
// _in_place (ark-poly is an un-vendored dependency, Cargo.lock:213-215; it is used inside ark_groth16::create_proof_no_zk called at
// crates/epoch-snark/src/api/prover.rs:78).  Textbook decimation-in-time: bit-reversal permutation, then log n butterfly
// levels.  The caller supplies the domain generator (omega for a forward transform, omega^-1 for an inverse one), the
// optional coset generator (forward: x_i *= g^i first; inverse: pass g^-1 and set coset_after: x_i *= g^-i last) and the
// optional final scale (n^-1).  data: n = 2^log_n elements, arkworks Montgomery limbs.  PARITY UNPINNED against the
// reference (it holds no NTT vector); pinned against the O(n^2) definition in oracle/py and by round-trip properties.
int orc_ntt_fq377(u64* data, unsigned log_n, const u64* omega6, const u64* coset6, int coset_after, const u64* scale6) {
  return ntt_impl<Fq377>(data, log_n, omega6, coset6, coset_after, scale6);
}
// the same transform over Fr(BLS12-377) (4-limb Montgomery elements): the hash-helper proof's witness map
// (crates/epoch-snark/src/api/prover.rs:83-118)
int orc_ntt_fr253(u64* data, unsigned log_n, const u64* omega4, const u64* coset4, int coset_after, const u64* scale4) {
  return ntt_impl<Fr253>(data, log_n, omega4, coset4, coset_after, scale4);
}
double orc_time_ntt_fq377(u64* data, unsigned log_n, const u64* omega6) {
  auto t0 = std::chrono::steady_clock::now();
  orc_ntt_fq377(data, log_n, omega6, nullptr, 0, nullptr);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---- MSM: naive != 0 selects the sum-of-scalar-muls definition instead of Pippenger
int orc_msm_bls12_377_g1(const u64* xy, const uint8_t* inf, const u64* sc, size_t n, int threads, int naive, u64* out18) {
  return msm_impl<Fq377, 4>(xy, inf, sc, n, 253, threads, naive, out18);
}
int orc_msm_bls12_377_g2(const u64* xy, const uint8_t* inf, const u64* sc, size_t n, int threads, int naive, u64* out36) {
  return msm_impl<Fq2_377, 4>(xy, inf, sc, n, 253, threads, naive, out36);
}
int orc_msm_bw6_761_g1(const u64* xy, const uint8_t* inf, const u64* sc, size_t n, int threads, int naive, u64* out36) {
  return msm_impl<Fq761, 6>(xy, inf, sc, n, 377, threads, naive, out36);
}
// BW6-761 G2 has the same coordinate field (Fq) and the same a = 0 formulas; only b differs, which
// the group law never uses, so the same routine serves both groups.
int orc_msm_bw6_761_g2(const u64* xy, const uint8_t* inf, const u64* sc, size_t n, int threads, int naive, u64* out36) {
  return msm_impl<Fq761, 6>(xy, inf, sc, n, 377, threads, naive, out36);
}

// ---- single scalar mul (k canonical, 4 / 6 limbs), Jacobian out
int orc_mul_bls12_377_g1(const u64* xy, const u64* k, u64* out18) { return mul_impl<Fq377, 4>(xy, k, out18); }
int orc_mul_bls12_377_g2(const u64* xy, const u64* k, u64* out36) { return mul_impl<Fq2_377, 4>(xy, k, out36); }
int orc_mul_bw6_761(const u64* xy, const u64* k, u64* out36) { return mul_impl<Fq761, 6>(xy, k, out36); }

// ---- Jacobian -> affine (Montgomery batch inversion), and plain sums
int orc_normalize_bls12_377_g1(const u64* jac, size_t n, u64* xy, uint8_t* inf) { return normalize_impl<Fq377>(jac, n, xy, inf); }
int orc_normalize_bls12_377_g2(const u64* jac, size_t n, u64* xy, uint8_t* inf) { return normalize_impl<Fq2_377>(jac, n, xy, inf); }
int orc_normalize_bw6_761(const u64* jac, size_t n, u64* xy, uint8_t* inf) { return normalize_impl<Fq761>(jac, n, xy, inf); }
int orc_sum_bls12_377_g1(const u64* jac, size_t n, u64* out) { return sum_impl<Fq377>(jac, n, out); }
int orc_sum_bls12_377_g2(const u64* jac, size_t n, u64* out) { return sum_impl<Fq2_377>(jac, n, out); }

// ---- pairings.  GT layout: 6 Fq2 (BLS12-377) or 6 Fq (BW6-761) coefficients in tower order
//      c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 ; 72 u64 either way.
int orc_miller_loop_bls12_377(const u64* g1, const uint8_t* inf1, const u64* g2, const uint8_t* inf2, size_t k, u64* gt72) {
  auto p = load_affine<Fq377>(g1, inf1, k);
  auto q = load_affine<Fq2_377>(g2, inf2, k);
  std::vector<Bls12_377::G2Prepared> prep(k);
  for (size_t i = 0; i < k; i++) prep[i] = Bls12_377::prepare(q[i]);
  store_gt377(Bls12_377::miller_loop(p.data(), prep.data(), k), gt72);
  return 0;
}
int orc_final_exp_bls12_377(const u64* in72, u64* out72) {
  store_gt377(Bls12_377::final_exponentiation(load_gt377(in72)), out72);
  return 0;
}
int orc_pairing_product_bls12_377(const u64* g1, const uint8_t* inf1, const u64* g2, const uint8_t* inf2, size_t k, u64* gt72, int* is_one) {
  auto p = load_affine<Fq377>(g1, inf1, k);
  auto q = load_affine<Fq2_377>(g2, inf2, k);
  Fq12_377 g = Bls12_377::product_of_pairings(p.data(), q.data(), k);
  if (gt72) store_gt377(g, gt72);
  if (is_one) *is_one = g.is_one();
  return 0;
}
int orc_pairing_product_bw6_761(const u64* g1, const uint8_t* inf1, const u64* g2, const uint8_t* inf2, size_t k, u64* gt72, int* is_one) {
  auto p = load_affine<Fq761>(g1, inf1, k);
  auto q = load_affine<Fq761>(g2, inf2, k);
  Fq6_761 g = Bw6_761::product_of_pairings(p.data(), q.data(), k);
  if (gt72) store_gt761(g, gt72);
  if (is_one) *is_one = (g == Fq6_761::one());
  return 0;
}
int orc_miller_loop_bw6_761(const u64* g1, const uint8_t* inf1, const u64* g2, const uint8_t* inf2, size_t k, u64* gt72) {
  auto p = load_affine<Fq761>(g1, inf1, k);
  auto q = load_affine<Fq761>(g2, inf2, k);
  std::vector<Bw6_761::G2Prepared> prep(k);
  for (size_t i = 0; i < k; i++) prep[i] = Bw6_761::prepare(q[i]);
  store_gt761(Bw6_761::miller_loop(p.data(), prep.data(), k), gt72);
  return 0;
}

// ---- timing helpers for bench.py's cpu_baseline leg (seconds of wall time)
double orc_time_msm_bls12_377_g1(const u64* xy, const u64* sc, size_t n, int threads, u64* out18) {
  auto t0 = std::chrono::steady_clock::now();
  msm_impl<Fq377, 4>(xy, nullptr, sc, n, 253, threads, 0, out18);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}  // extern "C"

// ---- compressed-point decoding (arkworks 0.1 GroupAffine::deserialize: x little-endian with the flag bits 0x80 "y is the
// larger root" / 0x40 infinity in the last byte, get_point_from_x, then is_in_correct_subgroup_assuming_on_curve = r*P == O;
// crates/bls-crypto/src/bls/public.rs:123-149, signature.rs:31-57).  status: 0 ok, 1 infinity, 2 invalid, 3 not in subgroup.
namespace {
const u64 R377_ORDER[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};
bool read_fq377(const uint8_t* in, Fq377& out) {
  u64 w[6];
  memcpy(w, in, 48);
  if (big_cmp<6>(w, Fq377::C().p) >= 0) return false;
  out = Fq377::from_canonical(w);
  return true;
}
template <class F> uint8_t finish_point(const F& x, const F& b, bool greatest, int check, u64* out) {
  constexpr int L = limbs_of<F>();
  F y;
  if (!(x.sqr() * x + b).sqrt(y)) return 2;
  if (y.lex_largest() != greatest) y = -y;
  Affine<F> p = {x, y, false};
  if (check && !Jac<F>::from_affine(p).template mul<4>(R377_ORDER).is_identity()) return 3;
  store_f(x, out);
  store_f(y, out + L);
  return 0;
}
uint8_t decode_one(int g2, const uint8_t* in, int check, u64* out) {
  const int nb = g2 ? 96 : 48;
  uint8_t buf[96];
  memcpy(buf, in, nb);
  const uint8_t flags = buf[nb - 1] & 0xC0;
  buf[nb - 1] &= 0x3F;
  memset(out, 0, (g2 ? 24 : 12) * 8);
  if (flags == 0xC0) return 2;      // ark-serialize SWFlags::from_u8 -> None
  if (flags & 0x40) return 1;
  if (g2) {
    Fq2_377 x;
    if (!read_fq377(buf, x.c0) || !read_fq377(buf + 48, x.c1)) return 2;
    return finish_point<Fq2_377>(x, Bls12_377::twist_b(), (flags & 0x80) != 0, check, out);
  }
  Fq377 x;
  if (!read_fq377(buf, x)) return 2;
  return finish_point<Fq377>(x, Fq377::one(), (flags & 0x80) != 0, check, out);
}
}  // namespace
extern "C" {
int orc_decompress_bls12_377(int g2, const uint8_t* in, size_t n, int check_subgroup, int threads, u64* out_xy, uint8_t* status) {
  const size_t nb = g2 ? 96 : 48, ow = g2 ? 24 : 12;
  if (threads < 1) threads = 1;
  (void)decode_one(g2, in, 0, out_xy);   // lazily built field constants before the threads start
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++)
    th.emplace_back([&, t]() {
      for (size_t i = n * t / threads; i < n * (t + 1) / threads; i++) status[i] = decode_one(g2, in + i * nb, check_subgroup, out_xy + i * ow);
    });
  for (auto& x : th) x.join();
  return 0;
}
double orc_time_decompress_bls12_377(int g2, const uint8_t* in, size_t n, int check_subgroup, int threads, u64* out_xy, uint8_t* status) {
  auto t0 = std::chrono::steady_clock::now();
  orc_decompress_bls12_377(g2, in, n, check_subgroup, threads, out_xy, status);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int orc_hardware_threads() { return (int)std::thread::hardware_concurrency(); }
}
