// Translation unit: the Groth16 prover's device work - what ark_groth16::create_proof_no_zk does after
// R1CS synthesis, for BOTH proofs of an epoch: the epoch proof over BW6-761 (crates/epoch-snark/src/api/prover.rs:78 ->
// generate_epoch_proof) and the hash-helper proof over BLS12-377 (prover.rs:83-118, create_proof_no_zk::<BLSCurve, _> at :112):
//   1. witness map (ark-groth16 r1cs_to_qap.rs R1CStoQAP::witness_map): the QAP evaluations a, b, c over the domain
//        ifft(a), ifft(b), ifft(c); coset_fft(a), coset_fft(b), coset_fft(c); ab = (a o b - c) / Z(coset); coset_ifft(ab) = h
//      seven radix-2 transforms over the curve's scalar field (ntt.h: Fr(BW6-761) = 377 bits, Fr(BLS12-377) = 253 bits) and one
//      pointwise kernel; Z is constant on the coset: g^n - 1.
//   2. the proof (create_proof with r = s = 0):
//        A = a_query[0] + MSM(a_query[1..], assignment) + alpha_g1
//        B = b_g2_query[0] + MSM(b_g2_query[1..], assignment) + beta_g2
//        C = MSM(l_query, aux_assignment) + MSM(h_query, h)
//      the four MSMs run concurrently on four engines (msm.h); the handful of point additions around them on the host.
// R1CS synthesis (evaluating the constraint matrices on the witness) is the circuit's business and stays with the caller.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp.h"
#include "runtime.h"
#include <thread>
#include <vector>
#include <cstring>

namespace celo {
typedef Fp<P377> Fr761;          // the scalar field of BW6-761 is the base field of BLS12-377 (ntt.h)
typedef Fp<P253> Fr377;          // the scalar field of BLS12-377
int ntt_run(uint64_t*, unsigned, const uint64_t*, const uint64_t*, int, const uint64_t*, int, void*);
int ntt_run_253(uint64_t*, unsigned, const uint64_t*, const uint64_t*, int, const uint64_t*, int, void*);
int msm_host_761(const uint64_t*, const uint8_t*, const uint64_t*, size_t, int, uint64_t*);
int sum_jac_761(const uint64_t*, size_t, uint64_t*);
int msm_host_g1_377(const uint64_t*, const uint8_t*, const uint64_t*, size_t, int, uint64_t*);
int msm_host_g2_377(const uint64_t*, const uint8_t*, const uint64_t*, size_t, int, uint64_t*);
int sum_jac_g1_377(const uint64_t*, size_t, uint64_t*);
int sum_jac_g2_377(const uint64_t*, size_t, uint64_t*);
struct FixedTable;                                   // msm.h: a query's fixed-base tables
int msm_fixed_build_761(const void*, const void*, size_t, int, int, FixedTable**);
int msm_fixed_run_761(const FixedTable*, const void*, size_t, int, uint64_t*, void*);
int msm_fixed_build_g1_377(const void*, const void*, size_t, int, int, FixedTable**);
int msm_fixed_run_g1_377(const FixedTable*, const void*, size_t, int, uint64_t*, void*);
int msm_fixed_build_g2_377(const void*, const void*, size_t, int, int, FixedTable**);
int msm_fixed_run_g2_377(const FixedTable*, const void*, size_t, int, uint64_t*, void*);
int fixed_table_release(FixedTable*);
template <class FR> struct NttOf;
template <> struct NttOf<Fr761> { static int run(uint64_t* d, unsigned l, const uint64_t* w, const uint64_t* g, int after, const uint64_t* sc, void* st) { return ntt_run(d, l, w, g, after, sc, 1, st); } };
template <> struct NttOf<Fr377> { static int run(uint64_t* d, unsigned l, const uint64_t* w, const uint64_t* g, int after, const uint64_t* sc, void* st) { return ntt_run_253(d, l, w, g, after, sc, 1, st); } };

// a[i] <- (a[i] b[i] - c[i]) z   (arkworks Montgomery limbs in and out; optionally the canonical integer: Fr::into_repr())
template <class FR>
__global__ void __launch_bounds__(256) k_qap_combine(uint64_t* __restrict__ a, const uint64_t* __restrict__ b, const uint64_t* __restrict__ c, uint32_t n,
                                                     const uint32_t* __restrict__ z_dev) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int A = FR::ARK64;
  const FR x = FR::from_ark(a + (size_t)i * A), y = FR::from_ark(b + (size_t)i * A), w = FR::from_ark(c + (size_t)i * A);
  const FR z = FR::load(z_dev);
  const FR t = FR::norm(FR::template sub<4, 1>(FR::mul(x, y), w));
  FR::mul(t, z).to_ark(a + (size_t)i * A);
}
template <class FR>
__global__ void __launch_bounds__(256) k_to_canonical(uint64_t* __restrict__ a, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  FR::from_ark(a + (size_t)i * FR::ARK64).to_canonical(a + (size_t)i * FR::ARK64);
}

#define PRV_OK(x)                                                                                                  \
  do {                                                                                                             \
    hipError_t e_ = (x);                                                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "[celo-amd] %s: %s\n", #x, hipGetErrorString(e_)); rc = 10; goto done; } \
  } while (0)

// dev = 1: a, b, c are DEVICE pointers (a is overwritten with h; b and c are overwritten with intermediate values).
template <class FR>
static int witness_map_t(uint64_t* a, uint64_t* b, uint64_t* c, unsigned log_n, const uint64_t* omega, const uint64_t* omega_inv, const uint64_t* coset,
                    const uint64_t* coset_inv, const uint64_t* n_inv, const uint64_t* z_inv, int out_canonical, int dev, void* stream_) {
  if (int rc0 = api_enter()) return rc0;
  if (!a || !b || !c || !omega || !omega_inv || !coset || !coset_inv || !n_inv || !z_inv || log_n > 28) return 2;
  const size_t n = size_t(1) << log_n, bytes = n * FR::ARK64 * 8;
  hipStream_t stream = (hipStream_t)stream_;
  uint64_t *da = a, *db = b, *dc = c;
  uint32_t* d_z = nullptr;
  int rc = 0;
  uint32_t zw[FR::WORDS];
  {
    FR v = FR::wred(FR::from_ark(z_inv));
    memset(zw, 0, sizeof zw);
    v.store(zw);
  }
  if (!dev) {
    da = db = dc = nullptr;
    PRV_OK(hipMalloc(&da, bytes)); PRV_OK(hipMalloc(&db, bytes)); PRV_OK(hipMalloc(&dc, bytes));
    PRV_OK(hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, stream));
    PRV_OK(hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, stream));
    PRV_OK(hipMemcpyAsync(dc, c, bytes, hipMemcpyHostToDevice, stream));
  }
  PRV_OK(hipMalloc(&d_z, sizeof zw));
  PRV_OK(hipMemcpyAsync(d_z, zw, sizeof zw, hipMemcpyHostToDevice, stream));
  // an NTT engine keeps the twiddle table of its last (omega, n) and the pool hands the same engine back to a serial caller: the
  // table is rebuilt three times per witness map (inverse, forward, inverse: ~20 us each at 2^20), not seven
  for (uint64_t* p : {da, db, dc}) if ((rc = NttOf<FR>::run(p, log_n, omega_inv, nullptr, 0, n_inv, stream))) goto done;        // ifft
  for (uint64_t* p : {da, db, dc}) if ((rc = NttOf<FR>::run(p, log_n, omega, coset, 0, nullptr, stream))) goto done;             // coset_fft
  hipLaunchKernelGGL((k_qap_combine<FR>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, da, db, dc, (uint32_t)n, d_z);
  if ((rc = NttOf<FR>::run(da, log_n, omega_inv, coset_inv, 1, n_inv, stream))) goto done;                                          // coset_ifft
  if (out_canonical) hipLaunchKernelGGL((k_to_canonical<FR>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, da, (uint32_t)n);
  PRV_OK(hipGetLastError());
  if (!dev) PRV_OK(hipMemcpyAsync(a, da, bytes, hipMemcpyDeviceToHost, stream));
  PRV_OK(hipStreamSynchronize(stream));
done:
  if (!dev) { if (da) (void)hipFree(da); if (db) (void)hipFree(db); if (dc) (void)hipFree(dc); }
  if (d_z) (void)hipFree(d_z);
  return rc;
}

int witness_map_run(uint64_t* a, uint64_t* b, uint64_t* c, unsigned log_n, const uint64_t* omega, const uint64_t* omega_inv, const uint64_t* coset,
                    const uint64_t* coset_inv, const uint64_t* n_inv, const uint64_t* z_inv, int out_canonical, int dev, void* stream_) {
  return witness_map_t<Fr761>(a, b, c, log_n, omega, omega_inv, coset, coset_inv, n_inv, z_inv, out_canonical, dev, stream_);
}
int witness_map_253_run(uint64_t* a, uint64_t* b, uint64_t* c, unsigned log_n, const uint64_t* omega, const uint64_t* omega_inv, const uint64_t* coset,
                        const uint64_t* coset_inv, const uint64_t* n_inv, const uint64_t* z_inv, int out_canonical, int dev, void* stream_) {
  return witness_map_t<Fr377>(a, b, c, log_n, omega, omega_inv, coset, coset_inv, n_inv, z_inv, out_canonical, dev, stream_);
}

// arkworks' GroupAffine::zero() handed over as coordinates: x = 0, y = 1 (never an element of a prime-order group here: msm.h)
static bool is_ark_zero(const uint64_t* xy, const uint64_t* one, int coord64, int one64) {
  uint64_t o = 0;
  for (int k = 0; k < coord64; k++) o |= xy[k] | (xy[coord64 + k] ^ (k < one64 ? one[k] : 0));
  return o == 0;
}
// all pointers HOST.  Queries: affine points, arkworks layout (24 u64 each).  assignment: n_assign = na - 1 canonical scalars (public
// inputs without the leading one, then the witness); aux = its last n_aux entries; h: n_h canonical scalars (the witness map's
// output).  Results: Jacobian (36 u64 each).
int groth16_prove_761_run(const uint64_t* a_query, size_t na, const uint64_t* b_g2_query, size_t nb, const uint64_t* h_query, size_t nh,
                          const uint64_t* l_query, size_t nl, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* assignment,
                          size_t n_assign, size_t n_aux, const uint64_t* h, size_t n_h, uint64_t* out_a, uint64_t* out_b, uint64_t* out_c) {
  if (int rc0 = api_enter()) return rc0;
  if (!a_query || !b_g2_query || !alpha_g1 || !beta_g2 || !out_a || !out_b || !out_c || na == 0 || nb == 0) return 2;
  if ((n_assign && !assignment) || n_aux > n_assign || (n_h && !h) || (nh && !h_query) || (nl && !l_query)) return 2;
  // VariableBaseMSM::multi_scalar_mul zips bases with scalars: the shorter side decides (ark-ec msm/variable_base.rs)
  const size_t ka = (na - 1 < n_assign) ? na - 1 : n_assign, kb = (nb - 1 < n_assign) ? nb - 1 : n_assign;
  const size_t kl = nl < n_aux ? nl : n_aux, kh = nh < n_h ? nh : n_h;
  const uint64_t* aux = assignment + (n_assign - n_aux) * 6;
  uint64_t acc[4][36];
  int rcs[4] = {0, 0, 0, 0};
  const int dev = api_device();
  auto run = [&](int i, const uint64_t* bases, const uint64_t* sc, size_t k) {
    rcs[i] = api_bind_thread(dev);
    if (!rcs[i]) rcs[i] = msm_host_761(bases, nullptr, sc, k, 2, acc[i]);      // flags 2: a query row (0, 1) is arkworks' identity (msm.h k_flag_ark_zero)
  };
  {
    std::thread t0(run, 0, a_query + 24, assignment, ka), t1(run, 1, b_g2_query + 24, assignment, kb), t2(run, 2, l_query, aux, kl);
    run(3, h_query, h, kh);
    t0.join(); t1.join(); t2.join();
  }
  for (int r : rcs) if (r) return r;
  // calculate_coeff(initial = 0, query, vk_param, assignment): query[0] + acc + vk_param
  auto affine_as_jac = [](const uint64_t* xy, uint64_t* j) {
    memcpy(j, xy, 192);
    Fq761d::one().to_ark(j + 24);
    if (is_ark_zero(xy, j + 24, 12, 12)) memset(j + 24, 0, 96);          // query[0] / a key element given as arkworks' identity: Z = 0
  };
  uint64_t terms[3][36];
  affine_as_jac(a_query, terms[0]); memcpy(terms[1], acc[0], 288); affine_as_jac(alpha_g1, terms[2]);
  if (int rc = sum_jac_761(&terms[0][0], 3, out_a)) return rc;
  affine_as_jac(b_g2_query, terms[0]); memcpy(terms[1], acc[1], 288); affine_as_jac(beta_g2, terms[2]);
  if (int rc = sum_jac_761(&terms[0][0], 3, out_b)) return rc;
  memcpy(terms[0], acc[2], 288); memcpy(terms[1], acc[3], 288);
  return sum_jac_761(&terms[0][0], 2, out_c);
}

// The same composition over BLS12-377 (the hash-helper proof, prover.rs:112): A, C and the a / h / l queries live in G1 (affine 12 u64,
// Jacobian 18), B and the b_g2 query in G2 (24 / 36), scalars are 4 u64.  (create_proof also accumulates B in G1 - g_b_g1 - for the
// r B term of C; with r = s = 0 that term vanishes whatever B's G1 image is, so it is not computed.)
int groth16_prove_377_run(const uint64_t* a_query, size_t na, const uint64_t* b_g2_query, size_t nb, const uint64_t* h_query, size_t nh,
                          const uint64_t* l_query, size_t nl, const uint64_t* alpha_g1, const uint64_t* beta_g2, const uint64_t* assignment,
                          size_t n_assign, size_t n_aux, const uint64_t* h, size_t n_h, uint64_t* out_a, uint64_t* out_b, uint64_t* out_c) {
  if (int rc0 = api_enter()) return rc0;
  if (!a_query || !b_g2_query || !alpha_g1 || !beta_g2 || !out_a || !out_b || !out_c || na == 0 || nb == 0) return 2;
  if ((n_assign && !assignment) || n_aux > n_assign || (n_h && !h) || (nh && !h_query) || (nl && !l_query)) return 2;
  const size_t ka = (na - 1 < n_assign) ? na - 1 : n_assign, kb = (nb - 1 < n_assign) ? nb - 1 : n_assign;
  const size_t kl = nl < n_aux ? nl : n_aux, kh = nh < n_h ? nh : n_h;
  const uint64_t* aux = assignment + (n_assign - n_aux) * 4;
  uint64_t acc1[3][18], acc2[36];
  int rcs[4] = {0, 0, 0, 0};
  const int dev = api_device();
  auto run1 = [&](int i, uint64_t* out, const uint64_t* bases, const uint64_t* sc, size_t k) {
    rcs[i] = api_bind_thread(dev);
    if (!rcs[i]) rcs[i] = msm_host_g1_377(bases, nullptr, sc, k, 3, out);      // flags: 1 = a proving key's G1 queries are elements of G1 (the GLV split applies, msm.h k_glv_expand), 2 = rows (0, 1) are the identity
  };
  auto run2 = [&]() {
    rcs[1] = api_bind_thread(dev);
    if (!rcs[1]) rcs[1] = msm_host_g2_377(b_g2_query + 24, nullptr, assignment, kb, 3, acc2);      // likewise elements of G2 or the identity
  };
  {
    std::thread t0(run1, 0, acc1[0], a_query + 12, assignment, ka), t1(run2), t2(run1, 2, acc1[1], l_query, aux, kl);
    run1(3, acc1[2], h_query, h, kh);
    t0.join(); t1.join(); t2.join();
  }
  for (int r : rcs) if (r) return r;
  uint64_t t1[3][18], t2[3][36];
  auto g1_as_jac = [](const uint64_t* xy, uint64_t* j) {
    memcpy(j, xy, 96); Fq377d::one().to_ark(j + 12);
    if (is_ark_zero(xy, j + 12, 6, 6)) memset(j + 12, 0, 48);
  };
  auto g2_as_jac = [](const uint64_t* xy, uint64_t* j) {
    memcpy(j, xy, 192); Fq377d::one().to_ark(j + 24); Fq377d::zero().to_ark(j + 30);
    if (is_ark_zero(xy, j + 24, 12, 6)) memset(j + 24, 0, 96);
  };
  g1_as_jac(a_query, t1[0]);
  memcpy(t1[1], acc1[0], 144);
  g1_as_jac(alpha_g1, t1[2]);
  if (int rc = sum_jac_g1_377(&t1[0][0], 3, out_a)) return rc;
  g2_as_jac(b_g2_query, t2[0]);
  memcpy(t2[1], acc2, 288);
  g2_as_jac(beta_g2, t2[2]);
  if (int rc = sum_jac_g2_377(&t2[0][0], 3, out_b)) return rc;
  memcpy(t1[0], acc1[1], 144); memcpy(t1[1], acc1[2], 144);
  return sum_jac_g1_377(&t1[0][0], 2, out_c);
}

// ---- the same two compositions against a LOADED proving key: the queries' fixed-base tables are built once (msm.h FixedTable) and every
// proof is four msm_*_fixed calls - the reference creates its Parameters once (crates/epoch-snark/src/api/setup.rs:63-105) and hands the
// same ones to every create_proof_no_zk (prover.rs:78,112).  CURVE: 0 = BW6-761 (all coordinates 12 u64), 1 = BLS12-377 (G1 6, G2 12).
struct ProvingKey {
  int curve = 0;
  FixedTable *a = nullptr, *b = nullptr, *l = nullptr, *h = nullptr;      // over a_query[1..], b_g2_query[1..], l_query, h_query
  size_t na = 0, nb = 0, nl = 0, nh = 0;
  std::vector<uint64_t> a0, b0, alpha, beta;                              // query[0] and the key elements, affine arkworks limbs
  int device = 0;
};
static std::vector<uint8_t> ark_zero_flags(const uint64_t* q, size_t n, int coord64, int one64, const uint64_t* one) {
  std::vector<uint8_t> f(n);
  for (size_t i = 0; i < n; i++) f[i] = is_ark_zero(q + i * 2 * coord64, one, coord64, one64) ? 1 : 0;
  return f;
}
void groth16_key_free(ProvingKey* k) {
  if (!k) return;
  for (FixedTable* t : {k->a, k->b, k->l, k->h}) if (t) (void)fixed_table_release(t);
  delete k;
}
int groth16_key_load(int curve, const uint64_t* a_query, size_t na, const uint64_t* b_g2_query, size_t nb, const uint64_t* h_query, size_t nh, const uint64_t* l_query,
                     size_t nl, const uint64_t* alpha_g1, const uint64_t* beta_g2, int window_bits, ProvingKey** out) {
  if (int rc0 = api_enter()) return rc0;
  if (!a_query || !b_g2_query || !alpha_g1 || !beta_g2 || !out || na == 0 || nb == 0 || (nh && !h_query) || (nl && !l_query) || curve < 0 || curve > 1) return 2;
  const int g1c = curve ? 6 : 12, g2c = 12;                // u64 per coordinate of a G1 / G2 point
  uint64_t one[12];
  memset(one, 0, sizeof one);
  if (curve) Fq377d::one().to_ark(one); else Fq761d::one().to_ark(one);
  const int one64 = curve ? 6 : 12;
  ProvingKey* k = new ProvingKey();
  k->curve = curve; k->na = na; k->nb = nb; k->nl = nl; k->nh = nh; k->device = api_device();
  k->a0.assign(a_query, a_query + 2 * g1c); k->b0.assign(b_g2_query, b_g2_query + 2 * g2c);
  k->alpha.assign(alpha_g1, alpha_g1 + 2 * g1c); k->beta.assign(beta_g2, beta_g2 + 2 * g2c);
  auto build1 = [&](const uint64_t* q, size_t n, FixedTable** t) -> int {        // a G1 query
    if (n == 0) return 0;
    const std::vector<uint8_t> f = ark_zero_flags(q, n, g1c, one64, one);
    return curve ? msm_fixed_build_g1_377(q, f.data(), n, 0, window_bits, t) : msm_fixed_build_761(q, f.data(), n, 0, window_bits, t);
  };
  int rc = build1(a_query + 2 * g1c, na - 1, &k->a);
  if (!rc && nb > 1) {
    const std::vector<uint8_t> f = ark_zero_flags(b_g2_query + 2 * g2c, nb - 1, g2c, one64, one);
    rc = curve ? msm_fixed_build_g2_377(b_g2_query + 2 * g2c, f.data(), nb - 1, 0, window_bits, &k->b) : msm_fixed_build_761(b_g2_query + 2 * g2c, f.data(), nb - 1, 0, window_bits, &k->b);
  }
  if (!rc) rc = build1(l_query, nl, &k->l);
  if (!rc) rc = build1(h_query, nh, &k->h);
  if (rc) { groth16_key_free(k); return rc; }
  *out = k;
  return 0;
}
int groth16_prove_keyed(const ProvingKey* k, const uint64_t* assignment, size_t n_assign, size_t n_aux, const uint64_t* h, size_t n_h, uint64_t* out_a, uint64_t* out_b,
                        uint64_t* out_c) {
  if (int rc0 = api_enter()) return rc0;
  if (!k || !out_a || !out_b || !out_c || (n_assign && !assignment) || n_aux > n_assign || (n_h && !h)) return 2;
  if (k->device != api_device()) return 101;
  const int curve = k->curve, sw = curve ? 4 : 6, g1c = curve ? 6 : 12, J1 = 3 * g1c, J2 = 36;
  const size_t ka = (k->na - 1 < n_assign) ? k->na - 1 : n_assign, kb = (k->nb - 1 < n_assign) ? k->nb - 1 : n_assign;
  const size_t kl = k->nl < n_aux ? k->nl : n_aux, kh = k->nh < n_h ? k->nh : n_h;
  const uint64_t* aux = assignment + (n_assign - n_aux) * sw;
  uint64_t acc[4][36];
  int rcs[4] = {0, 0, 0, 0};
  const int dev = api_device();
  auto ident = [&](uint64_t* o, int words) { memset(o, 0, words * 8); if (curve) Fq377d::one().to_ark(o + words / 3); else Fq761d::one().to_ark(o + words / 3); };
  auto run = [&](int i, const FixedTable* t, bool g2, const uint64_t* sc, size_t n) {
    rcs[i] = api_bind_thread(dev);
    if (rcs[i]) return;
    if (!t || n == 0) { ident(acc[i], g2 ? J2 : J1); return; }
    if (!curve) rcs[i] = msm_fixed_run_761(t, sc, n, 0, acc[i], nullptr);
    else rcs[i] = g2 ? msm_fixed_run_g2_377(t, sc, n, 0, acc[i], nullptr) : msm_fixed_run_g1_377(t, sc, n, 0, acc[i], nullptr);
  };
  {
    std::thread t0(run, 0, k->a, false, assignment, ka), t1(run, 1, k->b, true, assignment, kb), t2(run, 2, k->l, false, aux, kl);
    run(3, k->h, false, h, kh);
    t0.join(); t1.join(); t2.join();
  }
  for (int r : rcs) if (r) return r;
  // affine (x, y) -> Jacobian (x, y, 1), or Z = 0 for arkworks' identity encoding
  auto as_jac = [&](const uint64_t* xy, int c64, uint64_t* j) {
    memcpy(j, xy, 2 * c64 * 8);
    memset(j + 2 * c64, 0, c64 * 8);
    uint64_t one[12];
    if (curve) Fq377d::one().to_ark(one); else Fq761d::one().to_ark(one);
    const int one64 = curve ? 6 : 12;
    if (!is_ark_zero(xy, one, c64, one64)) memcpy(j + 2 * c64, one, one64 * 8);
  };
  uint64_t t1[3][36], t2[3][36];
  as_jac(k->a0.data(), g1c, t1[0]); memcpy(t1[1], acc[0], J1 * 8); as_jac(k->alpha.data(), g1c, t1[2]);
  uint64_t pack1[3 * 36];
  for (int q = 0; q < 3; q++) memcpy(pack1 + q * J1, t1[q], J1 * 8);
  if (int rc = curve ? sum_jac_g1_377(pack1, 3, out_a) : sum_jac_761(pack1, 3, out_a)) return rc;
  as_jac(k->b0.data(), 12, t2[0]); memcpy(t2[1], acc[1], J2 * 8); as_jac(k->beta.data(), 12, t2[2]);
  if (int rc = curve ? sum_jac_g2_377(&t2[0][0], 3, out_b) : sum_jac_761(&t2[0][0], 3, out_b)) return rc;
  memcpy(pack1, acc[2], J1 * 8); memcpy(pack1 + J1, acc[3], J1 * 8);
  return curve ? sum_jac_g1_377(pack1, 2, out_c) : sum_jac_761(pack1, 2, out_c);
}
}  // namespace celo
