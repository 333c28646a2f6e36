// Translation unit: the Groth16 prover's device work for BW6-761 - what
// ark_groth16::create_proof_no_zk does after R1CS synthesis, as called at crates/epoch-snark/src/api/prover.rs:78,112:
//   1. witness map (ark-groth16 r1cs_to_qap.rs R1CStoQAP::witness_map): the QAP evaluations a, b, c over the domain
//        ifft(a), ifft(b), ifft(c); coset_fft(a), coset_fft(b), coset_fft(c); ab = (a o b - c) / Z(coset); coset_ifft(ab) = h
//      seven radix-2 transforms over Fr(BW6-761) (ntt.h) and one pointwise kernel; Z is constant on the coset: g^n - 1.
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
constexpr int NTT_WORDS = 16;    // device form: 14 limbs padded to 16 words
int ntt_run(uint64_t*, unsigned, const uint64_t*, const uint64_t*, int, const uint64_t*, int, void*);
int msm_host_761(const uint64_t*, const uint8_t*, const uint64_t*, size_t, uint64_t*);
int sum_jac_761(const uint64_t*, size_t, uint64_t*);

// a[i] <- (a[i] b[i] - c[i]) z   (arkworks Montgomery limbs in and out; optionally the canonical integer: Fr::into_repr())
__global__ void __launch_bounds__(256) k_qap_combine(uint64_t* __restrict__ a, const uint64_t* __restrict__ b, const uint64_t* __restrict__ c, uint32_t n,
                                                     const uint32_t* __restrict__ z_dev) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr761 x = Fr761::from_ark(a + (size_t)i * 6), y = Fr761::from_ark(b + (size_t)i * 6), w = Fr761::from_ark(c + (size_t)i * 6);
  const Fr761 z = Fr761::load(z_dev);
  const Fr761 t = Fr761::norm(Fr761::sub<4, 1>(Fr761::mul(x, y), w));
  Fr761::mul(t, z).to_ark(a + (size_t)i * 6);
}
__global__ void __launch_bounds__(256) k_to_canonical(uint64_t* __restrict__ a, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr761::from_ark(a + (size_t)i * 6).to_canonical(a + (size_t)i * 6);
}

#define PRV_OK(x)                                                                                                  \
  do {                                                                                                             \
    hipError_t e_ = (x);                                                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "[celo-amd] %s: %s\n", #x, hipGetErrorString(e_)); rc = 10; goto done; } \
  } while (0)

// dev = 1: a, b, c are DEVICE pointers (a is overwritten with h; b and c are overwritten with intermediate values).
int witness_map_run(uint64_t* a, uint64_t* b, uint64_t* c, unsigned log_n, const uint64_t* omega, const uint64_t* omega_inv, const uint64_t* coset,
                    const uint64_t* coset_inv, const uint64_t* n_inv, const uint64_t* z_inv, int out_canonical, int dev, void* stream_) {
  if (int rc0 = api_enter()) return rc0;
  if (!a || !b || !c || !omega || !omega_inv || !coset || !coset_inv || !n_inv || !z_inv || log_n > 28) return 2;
  const size_t n = size_t(1) << log_n, bytes = n * 48;
  hipStream_t stream = (hipStream_t)stream_;
  uint64_t *da = a, *db = b, *dc = c;
  uint32_t* d_z = nullptr;
  int rc = 0;
  uint32_t zw[NTT_WORDS];
  {
    Fr761 v = Fr761::wred(Fr761::from_ark(z_inv));
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
  for (uint64_t* p : {da, db, dc}) if ((rc = ntt_run(p, log_n, omega_inv, nullptr, 0, n_inv, 1, stream))) goto done;        // ifft
  for (uint64_t* p : {da, db, dc}) if ((rc = ntt_run(p, log_n, omega, coset, 0, nullptr, 1, stream))) goto done;             // coset_fft
  hipLaunchKernelGGL(k_qap_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, da, db, dc, (uint32_t)n, d_z);
  if ((rc = ntt_run(da, log_n, omega_inv, coset_inv, 1, n_inv, 1, stream))) goto done;                                          // coset_ifft
  if (out_canonical) hipLaunchKernelGGL(k_to_canonical, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, da, (uint32_t)n);
  PRV_OK(hipGetLastError());
  if (!dev) PRV_OK(hipMemcpyAsync(a, da, bytes, hipMemcpyDeviceToHost, stream));
  PRV_OK(hipStreamSynchronize(stream));
done:
  if (!dev) { if (da) (void)hipFree(da); if (db) (void)hipFree(db); if (dc) (void)hipFree(dc); }
  if (d_z) (void)hipFree(d_z);
  return rc;
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
    if (!rcs[i]) rcs[i] = msm_host_761(bases, nullptr, sc, k, acc[i]);
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
  };
  uint64_t terms[3][36];
  affine_as_jac(a_query, terms[0]); memcpy(terms[1], acc[0], 288); affine_as_jac(alpha_g1, terms[2]);
  if (int rc = sum_jac_761(&terms[0][0], 3, out_a)) return rc;
  affine_as_jac(b_g2_query, terms[0]); memcpy(terms[1], acc[1], 288); affine_as_jac(beta_g2, terms[2]);
  if (int rc = sum_jac_761(&terms[0][0], 3, out_b)) return rc;
  memcpy(terms[0], acc[2], 288); memcpy(terms[1], acc[3], 288);
  return sum_jac_761(&terms[0][0], 2, out_c);
}
}  // namespace celo
