// Translation unit: Batch::verify for many batches with everything between the two MSMs and the pairing check kept on the
// device (crates/bls-crypto/src/bls/batch.rs:44-84, looped by crates/bls-snark-sys/src/signatures.rs:343-400).
//
//   G2 batch MSM  (sum_j e_bj pk_bj  per batch b)   \  two engines, two streams, in flight together
//   G1 batch MSM  (sum_j e_bj sig_bj per batch b)   /
//   k_pack_verify_pairs   one lane per (batch, group): Jacobian -> affine (one inversion) written straight into the pairing
//                         engine's input slots as the pairs (S_b, -g2), (H(m_b), P_b)
//   pairing engine        m two-pair products -> m verdicts
// The reference does this per batch on the CPU: two MSMs, batch_normalization, product_of_pairings == 1.
#include "msm.h"
#include "runtime.h"
#include <thread>

namespace celo {
int msm_batch_begin_g1_377(const void*, const void*, const void*, int, const uint32_t*, size_t, int, BatchRun*);
int msm_batch_begin_g2_377(const void*, const void*, const void*, int, const uint32_t*, size_t, int, BatchRun*);
void msm_batch_end_g1_377(BatchRun*, int);
void msm_batch_end_g2_377(BatchRun*, int);
int pairing_stage_377(uint32_t, size_t, PairingStage*);
int pairing_run_staged_377(PairingStage*, const uint32_t*, size_t, uint8_t*);

struct NegG2 { uint64_t xy[24]; };   // -g2 generator, affine, arkworks Montgomery limbs

// lanes [0, m): G1 half of batch b = lane; lanes [m_pad, m_pad + m): G2 half (m_pad = m rounded up to the wave size, so a wave
// never mixes the two fields)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_pack_verify_pairs(const uint64_t* __restrict__ sig_sum /* m x 18 */, const uint64_t* __restrict__ pk_sum /* m x 36 */,
                    const uint64_t* __restrict__ hash_xy /* m x 12 */, const uint8_t* __restrict__ hash_inf, NegG2 ng2, uint32_t m, uint32_t m_pad,
                    uint64_t* __restrict__ g1 /* 2m x 12 */, uint64_t* __restrict__ g2 /* 2m x 24 */, uint8_t* __restrict__ i1, uint8_t* __restrict__ i2) {
  typedef Fp<P377> Fq;
  typedef Fp2<P377> Fq2;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) {
    const uint32_t b = t;
    const uint64_t* s = sig_sum + (size_t)b * 18;
    uint64_t* o = g1 + (size_t)(2 * b) * 12;
    const Fq Z = Fq::norm(Fq::from_ark(s + 12));
    const bool id = Z.is_zero_mod_p();
    i1[2 * b] = id ? 1 : 0;
    if (id) { for (int q = 0; q < 12; q++) o[q] = 0; }
    else {
      const Fq zi = Fq::norm(Fq::inv(Z)), zi2 = Fq::norm(Fq::sqr(zi));
      Fq::mul(Fq::from_ark(s), zi2).to_ark(o);
      Fq::mul(Fq::from_ark(s + 6), Fq::norm(Fq::mul(zi2, zi))).to_ark(o + 6);
    }
    uint64_t* oh = g1 + (size_t)(2 * b + 1) * 12;
    for (int q = 0; q < 12; q++) oh[q] = hash_xy[(size_t)b * 12 + q];
    i1[2 * b + 1] = hash_inf ? hash_inf[b] : 0;
  } else if (t >= m_pad && t - m_pad < m) {
    const uint32_t b = t - m_pad;
    const uint64_t* s = pk_sum + (size_t)b * 36;
    uint64_t* on = g2 + (size_t)(2 * b) * 24;
    for (int q = 0; q < 24; q++) on[q] = ng2.xy[q];
    i2[2 * b] = 0;
    uint64_t* o = g2 + (size_t)(2 * b + 1) * 24;
    const Fq2 Z = Fq2::norm(Fq2::from_ark(s + 24));
    const bool id = Z.is_zero_mod_p();
    i2[2 * b + 1] = id ? 1 : 0;
    if (id) { for (int q = 0; q < 24; q++) o[q] = 0; }
    else {
      const Fq2 zi = Fq2::norm(Fq2::inv(Z)), zi2 = Fq2::norm(Fq2::sqr(zi));
      Fq2::mul(Fq2::from_ark(s), zi2).to_ark(o);
      Fq2::mul(Fq2::from_ark(s + 12), Fq2::norm(Fq2::mul(zi2, zi))).to_ark(o + 12);
    }
  }
}

// The chain in three steps, so that a caller whose inputs arrive in stages (Seam A gathers 10^6 key handles, then 10^6 signature
// handles) can start the longest leg - the G2 MSM - as soon as ITS inputs are there:
//   bv_begin_keys   enqueue the G2 batch MSM        bv_begin_sigs   enqueue the G1 batch MSM        bv_finish   pairs + products
// resident = 1: DEVICE pointers; 0: host pointers (staged by the engines).  offsets, out_ok: host.  *_inf: optional byte-per-point
// "is the identity" arrays (host or device like the points).  bv_finish releases everything, also after a failed begin.
struct BvJob { BatchRun keys, sigs; };
int bv_begin_keys(BvJob* j, const void* pk_xy, const void* pk_inf, const void* exponents, int resident, const uint32_t* offsets, size_t m) {
  if (int rc = api_enter()) return rc;
  // the keys of Batch::verify are PublicKey values: elements of the prime-order subgroup G2 by construction (checked deserialisation,
  // secret keys, sums) - what lets the batched MSM split their exponents with the endomorphism psi (msm.h, k_gls_expand)
  return msm_batch_begin_g2_377(pk_xy, pk_inf, exponents, resident, offsets, m, 1, &j->keys);
}
int bv_begin_sigs(BvJob* j, const void* sig_xy, const void* sig_inf, const void* exponents, int resident, const uint32_t* offsets, size_t m) {
  if (int rc = api_enter()) return rc;
  return msm_batch_begin_g1_377(sig_xy, sig_inf, exponents, resident, offsets, m, 0, &j->sigs);
}
int bv_finish(BvJob* j, int begun_ok, const void* hash_xy, const void* hash_inf, int resident, const uint64_t neg_g2_xy[24], size_t m, uint8_t* out_ok) {
  BatchRun& r1 = j->sigs;
  BatchRun& r2 = j->keys;
  PairingStage ps;
  int rc = begun_ok ? 0 : 1;
  uint64_t* d_hash = nullptr;
  uint8_t* d_hinf = nullptr;
  hipEvent_t e1 = nullptr, e2 = nullptr;
  bool drained = false;
  std::vector<uint32_t> po;
  NegG2 ng2;
  for (int q = 0; q < 24; q++) ng2.xy[q] = neg_g2_xy[q];
  const uint32_t m_pad = ((uint32_t)m + 63u) & ~63u;
  if (rc || !r1.lease || !r2.lease) { rc = rc ? rc : 1; goto done; }
  if ((rc = pairing_stage_377((uint32_t)(2 * m), m, &ps))) goto done;
  if (hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) { rc = 1; goto done; }
  if (!resident) {   // the message hashes: a small upload of our own
    if (hipMalloc(&d_hash, m * 96) != hipSuccess) { rc = 1; goto done; }
    if (hipMemcpyAsync(d_hash, hash_xy, m * 96, hipMemcpyHostToDevice, ps.stream) != hipSuccess) { rc = 1; goto done; }
    if (hash_inf) {
      if (hipMalloc(&d_hinf, m) != hipSuccess) { rc = 1; goto done; }
      if (hipMemcpyAsync(d_hinf, hash_inf, m, hipMemcpyHostToDevice, ps.stream) != hipSuccess) { rc = 1; goto done; }
    }
  }
  if (hipEventRecord(e1, r1.stream) != hipSuccess || hipEventRecord(e2, r2.stream) != hipSuccess ||
      hipStreamWaitEvent(ps.stream, e1, 0) != hipSuccess || hipStreamWaitEvent(ps.stream, e2, 0) != hipSuccess) { rc = 1; goto done; }
  hipLaunchKernelGGL(k_pack_verify_pairs, dim3((2 * m_pad + 63) / 64), dim3(64), 0, ps.stream, r1.d_out, r2.d_out,
                     resident ? (const uint64_t*)hash_xy : d_hash, resident ? (const uint8_t*)hash_inf : d_hinf, ng2, (uint32_t)m, m_pad, ps.d_g1, ps.d_g2, ps.d_i1, ps.d_i2);
  po.resize(m + 1);
  for (size_t b = 0; b <= m; b++) po[b] = (uint32_t)(2 * b);
  rc = pairing_run_staged_377(&ps, po.data(), m, out_ok);     // synchronises ps.stream: both MSM streams have drained by then
  drained = rc == 0;
done:
  if (ps.lease) (void)pairing_run_staged_377(&ps, nullptr, 0, nullptr);
  if (!drained) {   // error path: let the MSM streams finish before their engines go back
    if (r1.stream) (void)hipStreamSynchronize(r1.stream);
    if (r2.stream) (void)hipStreamSynchronize(r2.stream);
  }
  msm_batch_end_g1_377(&r1, drained ? 1 : 0);
  msm_batch_end_g2_377(&r2, drained ? 1 : 0);
  if (e1) (void)hipEventDestroy(e1);
  if (e2) (void)hipEventDestroy(e2);
  if (d_hash) (void)hipFree(d_hash);
  if (d_hinf) (void)hipFree(d_hinf);
  return rc;
}
int batch_verify_377_run(const void* pk_xy, const void* pk_inf, const void* sig_xy, const void* sig_inf, const void* exponents, int resident,
                         const uint32_t* offsets, const void* hash_xy, const void* hash_inf, const uint64_t neg_g2_xy[24], size_t m, uint8_t* out_ok) {
  if (int rc = api_enter()) return rc;
  if (m == 0) return 0;
  if (!pk_xy || !sig_xy || !exponents || !offsets || !hash_xy || !neg_g2_xy || !out_ok || m > 0x3fffffffu) return 2;
  BvJob job;
  int rc1 = 0, rc2 = 0;
  const int dev = api_device();
  {
    // each begin has one host round trip (the exponents' bit length sizes the window count): two threads keep both in flight.  The
    // key leg is the long one (G2 arithmetic) and goes first; the signature leg's latency-bound tail then hides under its accumulation.
    rc2 = bv_begin_keys(&job, pk_xy, pk_inf, exponents, resident, offsets, m);
    rc1 = bv_begin_sigs(&job, sig_xy, sig_inf, exponents, resident, offsets, m);
    (void)dev;
  }
  const int rc = bv_finish(&job, !rc1 && !rc2, hash_xy, hash_inf, resident, neg_g2_xy, m, out_ok);
  return rc1 ? rc1 : rc2 ? rc2 : rc;
}
}  // namespace celo
