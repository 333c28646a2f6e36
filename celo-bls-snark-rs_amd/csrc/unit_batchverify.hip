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

// ---- the per-device mirror of Seam A's handle arenas (seam_a.hip, batch_verify_strict): W u64 of affine coordinates + an identity
// byte per arena slot.  scatter: freshly staged entries go to their slots; gather: the call's slot numbers become the dense point
// arrays the batch MSMs read.  One lane per 64-bit word: both sides of either copy move whole 96 / 192-byte rows.
template <int W>
__global__ void __launch_bounds__(256) k_mirror_scatter(const uint64_t* __restrict__ up_xy, const uint8_t* __restrict__ up_inf, const uint32_t* __restrict__ slots,
                                                        uint64_t* __restrict__ mirror_xy, uint8_t* __restrict__ mirror_inf, uint32_t k) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t e = t / W;
  const uint32_t w = (uint32_t)(t % W);
  if (e >= k) return;
  const uint32_t s = slots[e];
  mirror_xy[(size_t)s * W + w] = up_xy[e * W + w];
  if (w == 0) mirror_inf[s] = up_inf[e];
}
template <int W>
__global__ void __launch_bounds__(256) k_mirror_gather(const uint64_t* __restrict__ mirror_xy, const uint8_t* __restrict__ mirror_inf, const uint32_t* __restrict__ slots,
                                                       uint64_t* __restrict__ out_xy, uint8_t* __restrict__ out_inf, uint32_t n) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t e = t / W;
  const uint32_t w = (uint32_t)(t % W);
  if (e >= n) return;
  const uint32_t s = slots[e];
  out_xy[e * W + w] = mirror_xy[(size_t)s * W + w];
  if (w == 0) out_inf[e] = mirror_inf[s];
}
// words = 24 (G2 keys) or 12 (G1 signatures); everything device memory; enqueued on `stream`
int bv_mirror_scatter(int words, const uint64_t* up_xy, const uint8_t* up_inf, const uint32_t* slots, uint64_t* mirror_xy, uint8_t* mirror_inf, size_t k, hipStream_t stream) {
  if (k == 0) return 0;
  if (k > 0xffffffffu || (words != 24 && words != 12)) return 2;
  const unsigned blocks = (unsigned)((k * (size_t)words + 255) / 256);
  if (words == 24) hipLaunchKernelGGL((k_mirror_scatter<24>), dim3(blocks), dim3(256), 0, stream, up_xy, up_inf, slots, mirror_xy, mirror_inf, (uint32_t)k);
  else hipLaunchKernelGGL((k_mirror_scatter<12>), dim3(blocks), dim3(256), 0, stream, up_xy, up_inf, slots, mirror_xy, mirror_inf, (uint32_t)k);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
int bv_mirror_gather(int words, const uint64_t* mirror_xy, const uint8_t* mirror_inf, const uint32_t* slots, uint64_t* out_xy, uint8_t* out_inf, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  if (n > 0xffffffffu || (words != 24 && words != 12)) return 2;
  const unsigned blocks = (unsigned)((n * (size_t)words + 255) / 256);
  if (words == 24) hipLaunchKernelGGL((k_mirror_gather<24>), dim3(blocks), dim3(256), 0, stream, mirror_xy, mirror_inf, slots, out_xy, out_inf, (uint32_t)n);
  else hipLaunchKernelGGL((k_mirror_gather<12>), dim3(blocks), dim3(256), 0, stream, mirror_xy, mirror_inf, slots, out_xy, out_inf, (uint32_t)n);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- Batch::verify's random exponents drawn ON THE DEVICE (crates/bls-crypto/src/bls/batch.rs:51-58: one exponent of
// byte_count_from_target_batch_size(128, n) = (128 + ceil(log2 n) + 7) / 8 random bytes per signer, from rand::thread_rng(), itself a
// ChaCha stream seeded by the OS).  Here: ChaCha20 (RFC 7539 block function, 64-bit block counter as rand_chacha) under a 256-bit key
// the HOST takes from the OS per call; signer `at` of the call owns block `at` of the stream and keeps its first nbytes bytes,
// little-endian, in a 4 x u64 container.  The call then ships 32 key bytes instead of 32 bytes per signer, and no host core draws them.
__device__ __forceinline__ uint32_t cc_rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
__global__ void __launch_bounds__(256) k_draw_exponents(const uint32_t* __restrict__ offsets, uint32_t m, uint32_t tot, uint32_t k0, uint32_t k1, uint32_t k2,
                                                        uint32_t k3, uint32_t k4, uint32_t k5, uint32_t k6, uint32_t k7, uint64_t* __restrict__ out) {
  const uint32_t at = blockIdx.x * 256 + threadIdx.x;
  if (at >= tot) return;
  uint32_t lo = 0, hi = m;                                    // the batch of signer `at`: the last b with offsets[b] <= at
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (offsets[mid] <= at) lo = mid; else hi = mid; }
  const uint32_t n = offsets[lo + 1] - offsets[lo];
  const uint32_t lg = n <= 1 ? 0 : 32 - __builtin_clz(n - 1);  // ark_std::log2 = ceil(log2 n)
  uint32_t nbytes = (128 + lg + 7) / 8;
  if (nbytes > 31) nbytes = 31;
  const uint32_t s[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u, k0, k1, k2, k3, k4, k5, k6, k7, at, 0u, 0u, 0u};
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = s[i];
#define CC_QR(a, b, c, d) w[a] += w[b]; w[d] = cc_rotl(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = cc_rotl(w[b] ^ w[c], 12); \
                          w[a] += w[b]; w[d] = cc_rotl(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = cc_rotl(w[b] ^ w[c], 7);
#pragma unroll 1
  for (int r = 0; r < 10; r++) {
    CC_QR(0, 4, 8, 12) CC_QR(1, 5, 9, 13) CC_QR(2, 6, 10, 14) CC_QR(3, 7, 11, 15)
    CC_QR(0, 5, 10, 15) CC_QR(1, 6, 11, 12) CC_QR(2, 7, 8, 13) CC_QR(3, 4, 9, 14)
  }
#undef CC_QR
  uint32_t v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t x = w[i] + s[i];
    const uint32_t keep = nbytes > 4u * i ? nbytes - 4u * i : 0u;      // bytes of word i that belong to the exponent
    if (keep < 4) x = keep ? x & ((1u << (8 * keep)) - 1u) : 0u;
    v[i] = x;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[(size_t)at * 4 + i] = (uint64_t)v[2 * i] | ((uint64_t)v[2 * i + 1] << 32);
}
// d_offsets: m + 1 device words; d_out: tot x 4 u64 on the device; enqueued on `stream`
int bv_draw_exponents(const uint32_t key[8], const uint32_t* d_offsets, size_t m, size_t tot, uint64_t* d_out, hipStream_t stream) {
  if (tot == 0) return 0;
  if (tot > 0xffffffffu || m == 0 || m > 0x7fffffffu) return 2;
  hipLaunchKernelGGL(k_draw_exponents, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, d_offsets, (uint32_t)m, (uint32_t)tot, key[0], key[1], key[2], key[3],
                     key[4], key[5], key[6], key[7], d_out);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
// test / tooling entry (include/celo_bls_amd.h): the exponents of one call, returned to the host
int draw_exponents_run(const uint32_t key[8], const uint32_t* offsets, size_t m, uint64_t* out) {
  if (int rc = api_enter()) return rc;
  if (!key || !offsets || !out || m == 0 || m > 0x7fffffffu) return 2;
  const size_t tot = offsets[m];
  if (tot == 0) return 0;
  uint32_t* d_off = nullptr; uint64_t* d_out = nullptr;
  int rc = 1;
  if (hipMalloc((void**)&d_off, (m + 1) * 4) == hipSuccess && hipMalloc((void**)&d_out, tot * 32) == hipSuccess &&
      hipMemcpy(d_off, offsets, (m + 1) * 4, hipMemcpyHostToDevice) == hipSuccess && (rc = bv_draw_exponents(key, d_off, m, tot, d_out, nullptr)) == 0)
    rc = hipMemcpy(out, d_out, tot * 32, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
  if (d_off) (void)hipFree(d_off);
  if (d_out) (void)hipFree(d_out);
  return rc;
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
  j->sigs.bits = j->keys.lease ? j->keys.bits : 0;     // the same exponents: the key leg has measured their length already
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
