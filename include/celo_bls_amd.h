/* celo_bls_amd.h — C ABI of the MI355X (gfx950) MSM / pairing / NTT hot path.
 *
 * "Seam B" of SURVEY.md §8b: the thin extern "C" shim that replaces the arkworks call sites
 * of celo-bls-snark-rs.  Each entry cites the reference interface it stands in for.
 *
 * Conventions (identical to what a Rust shim over arkworks 0.1 types would hand over):
 *   - limbs little-endian u64; base-field elements in arkworks Montgomery form
 *     (R = 2^384 for BLS12-377 Fq, 2^768 for BW6-761 Fq); Fq2 = c0 || c1;
 *   - scalars canonical (Fr::into_repr()): 4 x u64 (BLS12-377 Fr) / 6 x u64 (BW6-761 Fr), must be < r;
 *   - affine bases x || y, with an optional byte-per-point infinity array (NULL = none);
 *   - results: Jacobian (X, Y, Z) in Montgomery form, identity encoded Z = 0 — the in-memory
 *     layout of GroupProjective{x,y,z};
 *   - return 0 = ok, non-zero = device/runtime error (a caller maps it to `false` like
 *     crates/bls-snark-sys/src/lib.rs:21-27 convert_result_to_bool);
 *   - *_dev variants take DEVICE pointers (inputs already resident in HBM) and a hipStream_t
 *     (as void*; NULL = default stream); the result still lands in host memory.
 * Threading (the reference's callers are synchronous, re-entrant and multi-threaded -
 * crates/bls-snark-sys/src/cache.rs:5, signatures.rs:343-400): every function is synchronous and may be called from any host
 * thread at any time.  There is no global lock: each MSM / pairing / NTT call checks an engine instance (workspace, pinned
 * staging, its own non-blocking HIP stream) out of a per-device pool, so independent calls overlap on the GPU; single
 * pairing-product checks from concurrent threads are combined into shared launches.  Bulk decode / hash calls are serialised
 * among themselves only.  Several devices can be driven from one process: celo_amd_use_device() binds the calling thread to a
 * device, the msm_*_multi entry points shard one MSM over a device list internally.
 */
#ifndef CELO_BLS_AMD_H
#define CELO_BLS_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Selects the process's DEFAULT device: the one every host thread uses unless it bound itself to another with
 * celo_amd_use_device (HIP's current device is per thread; each entry point re-applies the right one on its calling thread).
 * Returns 0, or non-zero if there is no gfx950 device (100) or `device` is out of range (101).  Never calling it = device 0. */
int celo_amd_init(int device);
/* Binds the CALLING host thread to `device` for all its later calls (engines, tables and streams are per device). */
int celo_amd_use_device(int device);
/* Page-locked host memory for the arrays a caller hands to the host-pointer entry points (msm_<group>, msm_batch_<group>, ...): the
 * transfers of such a buffer run at the link's rate from the first call on and do not hold the calling thread (a pageable range is
 * pinned by the HIP runtime the first time it is copied from: DESIGN.md section 4 "Buffers the runtime has never seen" - 2^20 G1 terms
 * 4.6-4.9 ms per call from newly allocated vectors, 4.0-4.2 ms from reused or page-locked ones).  A wrapper that converts the caller's
 * points into limbs anyway writes them here.  Free with celo_amd_host_free; 0 on success. */
int celo_amd_host_alloc(size_t bytes, void** out);
int celo_amd_host_free(void* p);
/* Number of HIP devices visible to the process (at most 16 are used). */
int celo_amd_device_count(int* count);
/* "gfx950:..." string of the active device into buf; returns 0. */
int celo_amd_device_name(char* buf, size_t buflen);

/* ---- MSM.  Replaces VariableBaseMSM::multi_scalar_mul(&[GAffine], &[BigInt]) at
 *   crates/bls-crypto/src/bls/signature.rs:85  (G1, Signature::batch)
 *   crates/bls-crypto/src/bls/public.rs:61     (G2, PublicKey::batch)
 *   ark_groth16::create_proof_no_zk via crates/epoch-snark/src/api/prover.rs:78,112 (BW6-761 / BLS12-377 prover MSMs) */
int msm_bls12_377_g1(const uint64_t* bases_xy /* n*12 */, const uint8_t* inf, const uint64_t* scalars /* n*4 */, size_t n,
                     uint64_t out_xyz[18]);
int msm_bls12_377_g2(const uint64_t* bases_xy /* n*24 */, const uint8_t* inf, const uint64_t* scalars /* n*4 */, size_t n,
                     uint64_t out_xyz[36]);
int msm_bw6_761_g1(const uint64_t* bases_xy /* n*24 */, const uint8_t* inf, const uint64_t* scalars /* n*6 */, size_t n,
                   uint64_t out_xyz[36]);
int msm_bw6_761_g2(const uint64_t* bases_xy /* n*24 */, const uint8_t* inf, const uint64_t* scalars /* n*6 */, size_t n,
                   uint64_t out_xyz[36]);
int msm_bls12_377_g1_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, uint64_t out_xyz[18], void* stream);
/* The G1 MSM for bases the caller vouches to be elements of the prime-order subgroup G1 - what Signature::batch hands over
 * (crates/bls-crypto/src/bls/signature.rs:70-89: a Signature of the reference is one by construction - checked deserialisation
 * signature.rs:31-57, sign, sums) and what a Groth16 proving key holds.  Same result as msm_bls12_377_g1; the library may split every
 * scalar with the endomorphism phi(x, y) = (beta x, y) = -[x^2](x, y): n terms of 253 bits become 2 n terms of 127 bits - half the
 * windows, half the bucket reduction, half the final Horner chain (csrc/msm.h k_glv_expand, gls.h).  For a base outside the subgroup
 * the result is unspecified (use msm_bls12_377_g1, which is VariableBaseMSM on any curve point). */
int msm_bls12_377_g1_subgroup(const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[18]);
int msm_bls12_377_g1_subgroup_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, uint64_t out_xyz[18], void* hip_stream);
/* Likewise for G2 and bases that are elements of the prime-order subgroup G2 - what PublicKey::batch hands over
 * (crates/bls-crypto/src/bls/public.rs:47-65: a PublicKey is one by construction): k = k0 + k1 x^2, [x^2]P = psi^2(P). */
int msm_bls12_377_g2_subgroup(const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int msm_bls12_377_g2_subgroup_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, uint64_t out_xyz[36], void* hip_stream);
int msm_bls12_377_g2_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, uint64_t out_xyz[36], void* stream);
int msm_bw6_761_g1_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, uint64_t out_xyz[36], void* stream);
int msm_bw6_761_g2_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, uint64_t out_xyz[36], void* stream);

/* ---- one MSM sharded over several devices of this process (index-range shards, no data-path collective;
 * the reference's callers are ONE process - crates/bls-snark-sys/src/signatures.rs:343, crates/epoch-snark/src/api/prover.rs:78).
 * devices[ndev]: device ordinals (a device may appear more than once: that many engines run on it concurrently); one host
 * thread per entry computes the partial sum of its contiguous slice on its device, the Jacobian partials (144 / 288 bytes
 * each) are folded on the host.  Same conventions and results as the single-device entry points.
 * _multi: HOST pointers, n terms cut into ndev equal index ranges.
 * _multi_dev: per-shard DEVICE pointers (shard d resident on devices[d]: d_bases[d], d_inf[d] or d_inf == NULL, d_scalars[d],
 * n_per[d] terms). */
int msm_bls12_377_g1_multi(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[18]);
int msm_bls12_377_g2_multi(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int msm_bw6_761_g1_multi(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int msm_bw6_761_g2_multi(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int msm_bls12_377_g1_multi_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, const size_t* n_per, uint64_t out_xyz[18]);
int msm_bls12_377_g2_multi_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, const size_t* n_per, uint64_t out_xyz[36]);
int msm_bw6_761_g1_multi_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, const size_t* n_per, uint64_t out_xyz[36]);
int msm_bw6_761_g2_multi_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, const size_t* n_per, uint64_t out_xyz[36]);

/* ---- the same MSM partitioned by WINDOW instead of by index range ("alternative partitioning"): every listed
 * device holds ALL n bases and scalars and owns a contiguous range of the Pippenger windows - 1/ndev of the bucket additions, only its
 * windows' buckets to reduce, only its share of the final Horner chain; the ndev partial sums are joined on the host with the doublings
 * between the ranges (total = sum_g 2^bit(g) P_g).  This is the form that scales ONE MSM of up to ~2^21 terms over the GPUs of a node
 * (an index-range shard of 2^20 / 8 terms is bound by the pipeline's fixed latencies); it costs ndev-fold base memory, so the prover's
 * 2^24-term MSMs keep the index-range form above.  Same conventions and results as the single-device entry points; the _subgroup
 * forms take bases the caller vouches to be in the prime-order subgroup (see msm_bls12_377_g1_subgroup).
 * _multi_windows: HOST pointers (each shard stages the whole input onto its device).
 * _multi_windows_dev: d_bases[d] / d_inf[d] (or d_inf == NULL) / d_scalars[d] = the replica of the n terms resident on devices[d]. */
int msm_bls12_377_g1_multi_windows(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[18]);
int msm_bls12_377_g1_subgroup_multi_windows(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[18]);
int msm_bls12_377_g2_multi_windows(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int msm_bls12_377_g2_subgroup_multi_windows(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int msm_bw6_761_g1_multi_windows(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int msm_bw6_761_g2_multi_windows(const int* devices, int ndev, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t out_xyz[36]);
int msm_bls12_377_g1_multi_windows_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, size_t n, uint64_t out_xyz[18]);
int msm_bls12_377_g1_subgroup_multi_windows_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, size_t n, uint64_t out_xyz[18]);
int msm_bls12_377_g2_multi_windows_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, size_t n, uint64_t out_xyz[36]);
int msm_bls12_377_g2_subgroup_multi_windows_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, size_t n, uint64_t out_xyz[36]);
int msm_bw6_761_g1_multi_windows_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, size_t n, uint64_t out_xyz[36]);
int msm_bw6_761_g2_multi_windows_dev(const int* devices, int ndev, const void* const* d_bases, const void* const* d_inf, const void* const* d_scalars, size_t n, uint64_t out_xyz[36]);
/* One rank's share of a window-partitioned MSM when the GPUs belong to DIFFERENT processes (one process per GPU, bench.py under
 * torch.distributed.run): the partial sum over the windows shard `shard` of `nshards` owns, as X || Y || ZZ || ZZZ (arkworks limbs:
 * 4 x 6 u64 for G1 of BLS12-377, 4 x 12 u64 for the other groups; ZZ = 0: the identity), and *bit_lo = the first scalar bit of the
 * shard's range.  The ranks exchange the fixed-size records (one all-gather) and every rank joins them with msm_*_join_windows. */
int msm_bls12_377_g1_window_shard_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, int subgroup, int shard, int nshards,
                                      uint64_t out_xyzz[24], int* bit_lo, void* hip_stream);
int msm_bls12_377_g2_window_shard_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, int subgroup, int shard, int nshards,
                                      uint64_t out_xyzz[48], int* bit_lo, void* hip_stream);
int msm_bw6_761_window_shard_dev(const void* d_bases_xy, const void* d_inf, const void* d_scalars, size_t n, int shard, int nshards,
                                 uint64_t out_xyzz[48], int* bit_lo, void* hip_stream);
/* total = sum_g 2^bit_lo[g] P_g over nshards records in shard order (bit_lo ascending); host-side, 64-bit limbs. */
int msm_bls12_377_g1_join_windows(const uint64_t* xyzz /* nshards x 24 */, const int* bit_lo, int nshards, uint64_t out_xyz[18]);
int msm_bls12_377_g2_join_windows(const uint64_t* xyzz /* nshards x 48 */, const int* bit_lo, int nshards, uint64_t out_xyz[36]);
int msm_bw6_761_join_windows(const uint64_t* xyzz /* nshards x 48 */, const int* bit_lo, int nshards, uint64_t out_xyz[36]);

/* ---- FIXED-BASE MSM: per-key tables for bases that stay while the scalars change - the Groth16 prover's queries
 * (crates/epoch-snark/src/api/prover.rs:78,112 hand the SAME Parameters to every proof; they are created once,
 * crates/epoch-snark/src/api/setup.rs:63-105).  _precompute builds, in the device memory of the calling thread's device, the table
 * T[j][i] = 2^(c j) P_i for the W = ceil((scalar bits + 1) / c) digit positions of a scalar (affine, device form: n W x 128 B for G1 of
 * BLS12-377, n W x 256 B for the other groups; window_bits c = 0 picks it from n, 16 <= c <= 22) and returns an opaque handle.
 * _fixed then computes sum_i k_i P_i for the first n_scalars <= n bases (the shorter side decides, as VariableBaseMSM zips) with ALL
 * signed c-bit digits in ONE set of 2^(c-1) buckets: one bucket reduction instead of W, no Horner chain over windows, and - c not being
 * bound to the 16 bits of the variable-base windows - fewer digits per scalar, i.e. fewer bucket additions (19 instead of 24 per
 * 377-bit scalar at c = 20).  Same result as msm_* (which stays VariableBaseMSM: nothing is cached behind its back).
 * A base may be flagged as the identity (inf); a base whose 2^(c j) multiple is the identity is handled.  The handle is bound to the
 * device it was built on (a call from a thread bound to another device returns 101) and may be used from several host threads at once.
 * Size limits (the pipeline's 32-bit run offsets): n W < 2^31 and n W NV < 2^32 with NV = the 2^15-bucket virtual windows of a c-bit digit
 * (1 at c = 16, then 3, 5, 9, 17, 33, 65 at c = 17 .. 22).  For the 377-bit scalars of BW6-761 that is n < 2^26.4 at c = 16, 2^25.9 at 17,
 * 2^25.3 at 18, 2^24.5 at 19 (a 2^24-term key, the prover's size, fits), 2^23.7 at 20, 2^22.8 at 21, 2^21.8 at 22; for the 253-bit scalars
 * of BLS12-377 about 1.5 times that.  window_bits = 0 starts from the measured optimum (20 / 21) and steps down until the limits hold;
 * an explicit window_bits that does not fit returns 2.
 * celo_amd_msm_fixed_info: the table's shape, its size in bytes and its build time (HIP events). */
int msm_bls12_377_g1_precompute(const uint64_t* bases_xy /* n*12 */, const uint8_t* inf, size_t n, int window_bits, void** out_handle);
int msm_bls12_377_g2_precompute(const uint64_t* bases_xy /* n*24 */, const uint8_t* inf, size_t n, int window_bits, void** out_handle);
int msm_bw6_761_g1_precompute(const uint64_t* bases_xy /* n*24 */, const uint8_t* inf, size_t n, int window_bits, void** out_handle);
int msm_bw6_761_g2_precompute(const uint64_t* bases_xy /* n*24 */, const uint8_t* inf, size_t n, int window_bits, void** out_handle);
int msm_bls12_377_g1_precompute_dev(const void* d_bases_xy, const void* d_inf, size_t n, int window_bits, void** out_handle);
int msm_bls12_377_g2_precompute_dev(const void* d_bases_xy, const void* d_inf, size_t n, int window_bits, void** out_handle);
int msm_bw6_761_g1_precompute_dev(const void* d_bases_xy, const void* d_inf, size_t n, int window_bits, void** out_handle);
int msm_bw6_761_g2_precompute_dev(const void* d_bases_xy, const void* d_inf, size_t n, int window_bits, void** out_handle);
int msm_bls12_377_g1_fixed(const void* handle, const uint64_t* scalars /* n_scalars*4 */, size_t n_scalars, uint64_t out_xyz[18]);
int msm_bls12_377_g2_fixed(const void* handle, const uint64_t* scalars /* n_scalars*4 */, size_t n_scalars, uint64_t out_xyz[36]);
int msm_bw6_761_g1_fixed(const void* handle, const uint64_t* scalars /* n_scalars*6 */, size_t n_scalars, uint64_t out_xyz[36]);
int msm_bw6_761_g2_fixed(const void* handle, const uint64_t* scalars /* n_scalars*6 */, size_t n_scalars, uint64_t out_xyz[36]);
int msm_bls12_377_g1_fixed_dev(const void* handle, const void* d_scalars, size_t n_scalars, uint64_t out_xyz[18], void* hip_stream);
int msm_bls12_377_g2_fixed_dev(const void* handle, const void* d_scalars, size_t n_scalars, uint64_t out_xyz[36], void* hip_stream);
int msm_bw6_761_g1_fixed_dev(const void* handle, const void* d_scalars, size_t n_scalars, uint64_t out_xyz[36], void* hip_stream);
int msm_bw6_761_g2_fixed_dev(const void* handle, const void* d_scalars, size_t n_scalars, uint64_t out_xyz[36], void* hip_stream);
int celo_amd_msm_fixed_release(void* handle);
int celo_amd_msm_fixed_info(const void* handle, size_t* n, int* window_bits, int* windows, size_t* table_bytes, float* build_ms);

/* ---- batched MSMs: m independent instances in one call; instance p owns points/scalars [offsets[p], offsets[p+1])
 * (offsets has m+1 entries), out_xyz holds m Jacobian results back to back.  This is the shape of Batch::verify
 * (crates/bls-crypto/src/bls/batch.rs:69,76 — one G2 and one G1 MSM over the batch's signers) when
 * batch_verify_strict (crates/bls-snark-sys/src/signatures.rs:343-400) is handed many batches; the reference loops over
 * them serially (signatures.rs:358).  Instances of up to 1024 points run on the batched kernels; larger ones are
 * processed one at a time on the large-MSM pipeline. */
int msm_batch_bls12_377_g1(const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, const uint32_t* offsets, size_t m, uint64_t* out_xyz /* m*18 */);
int msm_batch_bls12_377_g2(const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, const uint32_t* offsets, size_t m, uint64_t* out_xyz /* m*36 */);
/* The same call for bases the caller vouches to be elements of the prime-order subgroup G2 - public keys: a PublicKey of the reference is
 * one by construction (checked deserialisation crates/bls-crypto/src/bls/public.rs:123-149, secret keys, sums) - which is what
 * Batch::verify hands over (batch.rs:69).  Same result; the library may then split every scalar with the endomorphism psi (psi(P) = [x]P
 * on G2): an instance of n points with up to 253-bit scalars becomes one of up to 4 n points with 64-bit scalars (csrc/msm.h, k_gls_expand).
 * For a base outside the subgroup the result is unspecified (use msm_batch_bls12_377_g2, which is VariableBaseMSM on any curve point). */
int msm_batch_bls12_377_g2_subgroup(const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, const uint32_t* offsets, size_t m, uint64_t* out_xyz /* m*36 */);
int msm_batch_bw6_761_g1(const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, const uint32_t* offsets, size_t m, uint64_t* out_xyz /* m*36 */);
int msm_batch_bw6_761_g2(const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, const uint32_t* offsets, size_t m, uint64_t* out_xyz /* m*36 */);

/* ---- Batch::verify for m batches in one call, chained on the device (crates/bls-crypto/src/bls/batch.rs:44-84: per batch
 * P = sum_j e_j pk_j (G2 MSM), S = sum_j e_j sig_j (G1 MSM), accept iff e(S, -g2) * e(H(m), P) == 1; the FFI loops over batches
 * serially, crates/bls-snark-sys/src/signatures.rs:358).  Batch b owns keys / signatures / exponents [offsets[b], offsets[b+1])
 * (at most 1024 each); hash_xy[b] = H(m_b) affine; neg_g2_xy = the negated G2 generator, affine (the caller's constant; the
 * library restates no curve constant here).  Both batch MSMs run concurrently on two engines; their Jacobian results are
 * normalised ON the device straight into the pairing engine's input slots - nothing returns to the host but the m verdicts
 * out_ok[b] in {0, 1}.  *_inf: optional byte-per-point identity flags (NULL = none; a batch whose hash is flagged is checked without
 * that pair).  Exponents: canonical 4 x u64 (Batch::verify draws 128 + log2(n) random bits; the window count adapts
 * to the longest one present).  _dev: every pointer except offsets, neg_g2_xy and out_ok is a DEVICE pointer.
 * PRECONDITION: every public key pk_xy[i] is an element of the prime-order subgroup G2 and every signature an element of G1 - what the
 * reference's PublicKey / Signature values are by construction (checked deserialisation crates/bls-crypto/src/bls/public.rs:123-149,
 * signature.rs:31-57; secret keys; sums of such) and what Batch::verify therefore assumes.  The key sums use the endomorphism psi(P) = [x]P,
 * which holds on G2 only: for a key that is on the twist but outside G2 (e.g. decoded with decompress_bls12_377_g2(check_subgroup = 0))
 * the verdict is unspecified.  A caller holding unchecked points must run them through decompress_*(check_subgroup = 1) first - or use
 * msm_batch_bls12_377_g2 + msm_batch_bls12_377_g1 + pairing_product_is_one_batch_bls12_377, which are VariableBaseMSM on any curve point. */
int batch_verify_bls12_377(const uint64_t* pk_xy /* tot x 24 */, const uint8_t* pk_inf /* tot or NULL */, const uint64_t* sig_xy /* tot x 12 */,
                           const uint8_t* sig_inf /* tot or NULL */, const uint64_t* exponents /* tot x 4 */, const uint32_t* offsets /* m+1 */,
                           const uint64_t* hash_xy /* m x 12 */, const uint8_t* hash_inf /* m or NULL */, const uint64_t neg_g2_xy[24], size_t m,
                           uint8_t* out_ok /* m */);
int batch_verify_bls12_377_dev(const void* d_pk_xy, const void* d_pk_inf, const void* d_sig_xy, const void* d_sig_inf, const void* d_exponents,
                               const uint32_t* offsets, const void* d_hash_xy, const void* d_hash_inf, const uint64_t neg_g2_xy[24], size_t m,
                               uint8_t* out_ok /* m */);

/* The random exponents Seam A's batch_verify_strict draws for one call, as its device kernel produces them (tests / tooling): signer i
 * of the call (i < offsets[m]) gets the first (128 + ceil(log2 n_b) + 7) / 8 bytes - n_b the size of its batch, as
 * byte_count_from_target_batch_size, crates/bls-crypto/src/bls/batch.rs:23-28 - of block i of the ChaCha20 stream (RFC 7539 block
 * function, 64-bit block counter, zero nonce) under `key`, little-endian in 4 x u64.  out: offsets[m] x 4 u64, host.  batch.rs:51-58 draws
 * the same number of bytes per signer from rand::thread_rng(). */
int celo_amd_draw_batch_exponents(const uint32_t key[8], const uint32_t* offsets /* m+1 */, size_t m, uint64_t* out);

/* ---- pairing product check.  Replaces `Bls12_377::product_of_pairings(&pairs) == Fq12::one()` at
 *   crates/bls-crypto/src/bls/public.rs:102    (PublicKey::verify_sig: 2 pairs)
 *   crates/bls-crypto/src/bls/signature.rs:149 (Signature::batch_verify_hashes: n+1 pairs, one final exponentiation)
 * g1_xy: k affine G1 points (k*12 u64), g2_xy: k affine G2 points (k*24 u64), optional infinity byte arrays (a pair with
 * an infinite point contributes 1, as in ark-ec).  *is_one = 1 iff the product of the k pairings is the identity of GT.
 * The batch form checks m independent products in one launch: product p covers pairs [offsets[p], offsets[p+1]),
 * offsets has m+1 entries — the per-batch checks of Batch::verify (crates/bls-crypto/src/bls/batch.rs:83) for many
 * batches at once (crates/bls-snark-sys/src/signatures.rs:358 loops over them serially). */
int pairing_product_is_one_bls12_377(const uint64_t* g1_xy, const uint8_t* inf1, const uint64_t* g2_xy, const uint8_t* inf2, size_t k,
                                     int* is_one);
int pairing_product_is_one_batch_bls12_377(const uint64_t* g1_xy, const uint8_t* inf1, const uint64_t* g2_xy, const uint8_t* inf2,
                                           const uint32_t* offsets, size_t m, uint8_t* is_one);
/* Test/inspection hook: the GT value itself (72 u64 per product: the arkworks in-memory Fq12, Montgomery form);
 * miller_only != 0 skips the final exponentiation (product of Miller-loop values). */
int celo_amd_pairing_gt_bls12_377(const uint64_t* g1_xy, const uint8_t* inf1, const uint64_t* g2_xy, const uint8_t* inf2,
                                  const uint32_t* offsets, size_t m, int miller_only, uint64_t* gt72);
/* BW6-761: the pairing product check underneath ark_groth16::verify_proof (crates/epoch-snark/src/api/verifier.rs:35, reached
 * from the FFI `verify` at crates/bls-snark-sys/src/snark/mod.rs:23-45):
 *   e(A,B) * e(acc,-gamma) * e(C,-delta) == e(alpha,beta)   <=>   product over {(A,B),(acc,-gamma),(C,-delta),(-alpha,beta)} == 1.
 * g1_xy: k*24 u64, g2_xy: k*24 u64 (G2 coordinates are in Fq: M-type sextic twist). */
int pairing_product_is_one_bw6_761(const uint64_t* g1_xy, const uint8_t* inf1, const uint64_t* g2_xy, const uint8_t* inf2, size_t k,
                                   int* is_one);
int celo_amd_pairing_gt_bw6_761(const uint64_t* g1_xy, const uint8_t* inf1, const uint64_t* g2_xy, const uint8_t* inf2,
                                const uint32_t* offsets, size_t m, int miller_only, uint64_t* gt72);
/* ms[4] = {miller loops, GT products, final exponentiations, total} of the last pairing call (HIP events). */
int celo_amd_pairing_last_timings(float ms[4]);

/* ---- radix-2 NTT over Fr(BW6-761) (= Fq of BLS12-377, 377 bits), in place, natural order in and out: the witness-map
 * FFTs of the Groth16 prover.  Replaces ark-poly 0.1 Radix2EvaluationDomain::{fft, ifft,
 * coset_fft, coset_ifft}_in_place as called by ark_groth16::create_proof_no_zk (crates/epoch-snark/src/api/prover.rs:78,112).
 * data: n = 2^log_n elements, arkworks Montgomery limbs (6 u64 each).  omega: the domain's group_gen (its inverse for an
 * inverse transform).  coset: NULL, or a generator g: every x_i is multiplied by g^i BEFORE the transform (coset_after =
 * 0: coset_fft with g = the coset offset) or AFTER it (coset_after = 1: coset_ifft with g = offset^-1).  scale: NULL, or a
 * factor applied to every output (size_inv for the inverse transforms).  log_n <= 28. */
int ntt_bw6_761_fr(uint64_t* data, unsigned log_n, const uint64_t omega[6], const uint64_t* coset, int coset_after, const uint64_t* scale);
int ntt_bw6_761_fr_dev(uint64_t* d_data, unsigned log_n, const uint64_t omega[6], const uint64_t* coset, int coset_after,
                       const uint64_t* scale, void* hip_stream);
/* ms[4] = {load/convert, butterfly passes, bit-reversal store, total}, passes = number of butterfly launches (last NTT call). */
int celo_amd_ntt_last_timings(float ms[4], int* passes);

/* ---- Groth16 prover over BW6-761 after R1CS synthesis: what ark_groth16::create_proof_no_zk does
 * with the constraint evaluations and the proving key, as called at crates/epoch-snark/src/api/prover.rs:78,112.
 *
 * groth16_witness_map_bw6_761: R1CStoQAP::witness_map (ark-groth16 0.1 r1cs_to_qap.rs) from the point where a, b, c hold the
 * evaluations of the QAP polynomials over the domain (n = 2^log_n elements each, arkworks Montgomery limbs; the caller builds
 * them from its constraint system, including the input-consistency rows): ifft(a, b, c); coset_fft(a, b, c);
 * ab = (a o b - c) * vanishing_inv; coset_ifft(ab).  On return a holds h (n coefficients; b and c are scratch) - as field
 * elements, or with out_canonical != 0 as canonical integers (Fr::into_repr(), what the h MSM takes).  Domain constants from the
 * caller (nothing of ark-poly is restated): omega = group_gen, omega_inv, coset = the coset offset (F::multiplicative_generator()),
 * coset_inv, size_inv = n^-1, vanishing_inv = (coset^n - 1)^-1 - all arkworks Montgomery limbs.
 *
 * groth16_prove_bw6_761: the proof with r = s = 0 (create_proof_no_zk):
 *   A = a_query[0] + MSM(a_query[1..], assignment) + alpha_g1          B = b_g2_query[0] + MSM(b_g2_query[1..], assignment) + beta_g2
 *   C = MSM(l_query, aux) + MSM(h_query, h)
 * queries: affine points (24 u64 each); assignment: n_assignment canonical scalars (public inputs without the leading 1, then the
 * witness), aux = its last n_aux entries; h: n_h canonical scalars.  As in VariableBaseMSM::multi_scalar_mul the shorter of bases
 * and scalars decides each MSM's length.  The four MSMs run concurrently on four engines.  Results: Jacobian, arkworks layout.
 * The point at infinity: a proving key holds it for every variable absent from A, B or the auxiliary part (ark-groth16 generator.rs), as
 * GroupAffine::zero() = (x, y, infinity) = (0, 1, true).  These entry points take coordinates only, so a query row - or query[0], alpha, beta
 * - with x = 0 and y = 1 IS the identity here (it contributes nothing whatever its scalar); on none of the groups involved is (0, 1) an
 * element of the prime-order group, so no key element is shadowed (csrc/msm.h k_flag_ark_zero).  The plain msm_* entry points do NOT
 * apply this rule: they take the identity through their `inf` byte arrays. */
int groth16_witness_map_bw6_761(uint64_t* a, uint64_t* b, uint64_t* c, unsigned log_n, const uint64_t omega[6], const uint64_t omega_inv[6], const uint64_t coset[6],
                                const uint64_t coset_inv[6], const uint64_t size_inv[6], const uint64_t vanishing_inv[6], int out_canonical);
int groth16_witness_map_bw6_761_dev(uint64_t* d_a, uint64_t* d_b, uint64_t* d_c, unsigned log_n, const uint64_t omega[6], const uint64_t omega_inv[6],
                                    const uint64_t coset[6], const uint64_t coset_inv[6], const uint64_t size_inv[6], const uint64_t vanishing_inv[6],
                                    int out_canonical, void* hip_stream);
int groth16_prove_bw6_761(const uint64_t* a_query, size_t na, const uint64_t* b_g2_query, size_t nb, const uint64_t* h_query, size_t nh, const uint64_t* l_query,
                          size_t nl, const uint64_t alpha_g1[24], const uint64_t beta_g2[24], const uint64_t* assignment, size_t n_assignment, size_t n_aux,
                          const uint64_t* h, size_t n_h, uint64_t out_a[36], uint64_t out_b[36], uint64_t out_c[36]);

/* ---- the proof against a LOADED key: groth16_load_key_* builds the fixed-base tables of the four queries once (msm_*_precompute above; the
 * reference creates its Parameters once, crates/epoch-snark/src/api/setup.rs:63-105, and hands the same ones to every
 * create_proof_no_zk, prover.rs:78,112); groth16_prove_with_key then computes A, B, C of groth16_prove_bw6_761 / _bls12_377 with four
 * msm_*_fixed calls (concurrent, one engine each) and no query crossing PCIe again.  Same arguments, identity-row rule, results and
 * error behaviour as the entry points above; a_query[0] / b_g2_query[0] / alpha / beta are kept with the key.  window_bits: the tables'
 * window size (0 = automatic).  A key is bound to the device it was loaded on (101 from another) and may prove from several threads. */
int groth16_load_key_bw6_761(const uint64_t* a_query, size_t na, const uint64_t* b_g2_query, size_t nb, const uint64_t* h_query, size_t nh, const uint64_t* l_query, size_t nl,
                             const uint64_t alpha_g1[24], const uint64_t beta_g2[24], int window_bits, void** out_key);
int groth16_load_key_bls12_377(const uint64_t* a_query, size_t na, const uint64_t* b_g2_query, size_t nb, const uint64_t* h_query, size_t nh, const uint64_t* l_query, size_t nl,
                               const uint64_t alpha_g1[12], const uint64_t beta_g2[24], int window_bits, void** out_key);
int groth16_prove_with_key(const void* key, const uint64_t* assignment, size_t n_assignment, size_t n_aux, const uint64_t* h, size_t n_h, uint64_t* out_a, uint64_t* out_b,
                           uint64_t* out_c);
int groth16_free_key(void* key);

/* ---- the same three steps over BLS12-377: the hash-helper proof of an epoch (crates/epoch-snark/src/api/prover.rs:83-118,
 * create_proof_no_zk::<BLSCurve, _> at :112), whose witness map runs over Fr(BLS12-377) (253 bits, 2-adicity 47; elements are 4 u64 in
 * arkworks Montgomery form, R = 2^256) and whose queries are BLS12-377 points: a_query / h_query / l_query / alpha_g1 in G1 (affine
 * 12 u64), b_g2_query / beta_g2 in G2 (24 u64); assignment and h are canonical 4-u64 scalars; A and C come back as G1 Jacobian
 * (18 u64), B as G2 Jacobian (36 u64).  Same argument meaning, transforms and error behaviour as the BW6-761 entry points above. */
int ntt_bls12_377_fr(uint64_t* data, unsigned log_n, const uint64_t omega[4], const uint64_t* coset, int coset_after, const uint64_t* scale);
int ntt_bls12_377_fr_dev(uint64_t* d_data, unsigned log_n, const uint64_t omega[4], const uint64_t* coset, int coset_after,
                         const uint64_t* scale, void* hip_stream);
int groth16_witness_map_bls12_377(uint64_t* a, uint64_t* b, uint64_t* c, unsigned log_n, const uint64_t omega[4], const uint64_t omega_inv[4], const uint64_t coset[4],
                                  const uint64_t coset_inv[4], const uint64_t size_inv[4], const uint64_t vanishing_inv[4], int out_canonical);
int groth16_witness_map_bls12_377_dev(uint64_t* d_a, uint64_t* d_b, uint64_t* d_c, unsigned log_n, const uint64_t omega[4], const uint64_t omega_inv[4],
                                      const uint64_t coset[4], const uint64_t coset_inv[4], const uint64_t size_inv[4], const uint64_t vanishing_inv[4],
                                      int out_canonical, void* hip_stream);
int groth16_prove_bls12_377(const uint64_t* a_query, size_t na, const uint64_t* b_g2_query, size_t nb, const uint64_t* h_query, size_t nh, const uint64_t* l_query,
                            size_t nl, const uint64_t alpha_g1[12], const uint64_t beta_g2[24], const uint64_t* assignment, size_t n_assignment, size_t n_aux,
                            const uint64_t* h, size_t n_h, uint64_t out_a[18], uint64_t out_b[36], uint64_t out_c[18]);

/* ---- bulk decoding of compressed points: n keys or signatures in arkworks 0.1 wire form
 * (G1: 48 B, G2: 96 B; x little-endian, flag bits 0x80 = "y is the larger root", 0x40 = infinity in the last byte) to
 * affine (x, y) in arkworks Montgomery limbs - the layout the MSM and pairing entry points take.  One point per GPU lane:
 * the square root (table-driven in Fq, the norm method in Fq2), the sign choice and, with check_subgroup != 0, r*P == O.
 * Replaces the per-key work of PublicKey::deserialize / Signature::deserialize (crates/bls-crypto/src/bls/public.rs:123-149,
 * signature.rs:31-57: GroupAffine::deserialize = get_point_from_x + is_in_correct_subgroup_assuming_on_curve) and of the
 * per-validator loop of the epoch FFI (crates/bls-snark-sys/src/snark/epoch_block.rs:187-196).
 * status[i]: 0 = decoded, 1 = the encoding of the point at infinity, 2 = not a valid encoding (x >= q or no y exists),
 * 3 = on the curve but outside the prime-order subgroup; out_xy[i] is all zero unless status[i] == 0. */
int decompress_bls12_377_g1(const uint8_t* in /* n x 48 */, size_t n, int check_subgroup, uint64_t* out_xy /* n x 12 */, uint8_t* status /* n */);
int decompress_bls12_377_g2(const uint8_t* in /* n x 96 */, size_t n, int check_subgroup, uint64_t* out_xy /* n x 24 */, uint8_t* status /* n */);
int decompress_bls12_377_g1_dev(const uint8_t* d_in, size_t n, int check_subgroup, uint64_t* d_out_xy, uint8_t* d_status, void* hip_stream);
int decompress_bls12_377_g2_dev(const uint8_t* d_in, size_t n, int check_subgroup, uint64_t* d_out_xy, uint8_t* d_status, void* hip_stream);
/* Jacobian -> affine for n points in one launch (Montgomery's trick inside each lane): replaces
 * ProjectiveCurve::batch_normalization_into_affine as called before every MSM (crates/bls-crypto/src/bls/signature.rs:82,
 * public.rs:58).  jac: n x (X, Y, Z) arkworks Montgomery limbs (G1: 18 u64, G2: 36 u64 per point; identity = Z == 0);
 * out_xy: n x (x, y) in the layout the MSM / pairing entry points take, a zero row and inf[i] = 1 for the identity. */
int normalize_bls12_377_g1(const uint64_t* jac /* n x 18 */, size_t n, uint64_t* out_xy /* n x 12 */, uint8_t* inf /* n */);
int normalize_bls12_377_g2(const uint64_t* jac /* n x 36 */, size_t n, uint64_t* out_xy /* n x 24 */, uint8_t* inf /* n */);
/* kernel time (HIP events) of the last decompress / normalize call */
int celo_amd_decompress_last_ms(float* ms);

/* ---- batched hash-to-G1, DIRECT hasher: n messages per launch, one per GPU lane.
 * Replaces n calls of TryAndIncrement<DirectHasher, G1>::hash_with_attempt(domain, message, extra_data)
 * (crates/bls-crypto/src/hash_to_curve/try_and_increment.rs:87-139; DirectHasher = Blake2s CRH + Blake2Xs XOF,
 * crates/bls-crypto/src/hashers/direct.rs:23-80) as Signature::batch_verify issues them (bls/signature.rs:111-114), with the
 * deployed `compat` bit logic.  domain: the 8-byte personalisation (SIG_DOMAIN "ULforxof" / POP_DOMAIN "ULforpop",
 * crates/bls-crypto/src/lib.rs:75,78).  msgs / extras: concatenated bytes, message i = msgs[msg_off[i] .. msg_off[i+1]),
 * likewise extras (extra_off == NULL: no extra data anywhere).  out_xy: n x 12 u64, affine (x, y) in arkworks Montgomery
 * limbs (the G1 layout of the MSM / pairing entry points); attempts[i]: the counter that produced the point, 255 = no
 * counter below 255 did (the reference returns an error there; out row zero). */
int hash_to_g1_direct_bls12_377(const uint8_t domain[8], const uint8_t* msgs, const uint64_t* msg_off /* n+1 */, const uint8_t* extras,
                                const uint64_t* extra_off /* n+1 or NULL */, size_t n, uint64_t* out_xy /* n x 12 */, uint8_t* attempts /* n */);
/* The try-and-increment loop of the CIP22 hashers (crates/bls-crypto/src/hash_to_curve/try_and_increment_cip22.rs:81-134), n
 * messages per launch: the caller computes the inner CRH of each message once (for the composite hasher: hash_crh of
 * include/celo_bls_snark_sys.h, 48 bytes), this entry point runs candidate = xof(domain, counter || extra_data || inner) ->
 * curve point -> cofactor for every message.  Same layout and result conventions as hash_to_g1_direct_bls12_377. */
int hash_to_g1_cip22_tail_bls12_377(const uint8_t domain[8], const uint8_t* inner, const uint64_t* inner_off /* n+1 */, const uint8_t* extras,
                                    const uint64_t* extra_off /* n+1 or NULL */, size_t n, uint64_t* out_xy /* n x 12 */, uint8_t* attempts /* n */);
/* Batched hash-to-G1 with the COMPOSITE hasher (Pedersen CRH + Blake2Xs XOF), n messages per call: cip22 == 0:
 * TryAndIncrement<CompositeHasher, G1>::hash_with_attempt (try_and_increment.rs:87-139: a CRH of counter || extra || message per
 * attempt); cip22 != 0: the CIP22 form (try_and_increment_cip22.rs:81-134: one CRH per message, then
 * hash_to_g1_cip22_tail_bls12_377).  What hash_composite / hash_composite_cip22 of the bls-snark-sys ABI compute for one
 * message (signatures.rs:143,215).  Layout and results as hash_to_g1_direct_bls12_377. */
int hash_to_g1_composite_bls12_377(const uint8_t domain[8], const uint8_t* msgs, const uint64_t* msg_off /* n+1 */, const uint8_t* extras,
                                   const uint64_t* extra_off /* n+1 or NULL */, size_t n, int cip22, uint64_t* out_xy /* n x 12 */, uint8_t* attempts /* n */);
/* The composite hasher's CRH for n messages in one launch: the Bowe-Hopwood-Pedersen hash over ed-on-BW6-761 that
 * CompositeHasher::crh evaluates (crates/bls-crypto/src/hashers/composite.rs:79-86; hash_crh of the bls-snark-sys ABI,
 * signatures.rs:169, is the one-message form).  out48: n x 48 bytes, the affine x coordinate little-endian.  A message longer
 * than 93 * 560 * 3 bits is an error for the whole call (the reference panics). */
int composite_crh_bls12_377(const uint8_t* msgs, const uint64_t* msg_off /* n+1 */, size_t n, uint8_t* out48 /* n x 48 */);
/* kernel time (HIP events) of the last hash_to_g1_* call */
int celo_amd_hash_last_ms(float* ms);

/* ---- plain sums of k Jacobian points (host pointers, arkworks layout; host-side, for small k): the fold of per-GPU
 * partial MSM results and small aggregates — Signature::aggregate / PublicKey::aggregate
 * (crates/bls-crypto/src/bls/signature.rs:61-67, public.rs:38-44). */
int celo_amd_sum_jacobian_bls12_377_g1(const uint64_t* jac /* k*18 */, size_t k, uint64_t out_xyz[18]);
int celo_amd_sum_jacobian_bls12_377_g2(const uint64_t* jac /* k*36 */, size_t k, uint64_t out_xyz[36]);
int celo_amd_sum_jacobian_bw6_761(const uint64_t* jac /* k*36 */, size_t k, uint64_t out_xyz[36]);

/* ---- instrumentation (bench.py / tests).  group: 0 = bls12_377_g1, 1 = bls12_377_g2, 2 = bw6_761.
 * ms[5] = {convert, sort, accumulate, reduce, total} of the last MSM on that engine, from HIP events recorded on the
 * MSM's own stream; cfg[3] = {window bits c, windows, buckets}. */
int celo_amd_msm_last_timings(int group, float ms[5], int cfg[3]);
/* Self-test of the bucket-accumulation kernels the library runs (tests/test_msm_gpu.py): `runs` runs of `len` random signed multiples of the
 * affine generator gen_xy (HOST pointer, arkworks layout) through k_accumulate<group> (chunked = 0) or through the host-pointer pipeline's
 * k_accumulate_chunk<group> twice, the second launch continuing the first's carried sums (chunked = 1); the first `check` partial sums are
 * compared limb for limb with the same formulas run on the host.  *differ = the number of runs that disagree (0 expected). */
int celo_amd_selftest_accumulate(int group, const uint64_t* gen_xy, uint32_t runs, uint32_t len, uint32_t seed, uint32_t check, int chunked, uint32_t* differ);
/* The multiplier roofline of THIS device in THIS process (bench.py's valu_roofline.peak): chip-wide rate of the library's own field-product
 * bodies in register-resident loops (csrc/unit_ubench.hip; ~0.3 s).  out[0..3] = 1e9 products/s: Fq(BLS12-377) mul, sqr, Fq(BW6-761) mul,
 * sqr; out[4] = shader clock in MHz during the first loop (s_memtime against the 100 MHz s_memrealtime); out[5..8] = the loops' kernel ms. */
int celo_amd_ubench_fp(float out[9]);
/* Forces the Pippenger window size (0 = automatic) — tuning and test hook. */
int celo_amd_msm_set_window_bits(int group, int c);
/* Host-pointer entry points (msm_<group>, from 2^18 terms - 2^17 when the count is set here): the number of index chunks in which scalars and bases cross PCIe while the
 * chunks already on the device are sorted and accumulated (csrc/msm.h HostIn).  0 = the unpipelined form (three transfers, then
 * the resident pipeline; 1 = one chunk: the sort runs beside the bases' transfer), -1 = the default (CELO_HOST_CHUNKS, else n / 2^18 within [4, 16] - [8, 16] for BW6-761, n / 2^19 within [4, 8] for G2 of BLS12-377).  The first chunk is cut
 * in halves once where the halves keep 2^17 points (else not at all), the last one never; `chunks | (h + 1) << 8 | (t + 1) << 12`
 * sets those counts to h and t (0 <= h, t <= 8) as well.  Process-wide — tuning and test hook. */
int celo_amd_msm_set_host_chunks(int chunks);
/* The BW6-761 bucket accumulation of the resident entry points (msm_bw6_761_*_dev, and msm_bw6_761_* below 2^18 terms): 1 = three batched-affine
 * tree levels before the XYZZ chain (csrc/msm_ba.h), 0 = the XYZZ chain alone, -1 = the default (CELO_BA, else 0: the two measure alike on
 * config 4, profiles/r6_ba_ab.txt).  Same group element either way.  Process-wide - test and measurement hook; 1 for an argument out of range. */
int celo_amd_msm_set_batched_affine(int on);
/* The chunk plan the pipelined host-pointer entry uses for n terms (pure host arithmetic, no device call - csrc/runtime.h host_chunk_plan):
 * returns the number of chunks K (<= 80) and their lengths lens[0..K) in the order they are sent, *cm = the chunk capacity (chunk k sits at
 * the virtual index k * cm on the device); -1 for arguments the entry points would not pipeline (chunks < 1 or > 64, n < chunks * 2^16, n >= 2^30; chunks = 1 is the one-chunk pipelined form celo_amd_msm_set_host_chunks(1) runs). */
int celo_amd_msm_host_chunk_plan(uint64_t n, int chunks, int head_split, int tail_split, uint32_t* cm, uint32_t lens[80]);

/* ---- synthetic-workload generators (bench / tests only): writes n affine points
 * P_i = k_i * G into DEVICE memory in the arkworks layout, k_i = 64-bit splitmix64(seed, i) | 1.
 * gen_xy: the generator G, affine, HOST pointer, arkworks layout. */
int celo_amd_gen_points_bls12_377_g1_dev(void* d_out_xy, size_t n, uint64_t seed, const uint64_t* gen_xy, void* stream);
int celo_amd_gen_points_bls12_377_g2_dev(void* d_out_xy, size_t n, uint64_t seed, const uint64_t* gen_xy, void* stream);
int celo_amd_gen_points_bw6_761_dev(void* d_out_xy, size_t n, uint64_t seed, const uint64_t* gen_xy, void* stream);
/* The same with one generator per group of `per` consecutive points: P_i = k_i * gens[i / per] (ngens >= ceil(n / per) affine
 * generators, HOST pointer).  Builds VALID Batch::verify inputs without a host big-int loop: the
 * signatures of batch b = k_{b,j} * H(m_b) with gens = the message hashes, the keys = k_{b,j} * g2 from the same seed. */
int celo_amd_gen_points_grouped_bls12_377_g1_dev(void* d_out_xy, size_t n, uint64_t seed, const uint64_t* gens_xy, size_t ngens, uint32_t per, void* stream);
int celo_amd_gen_points_grouped_bls12_377_g2_dev(void* d_out_xy, size_t n, uint64_t seed, const uint64_t* gens_xy, size_t ngens, uint32_t per, void* stream);

#ifdef __cplusplus
}
#endif
#endif
