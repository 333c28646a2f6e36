// Pippenger bucket MSM for gfx950, generic over the group (BLS12-377 G1/G2, BW6-761 G1/G2).
//
// Replaces ark-ec VariableBaseMSM::multi_scalar_mul as called at
//   crates/bls-crypto/src/bls/signature.rs:85 (G1), public.rs:61 (G2) and inside
//   ark_groth16::create_proof_no_zk (crates/epoch-snark/src/api/prover.rs:78,112).
// The reference's algorithm (unsigned c-bit windows, one rayon task per
// window, serial bucket fill + running sum) is NOT what runs here; only its result is matched.
//
// GPU pipeline (all on one HIP stream, no host sync until the window sums come back):
//   1 k_convert_bases   ark Montgomery (2^384 / 2^768 radix, 64-bit limbs) -> 28-bit-limb device form,
//                       128 B (G1-377) / 256 B per affine point, coalesced in, 16-B vector stores out
//   2 k_digits          signed c-bit digits per scalar, stored once as u16 per (window, scalar)
//     k_part_hist       two-level counting sort, level 1: a window's entries are partitioned into <= 128 bins by the low bits
//     k_part_scan       of the bucket index (per-block LDS histograms, one scan per window, LDS-staged scatter: every global
//     k_part_scatter    store is a run of consecutive addresses)
//     k_tile_count      level 2: a bin's region (tens of KB, L2-resident) is sorted by the remaining <= 8 bits in tiles, one
//     k_tile_sort       workgroup each, LDS counters and staging; the region's first tile also cuts its buckets' runs into
//   3                   pieces of <= SEG points (piece ids need no scan over the window)
//     k_size_*          counting sort of the pieces by length -> longest-first schedule, lanes of a wave get equal-length pieces
//   4 k_accumulate      one lane per piece: gathers its points (128 B each), XYZZ mixed adds      <- dominant kernel
//   5 k_combine_big     (skewed inputs only) folds buckets that were cut into many pieces
//     k_bitsum          bit-sliced bucket reduction: binary tree over the bucket index + the odd-node sums of every level
//                       (depth log2(buckets) point additions, no scalar multiplication; see the kernel)
//   6 host              <=4*NW XYZZ points: Horner over windows (c doublings each) -> Jacobian, ark form
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <type_traits>
#include <atomic>
#include <thread>
#include <system_error>
#include "curve.h"
#include "host64.h"
#include "curve_lanes.h"
#include "pairing_lanes.h"
#include "fp2.h"
#include "runtime.h"
#include "gls.h"

namespace celo {

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "[celo-amd] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

// ---------------------------------------------------------------- group configurations
struct G1_377 {
  typedef Fp<P377> F;
  static constexpr int SCALAR_WORDS = 8;   // BigInteger256
  static constexpr int SCALAR_BITS = 253;  // Fr::MODULUS_BITS
  static constexpr const char* NAME = "bls12_377_g1";
};
struct G2_377 {
  typedef Fp2<P377> F;
  static constexpr int SCALAR_WORDS = 8;
  static constexpr int SCALAR_BITS = 253;
  static constexpr const char* NAME = "bls12_377_g2";
};
struct G_761 {  // G1 and G2 of BW6-761 share the coordinate field and the a = 0 group law
  typedef Fp<P761> F;
  static constexpr int SCALAR_WORDS = 12;  // BigInteger384
  static constexpr int SCALAR_BITS = 377;
  static constexpr const char* NAME = "bw6_761";
};

template <class F> struct PointIO {
  static constexpr int FW = F::WORDS;       // device words per coordinate (padded to 16 B)
  static constexpr int AFF_WORDS = 2 * FW;  // affine point
  static constexpr int XYZZ_WORDS = 4 * FW;
  static constexpr int ARK64 = F::ARK64;    // u64 per coordinate in arkworks layout
  HD static Affine<F> load_affine(const uint32_t* p) { return {F::load(p), F::load(p + FW)}; }
  HD static void store_affine(uint32_t* p, const Affine<F>& a) { a.x.store(p); a.y.store(p + FW); }
  HD static Xyzz<F> load_xyzz(const uint32_t* p) {
    return {F::load(p), F::load(p + FW), F::load(p + 2 * FW), F::load(p + 3 * FW)};
  }
  HD static void store_xyzz(uint32_t* p, const Xyzz<F>& a) {
    a.X.store(p); a.Y.store(p + FW); a.ZZ.store(p + 2 * FW); a.ZZZ.store(p + 3 * FW);
  }
};

// ---------------------------------------------------------------- kernels
template <class G>
__global__ void __launch_bounds__(256) k_convert_bases(const uint64_t* __restrict__ ark, uint32_t* __restrict__ dev, size_t n) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* s = ark + i * 2 * IO::ARK64;
  Affine<F> a = {F::from_ark(s), F::from_ark(s + IO::ARK64)};
  IO::store_affine(dev + i * IO::AFF_WORDS, a);
}

// arkworks' GroupAffine::zero() is (x, y, infinity) = (0, 1, true); a caller that hands over coordinates only (a Groth16 proving key's
// queries hold the identity for every variable absent from A / B / the auxiliary part: ark-groth16 generator.rs) encodes it as x = 0,
// y = 1.  On the three curves of this library (0, 1) is never an element of the prime-order group (BLS12-377 G1: a point of order 3;
// its G2 and both BW6-761 groups: not on the curve), so the prover's entry points flag such rows as the identity.
// one_ark: y's arkworks limbs for 1 (Fq2: (1, 0)).
template <int ARK64X> struct ArkCoord { uint64_t v[ARK64X]; };
template <class G>
__global__ void __launch_bounds__(256) k_flag_ark_zero(const uint64_t* __restrict__ ark, const uint8_t* inf_in, uint8_t* inf_out, size_t n,     // (inf_in may BE inf_out: no __restrict__)
                                                       ArkCoord<PointIO<typename G::F>::ARK64> one_ark) {
  constexpr int A = PointIO<typename G::F>::ARK64;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* s = ark + i * 2 * A;
  uint64_t o = 0;
#pragma unroll
  for (int k = 0; k < A; k++) o |= s[k] | (s[A + k] ^ one_ark.v[k]);
  inf_out[i] = (o == 0 || (inf_in && inf_in[i])) ? 1 : 0;
}

}  // namespace celo

#include "msm_sort.h"        // 2-3  digits, two-level sort, pieces, longest-first schedule
#include "msm_accumulate.h"  // 4    k_accumulate / _pair / _chunk
#include "msm_reduce.h"      // 5    k_combine_*, k_bitsum*
#include "msm_batch.h"       //      batched small MSMs, GLV / GLS expansion
#include "msm_fixed.h"       //      fixed-base tables
#include "msm_engine.h"      // 6    host driver (and msm_ba.h: the batched-affine pre-levels, off by default)
