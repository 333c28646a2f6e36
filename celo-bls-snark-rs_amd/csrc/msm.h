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

// ---- 2a. signed-digit recoding, once per MSM: digits[w*n + i] (u16): 0xFFFF = zero digit, else (|d|-1) | (d<0)<<15.
// Windows of MIXED width: NW windows of CB bits cover more than the scalar needs, and a plain split leaves a ragged top window of
// few, heavy buckets (253 = 15 * 16 + 13: 4096 buckets of 256 points at 2^20; 377 = 23 * 16 + 9: 256 buckets of 4096 - pieces to
// fold, 0.68 ms of a BW6-761 MSM).  With KN = NW * CB - (SCALAR_BITS + 1) > 0 the top KN windows are CB - 1 bits wide instead:
// NW - KN windows of CB bits + KN of CB - 1 = SCALAR_BITS + 1 bits exactly (14 * 16 + 2 * 15 = 254, 18 * 16 + 6 * 15 = 378), every
// window full.  A narrow window uses the lower half of its bucket table; its top tree level is empty, so the host's Horner pass
// leaves that level and its doubling out.  The extra bit is the headroom of the signed recoding: bits from SCALAR_BITS up are
// ignored (as ark-ec's VariableBaseMSM ignores them), so the top digit plus its carry never exceeds 2^(width - 1).
// CHUNKED layout (round 5, the host-pointer pipeline: run_device_windows' HostIn): the n scalars are cut into index chunks of m and the
// digits of chunk k, window w form the row (k NW + w) of length m - every (chunk, window) pair is a "virtual window" of the sort below.
// The lanes from n up to npad (the last chunk's tail) write "no digit".  m = npad = n is the plain layout.
template <int SW, int CB, int NW, int KN, int BITS>
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf,
                                                uint16_t* __restrict__ digits, uint32_t n, uint32_t m, uint32_t npad, uint32_t ibase = 0) {
  uint32_t i = ibase + blockIdx.x * blockDim.x + threadIdx.x;     // (ibase .. npad: the range of one launch - the host-pointer pipeline takes chunk 0 first)
  if (i >= npad) return;
  if (m != n) {
    const uint32_t ch = i / m;
    digits += (size_t)ch * NW * m + (i - ch * m);
    if (i >= n) {
#pragma unroll
      for (int w = 0; w < NW; w++) digits[(size_t)w * m] = (uint16_t)0xFFFF;
      return;
    }
  } else digits += i;
  uint32_t s[SW + 1];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * SW);
#pragma unroll
  for (int k = 0; k < SW / 4; k++) {
    uint4 v = sp[k];
    s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w;
  }
  s[SW] = 0;
  if constexpr (BITS < 32 * SW) {          // bits from the scalar length up are not part of the scalar (ark-ec's windows never read them)
    s[BITS / 32] &= (1u << (BITS % 32)) - 1u;
#pragma unroll
    for (int k = BITS / 32 + 1; k < SW; k++) s[k] = 0;
  }
  const bool skip = inf && inf[i];
  constexpr int WIDE = NW - KN;
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    const int width = w < WIDE ? CB : CB - 1;
    const int bit = w < WIDE ? w * CB : WIDE * CB + (w - WIDE) * (CB - 1);
    const uint32_t B = 1u << (width - 1);
    const int wi = bit >> 5, off = bit & 31;
    uint32_t raw = 0;
    if (wi < SW) {
      uint64_t two = ((uint64_t)s[wi + 1] << 32) | s[wi];
      raw = (uint32_t)(two >> off) & ((1u << width) - 1);
    }
    if (KN > 0 && w == NW - 1) raw &= B - 1;     // the window's top bit is bit SCALAR_BITS: not part of the scalar
    uint32_t d = raw + carry;
    uint32_t neg = d > B ? 1u : 0u;
    uint32_t mag = neg ? ((1u << width) - d) : d;
    carry = neg;
    digits[(size_t)w * m] = (mag == 0 || skip) ? (uint16_t)0xFFFF : (uint16_t)((mag - 1) | (neg << 15));
  }
}

// ---- 2b. TWO-LEVEL counting sort of the (window, bucket) keys.  The first design sorted in one level - LDS histograms of all
// 2^15 buckets per block, then a scatter of 4-byte entries into 2^15 runs per window: a block's 65536 entries land two per run,
// every store dirties a line of its own (523 MB written for 67 MB of payload, 0.21 of that sort's 0.41 ms at 2^20; the new one
// takes 0.18 ms, and 0.25 instead of 0.49 ms for the 24 windows of BW6-761).  Here a window's entries are first PARTITIONED into
// NBIN <= 128 bins by the LOW bits of the bucket index (uniform even in a short top window, whose high bits are all zero): a
// block's entries of one bin form a contiguous run of hundreds of bytes that the L2 merges into whole lines.  A bin's region is
// then sorted by the remaining <= 8 high bits in tiles of TILE entries - one workgroup per tile, LDS counters, the region is
// tens of KB and stays in the L2; a heavy region (skewed scalars: unit scalars put every entry into one bucket) simply has
// more tiles, which meet through one global atomic per (tile, bucket).  The first tile of a region, knowing the final count of
// each of its buckets, also cuts them into pieces of <= SEG points (section 3 below) with piece ids that need no scan over the window:
//   pfirst(bucket) = w PW + bin NLO + floor(region_start / SEG) + (pieces of the earlier buckets of the region),   NLO = B / NBIN,
// disjoint between regions because sum ceil(c_i / SEG) <= NLO + floor(sum c_i / SEG), and < (w + 1) PW with PW = B + n / SEG + 1.
// Bucket of (bin, key): b = key << HIB | bin, HIB = log2(NBIN); runs of a window are laid out region by region, not by bucket
// index - nothing downstream assumes an order (pieces carry their own start).
template <class G>
__global__ void __launch_bounds__(1024) k_part_hist(const uint16_t* __restrict__ digits, uint32_t* __restrict__ blockcnt, uint32_t n,
                                                    uint32_t chunk, uint32_t NBIN, uint32_t wbase = 0) {
  // (wbase, here and in the four kernels below: the first window of the launch - the host-pointer pipeline sorts chunk 0's windows first)
  __shared__ uint32_t h[128];
  const uint32_t j = blockIdx.x, w = blockIdx.y + wbase, KB = gridDim.x;
  if (threadIdx.x < 128) h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t lo = j * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
  const uint16_t* dg = digits + (size_t)w * n;
  for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += 8 * 1024) {
    uint32_t d[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) { const uint32_t i = i0 + k * 1024; d[k] = i < hi ? dg[i] : 0xFFFFu; }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++)
      if (d[k] != 0xFFFFu) atomicAdd(&h[d[k] & (NBIN - 1)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < NBIN) blockcnt[((size_t)w * NBIN + threadIdx.x) * KB + j] = h[threadIdx.x];   // bin-major, block-minor
}
// exclusive scan of a window's NBIN x KB (<= 8192) block counts in place (-> each block's first position in each bin,
// window-relative), the NBIN + 1 region boundaries and the prefix of the regions' tile counts; one workgroup per window
template <class G>
__global__ void __launch_bounds__(1024) k_part_scan(uint32_t* __restrict__ blockcnt, uint32_t* __restrict__ binstart,
                                                    uint32_t* __restrict__ tileprefix, uint32_t NBIN, uint32_t KB, uint32_t TILE, uint32_t wbase = 0) {
  __shared__ uint32_t wave_tot[16], bs[129], tw[2];
  const uint32_t E = NBIN * KB, w = blockIdx.x + wbase;
  uint32_t* c = blockcnt + (size_t)w * E;
  const uint32_t PER = (E + 1023) / 1024;   // <= 8
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t v[8], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8; k++) {
    const uint32_t e = threadIdx.x * PER + k;
    v[k] = (k < PER && e < E) ? c[e] : 0;
    sum += v[k];
  }
  uint32_t x = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wave_tot[wv] = x;
  __syncthreads();
  uint32_t pre = 0;
  for (int k = 0; k < wv; k++) pre += wave_tot[k];
  uint32_t run = pre + x - sum;
#pragma unroll
  for (uint32_t k = 0; k < 8; k++) {
    const uint32_t e = threadIdx.x * PER + k;
    if (k < PER && e < E) {
      c[e] = run;
      if (e % KB == 0) bs[e / KB] = run;
      run += v[k];
    }
  }
  if (threadIdx.x == 1023) bs[NBIN] = pre + x;
  __syncthreads();
  if (threadIdx.x <= NBIN) binstart[w * (NBIN + 1) + threadIdx.x] = bs[threadIdx.x];
  uint32_t tiles = 0, ty = 0;
  if (threadIdx.x < 128) {   // two whole waves: exclusive scan of the regions' tile counts
    tiles = threadIdx.x < NBIN ? (bs[threadIdx.x + 1] - bs[threadIdx.x] + TILE - 1) / TILE : 0;
    ty = tiles;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t y2 = __shfl_up(ty, o, 64);
      if (lane >= o) ty += y2;
    }
    if (lane == 63) tw[wv] = ty;
  }
  __syncthreads();
  if (threadIdx.x < NBIN) {
    const uint32_t excl = ty - tiles + (wv == 1 ? tw[0] : 0u);
    tileprefix[w * (NBIN + 1) + threadIdx.x] = excl;
    if (threadIdx.x == NBIN - 1) tileprefix[w * (NBIN + 1) + NBIN] = excl + tiles;
  }
}
// Scattered 4-byte stores are bound by the L2's request rate (~128 per clock chip-wide: 2^24 entries = 0.1 ms however local
// the addresses), so both scatters stage a batch in LDS in output order and store it with consecutive lanes on consecutive
// addresses: a wave's store covers two or three runs instead of 64 lines.
template <class G>
__global__ void __launch_bounds__(1024) k_part_scatter(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ blockoff,
                                                       uint32_t* __restrict__ rec_idx, uint8_t* __restrict__ rec_key, uint32_t n,
                                                       uint32_t chunk, uint32_t HIB, uint32_t NBIN, const uint32_t* __restrict__ remap = nullptr,
                                                       uint32_t vw = 0, uint32_t wbase = 0) {
  // vw > 0: the windows are virtual (chunked layout, k_digits): window w holds index chunk w / vw, whose entries are the points from (w / vw) n on
  constexpr uint32_t SB = 8 * 1024;   // entries per batch
  __shared__ uint32_t cur[128], lcnt[128], loff[128], wt[2];
  __shared__ uint32_t st_idx[SB];
  __shared__ uint8_t st_key[SB], st_bin[SB];
  const uint32_t j = blockIdx.x, w = blockIdx.y + wbase, KB = gridDim.x, t = threadIdx.x;
  if (t < 128) cur[t] = t < NBIN ? blockoff[((size_t)w * NBIN + t) * KB + j] : 0u;
  const uint32_t lo = j * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
  const uint16_t* dg = digits + (size_t)w * n;
  uint32_t* oi = rec_idx + (size_t)w * n;
  uint8_t* ok = rec_key + (size_t)w * n;
  const int lane = t & 63, wv = t >> 6;
  const uint32_t ioff = vw ? (w / vw) * n : 0u;
  for (uint32_t i0 = lo; i0 < hi; i0 += SB) {
    uint32_t d[8], r[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) { const uint32_t i = i0 + k * 1024 + t; d[k] = i < hi ? dg[i] : 0xFFFFu; }
    if (t < 128) lcnt[t] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 8; k++)
      if (d[k] != 0xFFFFu) r[k] = atomicAdd(&lcnt[d[k] & (NBIN - 1)], 1u);
    __syncthreads();
    uint32_t c = 0, x = 0;
    if (t < 128) {   // two whole waves: exclusive scan of the batch's bin counts
      c = lcnt[t]; x = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t x2 = __shfl_up(x, o, 64);
        if (lane >= o) x += x2;
      }
      if (lane == 63) wt[wv] = x;
    }
    __syncthreads();
    if (t < 128) loff[t] = x - c + (wv == 1 ? wt[0] : 0u);
    const uint32_t valid = wt[0] + wt[1];
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 8; k++)
      if (d[k] != 0xFFFFu) {
        const uint32_t b = d[k] & 0x7FFFu, bin = b & (NBIN - 1);
        const uint32_t s_ = loff[bin] + r[k];
        const uint32_t e_ = i0 + k * 1024 + t;        // remap (fixed base, compacted virtual windows): slot e_ of window w holds table entry remap[w n + e_]
        st_idx[s_] = (remap ? remap[(size_t)w * n + e_] : e_ + ioff) | ((d[k] >> 15) << 31);
        st_key[s_] = (uint8_t)(b >> HIB);
        st_bin[s_] = (uint8_t)bin;
      }
    __syncthreads();
    for (uint32_t s_ = t; s_ < valid; s_ += 1024) {
      const uint32_t bin = st_bin[s_];
      const uint32_t dest = cur[bin] + (s_ - loff[bin]);
      oi[dest] = st_idx[s_];
      ok[dest] = st_key[s_];
    }
    __syncthreads();
    if (t < 128) cur[t] += c;
  }
}
// level 2.  Workgroup (x, w) is tile x of window w: region `bin` by binary search over the tile prefix, its z-th part (all of a
// lane's loads in flight at once).
constexpr uint32_t TILE_EPT = 10;   // entries per lane of a tile workgroup (registers); a tile holds up to TILE_EPT * blockDim entries
__device__ __forceinline__ bool tile_locate(const uint32_t* __restrict__ tp, uint32_t NBIN, uint32_t x, uint32_t& bin, uint32_t& z,
                                            uint32_t& tiles) {
  if (x >= tp[NBIN]) return false;
  uint32_t lo = 0, hi = NBIN;          // the bin with tp[bin] <= x < tp[bin + 1] (tp non-decreasing; empty regions repeat a value)
  while (hi - lo > 1) {
    const uint32_t m = (lo + hi) >> 1;
    if (tp[m] <= x) lo = m; else hi = m;
  }
  bin = lo; z = x - tp[lo]; tiles = tp[lo + 1] - tp[lo];
  return true;
}
// a region of rc entries is cut into `tiles` equal parts (the tile capacity leaves slack over the mean region, so that the usual
// region is ONE tile and not a full tile plus a sliver)
__device__ __forceinline__ void tile_range(uint32_t rs, uint32_t re, uint32_t z, uint32_t tiles, uint32_t& tile_lo, uint32_t& tile_n) {
  const uint32_t rc = re - rs, per = (rc + tiles - 1) / tiles;
  tile_lo = rs + z * per;
  tile_n = tile_lo >= re ? 0u : (re - tile_lo < per ? re - tile_lo : per);
}
template <class G>
__global__ void __launch_bounds__(1024) k_tile_count(const uint8_t* __restrict__ rec_key, const uint32_t* __restrict__ binstart,
                                                     const uint32_t* __restrict__ tileprefix, uint32_t* __restrict__ counts, uint32_t n,
                                                     uint32_t B, uint32_t HIB, uint32_t NBIN, uint32_t wbase = 0) {
  __shared__ uint32_t cnt[256];
  const uint32_t w = blockIdx.y + wbase, t = threadIdx.x, bd = blockDim.x;
  uint32_t bin, z, tiles, tile_lo, tile_n;
  if (!tile_locate(tileprefix + w * (NBIN + 1), NBIN, blockIdx.x, bin, z, tiles)) return;
  if (tiles == 1) return;      // a one-tile region (the usual case) is counted by its k_tile_sort workgroup itself
  tile_range(binstart[w * (NBIN + 1) + bin], binstart[w * (NBIN + 1) + bin + 1], z, tiles, tile_lo, tile_n);
  if (t < 256) cnt[t] = 0;
  __syncthreads();
  const uint8_t* kp = rec_key + (size_t)w * n + tile_lo;
  uint32_t key[TILE_EPT];
#pragma unroll
  for (uint32_t k = 0; k < TILE_EPT; k++) { const uint32_t e = k * bd + t; key[k] = e < tile_n ? kp[e] : 0xFFFFFFFFu; }
#pragma unroll
  for (uint32_t k = 0; k < TILE_EPT; k++)
    if (key[k] != 0xFFFFFFFFu) atomicAdd(&cnt[key[k]], 1u);
  __syncthreads();
  if (t < 256 && cnt[t]) atomicAdd(&counts[w * B + ((t << HIB) | bin)], cnt[t]);
}
// `cursor` (zeroed) hands every tile its offset inside each bucket's run: one global atomic per (tile, non-empty bucket)
template <class G>
__global__ void __launch_bounds__(1024) k_tile_sort(const uint32_t* __restrict__ rec_idx, const uint8_t* __restrict__ rec_key,
                                                    const uint32_t* __restrict__ binstart, const uint32_t* __restrict__ tileprefix,
                                                    uint32_t* __restrict__ counts, uint32_t* __restrict__ cursor,
                                                    uint32_t* __restrict__ sorted, uint32_t* __restrict__ pfirst,
                                                    uint32_t* __restrict__ pstart, uint32_t* __restrict__ plen, uint32_t* __restrict__ big,
                                                    uint32_t* __restrict__ nbig, uint32_t* __restrict__ mid, uint32_t* __restrict__ nmid,
                                                    uint32_t n, uint32_t B, uint32_t HIB, uint32_t NBIN, uint32_t SEG, uint32_t PW,
                                                    uint32_t* __restrict__ pbucket = nullptr, uint32_t vw = 0, uint32_t wbase = 0) {
  // pbucket (chunked layout): the FIRST piece of a bucket of virtual window w carries the sum of bucket (w mod vw, b) across the chunks
  // (k_accumulate_chunk): pbucket[piece] = (w mod vw) B + b for it, ~0 for the others
  __shared__ uint32_t cnt[256], cur[256], loff[256], wt[4], wt2[4], wt3[4], lists[4];
  __shared__ uint32_t st_idx[TILE_EPT * 1024];
  __shared__ uint8_t st_key[TILE_EPT * 1024];
  const uint32_t w = blockIdx.y + wbase, t = threadIdx.x, bd = blockDim.x, NLO = B >> HIB;
  uint32_t bin, z, tiles, tile_lo, tile_n;
  if (!tile_locate(tileprefix + w * (NBIN + 1), NBIN, blockIdx.x, bin, z, tiles)) return;
  const uint32_t rs = binstart[w * (NBIN + 1) + bin];
  tile_range(rs, binstart[w * (NBIN + 1) + bin + 1], z, tiles, tile_lo, tile_n);
  if (t < 256) cnt[t] = 0;
  if (t < 4) lists[t] = 0;
  uint32_t lrank = 0;
  __syncthreads();
  const uint8_t* kp = rec_key + (size_t)w * n + tile_lo;
  const uint32_t* ip = rec_idx + (size_t)w * n + tile_lo;
  uint32_t key[TILE_EPT], idx[TILE_EPT], r[TILE_EPT];
#pragma unroll
  for (uint32_t k = 0; k < TILE_EPT; k++) {
    const uint32_t e = k * bd + t;
    key[k] = e < tile_n ? kp[e] : 0xFFFFFFFFu;
    idx[k] = e < tile_n ? ip[e] : 0u;
  }
#pragma unroll
  for (uint32_t k = 0; k < TILE_EPT; k++)
    if (key[k] != 0xFFFFFFFFu) r[k] = atomicAdd(&cnt[key[k]], 1u);
  __syncthreads();
  uint32_t v = 0, p = 0, mine = 0, x = 0, y = 0, q = 0;
  const uint32_t g = w * B + ((t << HIB) | bin);      // bucket of thread t < NLO
  const int lane = t & 63, wv = t >> 6;
  if (t < 256) {   // four whole waves: exclusive scans of the region's bucket counts and piece counts and of the tile's own counts
    mine = cnt[t];
    v = t < NLO ? (tiles == 1 ? mine : counts[g]) : 0;     // the region's count: the tile's own if it is the only one
    p = (v + SEG - 1) / SEG;
    x = v; y = p; q = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t x2 = __shfl_up(x, o, 64), y2 = __shfl_up(y, o, 64), q2 = __shfl_up(q, o, 64);
      if (lane >= o) { x += x2; y += y2; q += q2; }
    }
    if (lane == 63) { wt[wv] = x; wt2[wv] = y; wt3[wv] = q; }
  }
  __syncthreads();
  if (t < 256) {
    uint32_t pre = 0, pre2 = 0, pre3 = 0;
    for (int k = 0; k < wv; k++) { pre += wt[k]; pre2 += wt2[k]; pre3 += wt3[k]; }
    loff[t] = pre3 + q - mine;
    if (t < NLO) {
      const uint32_t st = rs + pre + x - v;                                    // window-relative start of the bucket's run
      cur[t] = st + ((tiles > 1 && mine) ? atomicAdd(&cursor[g], mine) : 0u);
      if (tiles == 1) counts[g] = v;
      if (z == 0) {
        const uint32_t pf = w * PW + bin * NLO + rs / SEG + pre2 + y - p;
        pfirst[g] = pf;
        if (v) {
          const uint32_t s_ = w * n + st;
          for (uint32_t k = 0; k < p; k++) {
            pstart[pf + k] = s_ + k * SEG;
            plen[pf + k] = (v - k * SEG < SEG) ? v - k * SEG : SEG;
            if (pbucket) pbucket[pf + k] = k ? 0xFFFFFFFFu : (w % vw) * B + ((t << HIB) | bin);
          }
          // multi-piece buckets go on the fold lists: ranks from LDS, ONE global atomic per workgroup and list (every bucket of
          // a short top window is on the mid list - 4096 atomics on one address took 40 us)
          if (p > 16) lrank = atomicAdd(&lists[0], 1u) | 0x80000000u;
          else if (p > 1) lrank = atomicAdd(&lists[1], 1u) | 0x40000000u;
        }
      }
    }
  }
  __syncthreads();
  if (z == 0) {
    if (t < 2 && lists[t]) lists[2 + t] = atomicAdd(t == 0 ? nbig : nmid, lists[t]);
    __syncthreads();
    if (lrank & 0x80000000u) big[lists[2] + (lrank & 0x3FFFFFFFu)] = g;
    else if (lrank & 0x40000000u) mid[lists[3] + (lrank & 0x3FFFFFFFu)] = g;
  }
#pragma unroll
  for (uint32_t k = 0; k < TILE_EPT; k++)
    if (key[k] != 0xFFFFFFFFu) {
      const uint32_t s_ = loff[key[k]] + r[k];
      st_idx[s_] = idx[k];
      st_key[s_] = (uint8_t)key[k];
    }
  __syncthreads();
  uint32_t* out = sorted + (size_t)w * n;
  for (uint32_t s_ = t; s_ < tile_n; s_ += bd) {
    const uint32_t kk = st_key[s_];
    out[cur[kk] + (s_ - loff[kk])] = st_idx[s_];
  }
}

// ---- 3. work items.  A bucket's run is cut into pieces of at most SEG points so that no lane works much longer than
// the average (the top window of a 253-bit scalar has ~12 significant bits -> 16x fewer, 16x longer buckets; skewed
// inputs are worse).  Piece ids of bucket t: pfirst[t] .. pfirst[t] + ceil(count/SEG) - 1 (a static region per window).
// longest-first schedule of the pieces: counting sort by length (descending); zero-length slots are dropped
constexpr uint32_t SIZE_BINS = 2048;  // SEG < SIZE_BINS
template <class G>
__global__ void __launch_bounds__(256) k_size_hist(const uint32_t* __restrict__ plen, uint32_t* __restrict__ bins, uint32_t slots) {
  __shared__ uint32_t lh[SIZE_BINS];
  for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 256) lh[i] = 0;
  __syncthreads();
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < slots; t += gridDim.x * 256) {
    uint32_t c = plen[t];
    if (c) atomicAdd(&lh[SIZE_BINS - 1 - c], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 256)
    if (lh[i]) atomicAdd(&bins[i], lh[i]);
}
template <class G>
__global__ void __launch_bounds__(1024) k_size_scan(uint32_t* __restrict__ bins, uint32_t* __restrict__ nwork) {
  // in-place exclusive scan of SIZE_BINS (= 2 per thread) counters; nwork = number of non-empty pieces
  __shared__ uint32_t wave_tot[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t a = bins[2 * threadIdx.x], b = bins[2 * threadIdx.x + 1];
  uint32_t x = a + b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wave_tot[wv] = x;
  __syncthreads();
  uint32_t pre = 0;
  for (int k = 0; k < wv; k++) pre += wave_tot[k];
  uint32_t excl = pre + x - (a + b);
  bins[2 * threadIdx.x] = excl;
  bins[2 * threadIdx.x + 1] = excl + a;
  if (threadIdx.x == 1023) *nwork = pre + x;
}
template <class G>
__global__ void __launch_bounds__(1024) k_size_scatter(const uint32_t* __restrict__ plen, uint32_t* __restrict__ bins,
                                                       uint32_t* __restrict__ order, uint32_t slots) {
  // workgroup-aggregated: local ranks from LDS atomics, ONE global atomic per (workgroup, non-empty bin)
  __shared__ uint32_t lcnt[SIZE_BINS], lbase[SIZE_BINS];
  constexpr uint32_t PER = 4;
  for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 1024) lcnt[i] = 0;
  __syncthreads();
  uint32_t c[PER], r[PER];
  const uint32_t base = blockIdx.x * (1024 * PER);
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    uint32_t t = base + k * 1024 + threadIdx.x;
    c[k] = t < slots ? plen[t] : 0;
    r[k] = c[k] ? atomicAdd(&lcnt[SIZE_BINS - 1 - c[k]], 1u) : 0;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 1024)
    if (lcnt[i]) lbase[i] = atomicAdd(&bins[i], lcnt[i]);
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < PER; k++)
    if (c[k]) order[lbase[SIZE_BINS - 1 - c[k]] + r[k]] = base + k * 1024 + threadIdx.x;
}

// ---- 4. one lane per piece: XYZZ sum of its run of (signed) points
// Occupancy A/B (round 2, 2^20 terms): the 28-word fields (G2 of BLS12-377, BW6-761) take 256 VGPRs + ~160 AGPRs = ONE wave per
// SIMD.  Forcing two (-DCELO_ACC_OCC2: 520-744 B/lane of scratch instead of the AGPRs) is SLOWER - G2 8.69 -> 10.0 ms, BW6-761
// 16.0 -> 17.1 ms - because the one-wave kernels already issue an instruction every 5.1-5.3 cycles (the v_mad_u64_u32 rate:
// ~16-19k instructions per mixed addition x 2^20 x windows / 1024 SIMDs): their instruction stream has the independent work a
// second wave would bring.  What is left is the instruction count.
#ifdef CELO_ACC_OCC2
#define ACC_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define ACC_OCC
#endif
template <class G>
__global__ void __launch_bounds__(256) ACC_OCC k_accumulate(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ plen,
                                                    const uint32_t* __restrict__ order, const uint32_t* __restrict__ nwork,
                                                    uint32_t* __restrict__ partials) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= *nwork) return;
  uint32_t pid = order[tid];
  const uint32_t* run = sorted + pstart[pid];
  uint32_t len = plen[pid];
  Xyzz<F> acc = Xyzz<F>::identity();
  if constexpr (sizeof(F) <= 14 * sizeof(uint32_t)) {
    // 14-limb field: the first two points of the run are added as affine + affine (4 products + 2 squares instead of a mixed
    // addition's 8 + 2).  Same-box A/B (tools/ab_cmd_msm.sh, two rounds): G1 accumulate 0.392 / 0.393 -> 0.376 / 0.377 ms at 2^17
    // (runs of 8 points); at 2^20 (runs of 32) it is inside the run-to-run spread: 2.54 / 2.51 -> 2.56 / 2.51 ms.
    auto point = [&](uint32_t k) {
      const uint32_t v = run[k];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      if (v >> 31) p = affine_neg(p);
      return p;
    };
    uint32_t k0 = 0;
    if (len >= 2) { acc = xyzz_add_affine(point(0), point(1)); k0 = 2; }
    // (prefetching the next point, as the one-wave kernels below do, measures nothing here: two waves per SIMD hide the loads)
    for (uint32_t k = k0; k < len; k++) xyzz_madd(acc, point(k));
  } else {
    // the one-wave-per-SIMD kernels of the 28-word fields do not take the affine start: with the second inlined body they lose
    // (G2 8.15 -> 8.39 ms at 2^20, BW6-761 14.52 -> 14.72), and BW6-761 once lost 7 % to a mere restructuring of this loop
#ifdef CELO_ACC_NO_PREFETCH
    for (uint32_t k = 0; k < len; k++) {
      uint32_t v = run[k];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      if (v >> 31) p = affine_neg(p);
      xyzz_madd(acc, p);
    }
#else
    // one wave per SIMD: nothing else hides the two dependent global loads (index, then the point it names) at the head of an
    // iteration - the next point is fetched before the current addition starts (it waits in AGPRs: 169 -> 222 for G2, no scratch).
    // Same-box A/B (tools/ab_cmd_msm.sh): G2 8.32 -> 8.21 ms at 2^20, 1.186 -> 1.148 at 2^17, config 3 27.97 -> 27.71 ms;
    // BW6-761 unchanged (14.70 vs 14.70 ms).  -DCELO_ACC_NO_PREFETCH restores the plain loop.
    if (len) {
      uint32_t v = run[0];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      for (uint32_t k = 0; k < len; k++) {
        const uint32_t vn = run[k + 1 < len ? k + 1 : k];
        const Affine<F> pn = IO::load_affine(bases + (size_t)(vn & 0x7fffffffu) * IO::AFF_WORDS);
        if (v >> 31) p = affine_neg(p);
        xyzz_madd(acc, p);
        p = pn;
        v = vn;
      }
    }
#endif
  }
  IO::store_xyzz(partials + (size_t)pid * IO::XYZZ_WORDS, acc);
}

// ---- 4b. The accumulation over Fq2 on LANE PAIRS (late round 5).  k_accumulate<G2_377> holds a whole Fq2 mixed addition per lane: 256 VGPRs
// + 220 AGPRs, one wave per SIMD, an instruction every 5.1-5.7 cycles where the two-wave G1 kernel issues one every 4.0-4.2 (a lone wave
// cannot issue back to back; DESIGN.md section 4 showed it is not the dependent multiply-add chains).  Here the two halves c0, c1 of
// every Fq2 value sit on two ADJACENT lanes (half = lane & 1): a lane holds half the state (225 VGPRs, no AGPRs: two waves per SIMD) and
// an Fq2 product is QHex377::mul - the six-lane pairings' pair product: one signed two-product Montgomery pass per lane (Fp::mul2s,
// 2 x 196 + 182 multiply-adds: the same count as a half of the one-lane Fq2 product) after one DPP exchange with lane ^ 1.  Squarings and
// Y3 = R t - Y1 PPP are plain pair products here (10 per addition against 6 + 2 squarings + the fused Y3 of curve.h: 11 % more
// multiply-adds).  As register-resident loops (tools/ubench_g2_pair.hip) the pair form runs 2.54 G additions/s against 2.20 G/s; in the
// MSM the gain is 2-3 % alone on the device and 2-5 % beside other kernels (launch_accumulate below): the default; CELO_G2_PAIR=0 restores
// the one-lane kernel.  Same formulas (madd-2008-s, mdbl-2008-s-1), same stored bounds as curve.h (X < 19 p, Y < 7 p, ZZ, ZZZ
// < 3 p, limbs normalised: every hex:: operation carries), same partials layout: the reduction does not know which kernel ran.
// Control flow is PAIR-UNIFORM: both lanes of a pair walk the same piece and take the same branches (the exact-zero tests AND the two
// halves through DPP), so the partner lane is always there for the exchange.
struct PairAcc377 { Fp<P377> X, Y, ZZ, ZZZ; bool inf; };
__device__ __forceinline__ bool pair_both(bool z) {
  const int zi = z ? 1 : 0;
  return (zi & __builtin_amdgcn_mov_dpp(zi, 0xB1, 0xF, 0xF, true)) != 0;
}
// QHex377::mul in two steps: the first operand's exchanged form (X = the even lane's half in both lanes, cs = the odd lane's half times -5 | 1)
struct PairFirst {
  Fp<P377> X; int32_t cs[14];
  __device__ __forceinline__ explicit PairFirst(const Fp<P377>& a) {
    const uint32_t k = QHex377::hsel() ? 1u : 0u - 5u;
#pragma unroll
    for (int i = 0; i < 14; i++) {
      X.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xA0, 0xF, 0xF, true);
      cs[i] = (int32_t)((uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xF5, 0xF, 0xF, true) * k);
    }
  }
};
__device__ __forceinline__ Fp<P377> pair_mul(const PairFirst& f, const Fp<P377>& b, const Fp<P377>& bo) { return Fp<P377>::mul2s(f.X, b, f.cs, bo); }
__device__ __forceinline__ void pair_dbl_affine(PairAcc377& a, const Fp<P377>& px, const Fp<P377>& py) {
  typedef QHex377 Q;
  typedef Fp<P377> H;
  if (pair_both(py.is_zero_mod_p())) { a.inf = true; return; }
  const H U = Q::dbl(py);
  const H V = Q::mul(U, U), W = Q::mul(U, V), S = Q::mul(px, V), xx = Q::mul(px, px);
  const H M = Q::tpl(xx);
  const H X3 = Q::template sub<16>(Q::mul(M, M), Q::dbl(S));
  const H t = Q::template sub<32>(S, X3);
  a.Y = Q::template sub<4>(Q::mul(M, t), Q::mul(W, py));
  a.X = X3; a.ZZ = V; a.ZZZ = W; a.inf = false;
}
template <int V> __device__ __forceinline__ void pair_madd(PairAcc377& a, const Fp<P377>& px, const Fp<P377>& py) {
  typedef QHex377 Q;
  typedef Fp<P377> H;
  if (a.inf) { a.X = px; a.Y = py; a.ZZ = Q::one(); a.ZZZ = Q::one(); a.inf = false; return; }
  const H U2 = Q::mul(px, a.ZZ), S2 = Q::mul(py, a.ZZZ);
  const H Pd = Q::template sub<32>(U2, a.X), R = Q::template sub<16>(S2, a.Y);      // X < 19 p, Y < 7 p
  if (pair_both(Pd.is_zero_mod_p())) {
    if (pair_both(R.is_zero_mod_p())) pair_dbl_affine(a, px, py);
    else a.inf = true;
    return;
  }
  if constexpr (V == 0) {
    const H PP = Q::mul(Pd, Pd), PPP = Q::mul(Pd, PP), Qv = Q::mul(a.X, PP), R2 = Q::mul(R, R);
    const H X3 = Q::template sub<16>(R2, Q::add(Q::add(PPP, Qv), Qv));
    const H t = Q::template sub<32>(Qv, X3);
    a.Y = Q::template sub<4>(Q::mul(R, t), Q::mul(a.Y, PPP));
    a.ZZ = Q::mul(a.ZZ, PP);
    a.ZZZ = Q::mul(a.ZZZ, PPP);
    a.X = X3;
  } else {
    // the exchanged forms of operands that enter several products are built once: Pd and R as first operands (the even lane's half and the
    // scaled odd lane's half: 28 DPP moves + 14 multiplications by -5 | 1 each), PP and PPP as second operands (the partner's half: 14 DPP moves)
    const PairFirst fPd(Pd), fR(R);
    const H PP = pair_mul(fPd, Pd, Q::swap(Pd));
    const H PPo = Q::swap(PP);
    const H PPP = pair_mul(fPd, PP, PPo);
    const H PPPo = Q::swap(PPP);
    const H Qv = pair_mul(PairFirst(a.X), PP, PPo);
    const H R2 = pair_mul(fR, R, Q::swap(R));
    const H X3 = Q::template sub<16>(R2, Q::add(Q::add(PPP, Qv), Qv));
    const H t = Q::template sub<32>(Qv, X3);
    a.Y = Q::template sub<4>(pair_mul(fR, t, Q::swap(t)), pair_mul(PairFirst(a.Y), PPP, PPPo));
    a.ZZ = pair_mul(PairFirst(a.ZZ), PP, PPo);
    a.ZZZ = pair_mul(PairFirst(a.ZZZ), PPP, PPPo);
    a.X = X3;
  }
}
template <class G, int V>      // G = G2_377 (a template so that every translation unit that launches it owns an instantiation); V: see pair_madd
__global__ void __launch_bounds__(256) k_accumulate_pair(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ plen,
                                                            const uint32_t* __restrict__ order, const uint32_t* __restrict__ nwork,
                                                            uint32_t* __restrict__ partials) {
  static_assert(std::is_same<G, G2_377>::value, "lane pairs: Fq2 of BLS12-377");
  typedef Fp<P377> H;
  typedef PointIO<Fp2<P377>> IO;
  constexpr int HW = H::WORDS;                 // words per half; an Fq2 coordinate is c0 | c1
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t piece = tid >> 1, h = tid & 1;
  if (piece >= *nwork) return;
  const uint32_t pid = order[piece];
  const uint32_t* run = sorted + pstart[pid];
  const uint32_t len = plen[pid];
  PairAcc377 acc;
  acc.X = H::zero(); acc.Y = H::zero(); acc.ZZ = H::zero(); acc.ZZZ = H::zero(); acc.inf = true;
  if constexpr (V == 1) {
    for (uint32_t k = 0; k < len; k++) {
      const uint32_t v = run[k];
      const uint32_t* q = bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS + h * HW;
      const H px = H::load(q);
      H py = H::load(q + IO::FW);
      if (v >> 31) py = QHex377::template neg<4>(py);
      pair_madd<V>(acc, px, py);
    }
  } else if (len) {
    // the next point is fetched before the current addition starts: the two dependent loads (index, then the point it names) at the head of an
    // iteration are a larger share of a HALF addition than of a whole one
    uint32_t v = run[0];
    const uint32_t* q = bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS + h * HW;
    H px = H::load(q), py = H::load(q + IO::FW);
    for (uint32_t k = 0; k < len; k++) {
      const uint32_t vn = run[k + 1 < len ? k + 1 : k];
      const uint32_t* qn = bases + (size_t)(vn & 0x7fffffffu) * IO::AFF_WORDS + h * HW;
      const H nx = H::load(qn), ny = H::load(qn + IO::FW);
      if (v >> 31) py = QHex377::template neg<4>(py);
      pair_madd<V>(acc, px, py);
      px = nx; py = ny; v = vn;
    }
  }
  uint32_t* d = partials + (size_t)pid * IO::XYZZ_WORDS + h * HW;
  if (acc.inf) { acc.X = H::zero(); acc.Y = H::zero(); acc.ZZ = H::zero(); acc.ZZZ = H::zero(); }      // the identity is stored as exact zeros
  acc.X.store(d); acc.Y.store(d + IO::FW); acc.ZZ.store(d + 2 * IO::FW); acc.ZZZ.store(d + 3 * IO::FW);
}
// the launch of the accumulation: lane pairs for G2 of BLS12-377 (CELO_G2_PAIR=0: the one-lane kernel), one lane per piece otherwise
template <class G>
inline void launch_accumulate(uint32_t slots, hipStream_t stream, const uint32_t* d_bases, const uint32_t* d_sorted, const uint32_t* d_pstart, const uint32_t* d_plen,
                              const uint32_t* d_order, const uint32_t* d_nwork, uint32_t* d_partials) {
  if constexpr (std::is_same<G, G2_377>::value) {
    // 2 (default): lane pairs with the exchanged operand forms built once; 1: lane pairs, plain products with the next point prefetched;
    // 0: the one-lane kernel.  Same-box A/Bs (profiles/r5_ab_g2_lane_pairs.txt): one G2 MSM of 2^20 terms alone on the device 7.83-7.89 ms
    // one lane, 7.66-7.68 (1), 7.59-7.66 (2) - the pair kernels issue a VALU instruction every 3.95 cycles (the limit) where the one-lane
    // kernel issues one every 4.6, and need 12.5 % more of them (ten pair products of 574 multiply-adds per lane against six products, two
    // squarings and the fused Y3 of curve.h); BESIDE other kernels a two-wave kernel with half the registers shares the device better:
    // config 3 (the key leg beside the signature leg) 24.23-24.29 -> 23.67-23.84 ms (five alternations), config 5 (G2 beside G1 and the
    // pairings) 50.5-52.2 -> 48.9-49.0 ms.  The whole -m gpu suite passes on either.
    static const int pair = getenv("CELO_G2_PAIR") ? atoi(getenv("CELO_G2_PAIR")) : 2;
    if (pair == 2) {
      hipLaunchKernelGGL((k_accumulate_pair<G, 1>), dim3((2 * slots + 255) / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_partials);
      return;
    }
    if (pair) {
      hipLaunchKernelGGL((k_accumulate_pair<G, 0>), dim3((2 * slots + 255) / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_partials);
      return;
    }
  }
  hipLaunchKernelGGL((k_accumulate<G>), dim3((slots + 255) / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_partials);
}

// The accumulation of ONE INDEX CHUNK of the host-pointer pipeline (round 5; run_device_windows' HostIn): the bases arrive over PCIe chunk
// by chunk and every chunk is accumulated while the next one is in flight.  The sort ran over (chunk, window) virtual windows, so a
// bucket's points of chunk k are a run of their own, cut into pieces as usual; the lane of a bucket's FIRST piece starts from the
// bucket's carried sum - the value the same bucket reached in the earlier chunks (carrier[(w, b)], zeroes = the identity before chunk 0) -
// and stores it back, so the chunks cost no additions that one pass over all n points would not have spent.  Further pieces of a bucket
// (runs longer than SEG: skewed scalars) go to `partials` as in k_accumulate and are folded into the carrier after the last chunk
// (k_combine_* with first = 1, k_merge_carried).  Chunk launches are stream-ordered: no two lanes ever hold the same carrier.
template <class G>
__global__ void __launch_bounds__(256) ACC_OCC k_accumulate_chunk(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ plen,
                                                    const uint32_t* __restrict__ order, const uint32_t* __restrict__ nwork,
                                                    uint32_t* __restrict__ partials, const uint32_t* __restrict__ pbucket,
                                                    uint32_t* __restrict__ carrier, uint32_t cont) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= *nwork) return;
  uint32_t pid = order[tid];
  const uint32_t* run = sorted + pstart[pid];
  uint32_t len = plen[pid];
  const uint32_t pb = pbucket[pid];
  uint32_t* dst = pb != 0xFFFFFFFFu ? carrier + (size_t)pb * IO::XYZZ_WORDS : partials + (size_t)pid * IO::XYZZ_WORDS;
  const bool carried = cont && pb != 0xFFFFFFFFu;
  Xyzz<F> acc = Xyzz<F>::identity();
  if (carried) acc = IO::load_xyzz(dst);
  if constexpr (sizeof(F) <= 14 * sizeof(uint32_t)) {
    auto point = [&](uint32_t k) {
      const uint32_t v = run[k];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      if (v >> 31) p = affine_neg(p);
      return p;
    };
    uint32_t k0 = 0;
    if (!carried && len >= 2) { acc = xyzz_add_affine(point(0), point(1)); k0 = 2; }
    for (uint32_t k = k0; k < len; k++) xyzz_madd(acc, point(k));
  } else {
    if (len) {
      uint32_t v = run[0];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      for (uint32_t k = 0; k < len; k++) {
        const uint32_t vn = run[k + 1 < len ? k + 1 : k];
        const Affine<F> pn = IO::load_affine(bases + (size_t)(vn & 0x7fffffffu) * IO::AFF_WORDS);
        if (v >> 31) p = affine_neg(p);
        xyzz_madd(acc, p);
        p = pn;
        v = vn;
      }
    }
  }
  IO::store_xyzz(dst, acc);
}
// after the last chunk: carrier(w, b) += the folded further pieces of bucket (w, b) of every chunk (piece pfirst + 1 of virtual window
// k vw + w, where its run was longer than SEG).  One lane per bucket; for uniform scalars almost no lane has anything to add.
template <class G>
__global__ void __launch_bounds__(128) k_merge_carried(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pfirst,
                                                       const uint32_t* __restrict__ partials, uint32_t* __restrict__ carrier,
                                                       uint32_t real_total, uint32_t chunks, uint32_t SEG) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= real_total) return;
  bool any = false;
  for (uint32_t k = 0; k < chunks; k++) any |= counts[(size_t)k * real_total + t] > SEG;
  if (!any) return;
  Xyzz<F> acc = IO::load_xyzz(carrier + (size_t)t * IO::XYZZ_WORDS);
  for (uint32_t k = 0; k < chunks; k++) {
    const size_t g = (size_t)k * real_total + t;
    if (counts[g] > SEG) {
      const Xyzz<F> v = IO::load_xyzz(partials + (size_t)(pfirst[g] + 1) * IO::XYZZ_WORDS);
      xyzz_add_fn(acc, v);
    }
  }
  IO::store_xyzz(carrier + (size_t)t * IO::XYZZ_WORDS, acc);
}

// ---- 5a. buckets cut into 2..16 pieces (e.g. every bucket of a short top window): one lane folds the pieces
// (`first` = 1, chunked pipeline: the first piece is the bucket's carrier and stays out of the fold - the pieces from the second on are
// folded into the second)
template <class G>
__global__ void __launch_bounds__(128) k_combine_mid(const uint32_t* __restrict__ mid, const uint32_t* __restrict__ nmid,
                                                     const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pfirst,
                                                     uint32_t* __restrict__ partials, uint32_t* __restrict__ pieces_of, uint32_t SEG, uint32_t first = 0) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < *nmid; q += gridDim.x * blockDim.x) {
    uint32_t t = mid[q];
    uint32_t pc = (counts[t] + SEG - 1) / SEG - first, pf = pfirst[t] + first;
    Xyzz<F> acc = IO::load_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS);
    for (uint32_t k = 1; k < pc; k++) {
      Xyzz<F> v = IO::load_xyzz(partials + (size_t)(pf + k) * IO::XYZZ_WORDS);
      xyzz_add_fn(acc, v);
    }
    IO::store_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS, acc);
    pieces_of[t] = 1;
  }
}
// the same fold with three lanes per bucket (curve_lanes.h): the folds are a handful of dependent additions on lone waves (the
// 4096 buckets of a short top window, four pieces each), i.e. latency - see k_bitsum_lanes
template <class G>
__global__ void __launch_bounds__(64) k_combine_mid_lanes(const uint32_t* __restrict__ mid, const uint32_t* __restrict__ nmid,
                                                          const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pfirst,
                                                          uint32_t* __restrict__ partials, uint32_t* __restrict__ pieces_of, uint32_t SEG, uint32_t first = 0) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  typedef QTriT<FieldBase<F>> QB;
  typedef LanePoint<QB> LP;
  const int g = QB::group();
  if (g >= 21) return;
  for (uint32_t q = blockIdx.x * 21u + (uint32_t)g; q < *nmid; q += gridDim.x * 21u) {
    const uint32_t t = mid[q];
    const uint32_t pc = (counts[t] + SEG - 1) / SEG - first, pf = pfirst[t] + first;
    const Xyzz<F> a = IO::load_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS);
    typename LP::Pt acc = {{a.X, a.Y, a.ZZ, a.ZZZ}, a.is_identity()};
    for (uint32_t k = 1; k < pc; k++) {
      const Xyzz<F> v = IO::load_xyzz(partials + (size_t)(pf + k) * IO::XYZZ_WORDS);
      const typename LP::P pb = {v.X, v.Y, v.ZZ, v.ZZZ};
      LP::add(acc, pb, v.is_identity());
    }
    if (QB::lane() == 0) {
      const Xyzz<F> r = acc.inf ? Xyzz<F>::identity() : Xyzz<F>{acc.p.X, acc.p.Y, acc.p.ZZ, acc.p.ZZZ};
      IO::store_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS, r);
      pieces_of[t] = 1;
    }
  }
}

// ---- 5b. buckets cut into many pieces (skewed inputs): one workgroup per such bucket folds its pieces into the first
template <class G>
__global__ void __launch_bounds__(256) k_combine_big(const uint32_t* __restrict__ big, const uint32_t* __restrict__ nbig,
                                                     const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pfirst,
                                                     uint32_t* __restrict__ partials, uint32_t* __restrict__ pieces_of, uint32_t SEG, uint32_t first = 0) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  __shared__ uint32_t stage[64 * IO::XYZZ_WORDS];
  for (uint32_t q = blockIdx.x; q < *nbig; q += gridDim.x) {
    uint32_t t = big[q];
    uint32_t pc = (counts[t] + SEG - 1) / SEG - first, pf = pfirst[t] + first;
    Xyzz<F> acc = Xyzz<F>::identity();
    for (uint32_t k = threadIdx.x; k < pc; k += 256) {
      Xyzz<F> v = IO::load_xyzz(partials + (size_t)(pf + k) * IO::XYZZ_WORDS);
      xyzz_add_fn(acc, v);
    }
    // fold 256 -> 64 -> 1 through LDS (64 slots)
    for (uint32_t width = 256; width > 1; width >>= 2) {
      uint32_t q4 = width >> 2;
      for (uint32_t r = 1; r < 4; r++) {
        __syncthreads();
        if (threadIdx.x >= r * q4 && threadIdx.x < (r + 1) * q4) IO::store_xyzz(stage + (threadIdx.x - r * q4) * IO::XYZZ_WORDS, acc);
        __syncthreads();
        if (threadIdx.x < q4) { Xyzz<F> v = IO::load_xyzz(stage + threadIdx.x * IO::XYZZ_WORDS); xyzz_add_fn(acc, v); }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { IO::store_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS, acc); pieces_of[t] = 1; }
    __syncthreads();
  }
}

template <class G> HD Xyzz<typename G::F> load_bucket(const uint32_t* partials, const uint32_t* counts, const uint32_t* pfirst,
                                                       const uint32_t* pieces_of, uint32_t t, uint32_t SEG) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  if (!counts) return IO::load_xyzz(partials + (size_t)t * IO::XYZZ_WORDS);    // chunked pipeline: `partials` is the carrier table, one slot per bucket
  uint32_t c = counts[t];
  if (c == 0) return Xyzz<F>::identity();
  (void)pieces_of; (void)SEG;  // k_combine_mid / k_combine_big have folded multi-piece buckets into their first piece
  return IO::load_xyzz(partials + (size_t)pfirst[t] * IO::XYZZ_WORDS);
}

// ---- bucket reduction: window sum S = sum_b (b + 1) B_b with no scalar multiplication and depth log2(B).
// Binary tree over the bucket index: node(l, p) = sum of the buckets whose top l index bits are p (leaves at level LB).
// Bit k = LB - l of b is set exactly for the leaves under the odd-indexed nodes of level l, so
//   S = node(0, 0) + sum_{l=1..LB} 2^(LB - l) O_l,   O_l = sum_{p odd} node(l, p).
// Launch t builds level LB - t from level LB - t + 1 and halves every pending odd list once; a list is born strided
// (its first halving reads nodes 4i+1 and 4i+3 of its level).  LB launches, 2 point additions of work per bucket (the same
// as a running sum), every addition independent of the others of its launch: the depth of the whole reduction is LB
// additions instead of 16 (running sum) + ~18 (fix-up scalar) + log2 (tree).  The 2^(LB-l) weights are applied by the host
// inside the Horner recombination it runs anyway (one addition per doubling).
struct BitsumJobs {
  static constexpr int MAXJ = 20;
  uint32_t njobs;
  uint32_t end[MAXJ];    // cumulative number of outputs
  uint32_t src[MAXJ];    // point index into the work area (modes 0, 1, 4); unused for the leaf modes
  uint32_t dst[MAXJ];    // point index into the work area
  uint32_t mode[MAXJ];   // 0: in[2i] + in[2i+1]   1: in[4i+1] + in[4i+3]   2, 3: the same on the buckets themselves   4: in[2i+1]
};
template <class G>
__global__ void __launch_bounds__(128) k_bitsum(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ counts,
                                                const uint32_t* __restrict__ pfirst, const uint32_t* __restrict__ pieces_of, uint32_t SEG,
                                                uint32_t* __restrict__ work, BitsumJobs jobs) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= jobs.end[jobs.njobs - 1]) return;
  uint32_t j = 0;
  while (t >= jobs.end[j]) j++;
  const uint32_t i = t - (j ? jobs.end[j - 1] : 0u);
  const uint32_t mode = jobs.mode[j];
  Xyzz<F> a, b;
  if (mode == 2) {
    a = load_bucket<G>(partials, counts, pfirst, pieces_of, 2 * i, SEG);
    b = load_bucket<G>(partials, counts, pfirst, pieces_of, 2 * i + 1, SEG);
  } else if (mode == 3) {
    a = load_bucket<G>(partials, counts, pfirst, pieces_of, 4 * i + 1, SEG);
    b = load_bucket<G>(partials, counts, pfirst, pieces_of, 4 * i + 3, SEG);
  } else {
    const uint32_t* in = work + (size_t)jobs.src[j] * IO::XYZZ_WORDS;
    const size_t ia = mode == 0 ? 2 * (size_t)i : mode == 1 ? 4 * (size_t)i + 1 : 2 * (size_t)i + 1;
    a = IO::load_xyzz(in + ia * IO::XYZZ_WORDS);
    if (mode != 4) b = IO::load_xyzz(in + (ia + (mode == 0 ? 1 : 2)) * IO::XYZZ_WORDS);
  }
  if (mode != 4) xyzz_add(a, b);     // inlined for every field: each launch is one addition deep, its latency is the cost
  IO::store_xyzz(work + ((size_t)jobs.dst[j] + i) * IO::XYZZ_WORDS, a);
}
// The same launch with THREE LANES PER ADDITION (curve_lanes.h): the late levels of the reduction hold fewer additions than the
// chip has SIMDs, each launch costs the latency of one addition on a lone wave (~20 us for G1: 14 dependent-ish products), and
// spreading an addition's independent products over a lane group cuts that chain to 5 product rounds.  Used once a launch has
// few enough outputs that the tripled lane count still leaves every wave a SIMD of its own.
template <class G>
__global__ void __launch_bounds__(64) k_bitsum_lanes(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ counts,
                                                     const uint32_t* __restrict__ pfirst, const uint32_t* __restrict__ pieces_of, uint32_t SEG,
                                                     uint32_t* __restrict__ work, BitsumJobs jobs) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  typedef QTriT<FieldBase<F>> QB;
  typedef LanePoint<QB> LP;
  const int g = QB::group();
  const uint32_t t = blockIdx.x * 21u + (uint32_t)g;
  if (g >= 21 || t >= jobs.end[jobs.njobs - 1]) return;
  uint32_t j = 0;
  while (t >= jobs.end[j]) j++;
  const uint32_t i = t - (j ? jobs.end[j - 1] : 0u);
  const uint32_t mode = jobs.mode[j];
  Xyzz<F> a, b = Xyzz<F>::identity();
  if (mode == 2) {
    a = load_bucket<G>(partials, counts, pfirst, pieces_of, 2 * i, SEG);
    b = load_bucket<G>(partials, counts, pfirst, pieces_of, 2 * i + 1, SEG);
  } else if (mode == 3) {
    a = load_bucket<G>(partials, counts, pfirst, pieces_of, 4 * i + 1, SEG);
    b = load_bucket<G>(partials, counts, pfirst, pieces_of, 4 * i + 3, SEG);
  } else {
    const uint32_t* in = work + (size_t)jobs.src[j] * IO::XYZZ_WORDS;
    const size_t ia = mode == 0 ? 2 * (size_t)i : mode == 1 ? 4 * (size_t)i + 1 : 2 * (size_t)i + 1;
    a = IO::load_xyzz(in + ia * IO::XYZZ_WORDS);
    if (mode != 4) b = IO::load_xyzz(in + (ia + (mode == 0 ? 1 : 2)) * IO::XYZZ_WORDS);
  }
  typename LP::Pt acc = {{a.X, a.Y, a.ZZ, a.ZZZ}, a.is_identity()};
  if (mode != 4) {
    const typename LP::P pb = {b.X, b.Y, b.ZZ, b.ZZZ};
    LP::add(acc, pb, b.is_identity());
  }
  if (QB::lane() != 0) return;
  const Xyzz<F> r = acc.inf ? Xyzz<F>::identity() : Xyzz<F>{acc.p.X, acc.p.Y, acc.p.ZZ, acc.p.ZZZ};
  IO::store_xyzz(work + ((size_t)jobs.dst[j] + i) * IO::XYZZ_WORDS, r);
}
// the final results (node(0,0) and the O_l of every window), in place: device form -> arkworks limbs for the host's 64-bit
// Horner pass (host64.h).  Its own tiny launch: inside k_bitsum the conversion doubled the register count of every level.
template <class G>
__global__ void __launch_bounds__(64) k_results_to_ark(uint32_t* __restrict__ work, uint32_t res_pts) {
  // one lane per COORDINATE (the launch is a single conversion deep); a point's four lanes sit in one workgroup, and every load of
  // the workgroup precedes its stores: the arkworks form is shorter, so a coordinate's output overlaps its neighbour's input
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, pt = t >> 2, q = t & 3;
  const bool live = pt < res_pts;
  uint64_t out[IO::ARK64];
  if (live) F::load(work + (size_t)pt * IO::XYZZ_WORDS + q * IO::FW).to_ark(out);   // exact zeros (the identity's ZZ) stay exact zeros
  __syncthreads();
  if (live) {
    uint64_t* o = reinterpret_cast<uint64_t*>(work + (size_t)pt * IO::XYZZ_WORDS) + q * IO::ARK64;
#pragma unroll
    for (int i = 0; i < IO::ARK64; i++) o[i] = out[i];
  }
}

// =====================================================================================================================
// Batched small MSMs: m independent instances (instance p owns points [offsets[p], offsets[p+1])), the shape of
// bls-crypto's Batch::verify (crates/bls-crypto/src/bls/batch.rs:69,76: one n-term G2 MSM + one n-term G1 MSM per batch,
// n = number of signers, a few hundred) when bls-snark-sys' batch_verify_strict (crates/bls-snark-sys/src/signatures.rs:358)
// hands over thousands of batches.  Per-instance Pippenger with a small window; the instances are the parallel axis.
//   k_batch_sort      one workgroup per instance: signed digits, per-window LDS counting sort, runs written to HBM
//   k_size_* + k_accumulate (shared with the big path): every (instance, window, bucket) run is a work item, longest first
//   k_batch_reduce    one lane per (instance, window): running sum over its <= 64 buckets
//   k_batch_horner_lanes  three lanes per instance: Horner over the windows, Jacobian result in arkworks form
// OR of every scalar, limb by limb: the batch path sizes its window count by the longest scalar actually present (Batch::verify
// hands over 136-bit exponents in 253-bit containers: 28 windows of 5 bits instead of 51, and no idle lanes in the per-window
// kernels)
// (round 4: 16-byte loads on a grid that covers the chip several times over - the 65 536-lane, 4-byte-load version took 0.43 ms for the
// 32 MB of config 3, on the critical path of both legs of every batch_verify call; limb q of a scalar sits in lane group q / 4)
template <int SW>
__global__ void __launch_bounds__(256) k_scalar_or(const uint32_t* __restrict__ scalars, size_t words, uint32_t* __restrict__ out) {
  static_assert(SW % 4 == 0, "scalars are whole 16-byte groups");
  constexpr size_t G4 = SW / 4;
  const size_t quads = words / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x / G4 * G4;    // a multiple of G4: a lane only ever sees one 16-byte group of the scalar
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4* q = (const uint4*)scalars;
  uint4 acc = {0u, 0u, 0u, 0u};
  for (i = i < stride ? i : quads; i < quads; i += stride) { const uint4 v = q[i]; acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w; }
  __shared__ uint32_t blk[SW];                                      // per block in LDS first: SW global atomics per block, not 4 per lane
  if (threadIdx.x < SW) blk[threadIdx.x] = 0;
  __syncthreads();
  uint32_t* o = blk + 4 * (((size_t)blockIdx.x * blockDim.x + threadIdx.x) % G4);
  if (acc.x) atomicOr(o, acc.x);
  if (acc.y) atomicOr(o + 1, acc.y);
  if (acc.z) atomicOr(o + 2, acc.z);
  if (acc.w) atomicOr(o + 3, acc.w);
  __syncthreads();
  if (threadIdx.x < SW && blk[threadIdx.x]) atomicOr(out + threadIdx.x, blk[threadIdx.x]);
}
template <int SW, int CB, int PT>
__global__ void __launch_bounds__(256) k_batch_sort(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf,
                                                    const uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted,
                                                    uint32_t* __restrict__ pstart, uint32_t* __restrict__ plen, const int NW) {
  constexpr uint32_t B = 1u << (CB - 1);
  static_assert(B <= 64, "batch path supports window sizes up to 7 bits");
  __shared__ uint32_t cnt[B], cur[B];
  const uint32_t inst = blockIdx.x;
  const uint32_t lo = offsets[inst], n = offsets[inst + 1] - lo;
  const size_t entry_base = (size_t)lo * NW;
  uint32_t s[PT][SW + 1];
  uint32_t carry[PT];
  bool live[PT];
#pragma unroll
  for (int q = 0; q < PT; q++) {
    uint32_t i = q * 256 + threadIdx.x;
    live[q] = i < n && !(inf && inf[lo + i]);
    carry[q] = 0;
#pragma unroll
    for (int k = 0; k <= SW; k++) s[q][k] = 0;
    if (i < n) {
      const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)(lo + i) * SW);
#pragma unroll
      for (int k = 0; k < SW / 4; k++) {
        uint4 v = sp[k];
        s[q][4 * k] = v.x; s[q][4 * k + 1] = v.y; s[q][4 * k + 2] = v.z; s[q][4 * k + 3] = v.w;
      }
    }
  }
#pragma unroll 1
  for (int w = 0; w < NW; w++) {
    if (threadIdx.x < B) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mag[PT], neg[PT];
    const int bit = w * CB;
    const int wi = bit >> 5, off = bit & 31;
#pragma unroll
    for (int q = 0; q < PT; q++) {
      uint32_t raw = 0;
      if (wi < SW) {
        // dynamic word index: select from the register array (SW <= 12)
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int k = 0; k <= SW; k++) { if (k == wi) w0 = s[q][k]; if (k == wi + 1) w1 = s[q][k]; }
        uint64_t two = ((uint64_t)w1 << 32) | w0;
        raw = (uint32_t)(two >> off) & ((1u << CB) - 1);
      }
      uint32_t d = raw + carry[q];
      neg[q] = d > B ? 1u : 0u;
      mag[q] = neg[q] ? ((1u << CB) - d) : d;
      carry[q] = neg[q];
      if (!live[q]) mag[q] = 0;
      if (mag[q]) atomicAdd(&cnt[mag[q] - 1], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of B <= 64 counters by the first wave
      uint32_t v = threadIdx.x < B ? cnt[threadIdx.x] : 0, x = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if ((int)threadIdx.x >= o) x += y;
      }
      if (threadIdx.x < B) {
        uint32_t excl = x - v;
        cur[threadIdx.x] = excl;
        size_t bucket = ((size_t)inst * NW + w) * B + threadIdx.x;
        pstart[bucket] = (uint32_t)(entry_base + (size_t)w * n + excl);
        plen[bucket] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PT; q++) {
      if (mag[q]) {
        uint32_t pos = atomicAdd(&cur[mag[q] - 1], 1u);
        sorted[entry_base + (size_t)w * n + pos] = (lo + q * 256 + threadIdx.x) | (neg[q] << 31);
      }
    }
    __syncthreads();
  }
}

// (Round 4, measured and not kept - DESIGN.md section 6: this kernel is NOT latency-bound at config 3's scale.  The G2 leg brings
// 11 windows x 4096 instances = 45 056 running sums of 64 full additions, 2.9 M full additions of Fq2 points in 3.1 ms = 0.93 G/s, 1.6x
// what a mixed addition costs k_accumulate - the 3.1 ms are work.  Six lanes per running sum (LanePoint on the hex backend): 3.46 ms;
// window sums by bit position - chains of 15 additions on 5x the lanes, the Horner pass taking one addition per bit - 3.7 + 1.2 ms
// against 3.1 + 0.6.  Kept from it: the bucket's coordinates are loaded where they are used, curve.h xyzz_add_mem - scratch 1264 -> 260 B.)
template <class G>
__global__ void __launch_bounds__(128) k_batch_reduce(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ plen,
                                                      uint32_t* __restrict__ wsum, uint32_t B, uint32_t nvw) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t vw = blockIdx.x * blockDim.x + threadIdx.x;
  if (vw >= nvw) return;
  Xyzz<F> running = Xyzz<F>::identity(), acc = Xyzz<F>::identity();
  for (int b = (int)B - 1; b >= 0; b--) {
    size_t bucket = (size_t)vw * B + b;
    if (plen[bucket]) xyzz_add_mem(running, partials + bucket * IO::XYZZ_WORDS);   // the bucket's coordinates loaded where they are used (curve.h)
    xyzz_add(acc, running);
  }
  IO::store_xyzz(wsum + (size_t)vw * IO::XYZZ_WORDS, acc);
}

template <class G>
__global__ void __launch_bounds__(128) k_batch_horner(const uint32_t* __restrict__ wsum, uint64_t* __restrict__ out, uint32_t nw,
                                                      uint32_t c, uint32_t m) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= m) return;
  Xyzz<F> acc = Xyzz<F>::identity();
  for (int w = (int)nw - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < c; k++) acc = xyzz_dbl(acc);      // inlined: this chain is pure latency (c * nw dependent doublings)
    Xyzz<F> v = IO::load_xyzz(wsum + ((size_t)inst * nw + w) * IO::XYZZ_WORDS);
    xyzz_add(acc, v);
  }
  uint64_t* o = out + (size_t)inst * 3 * IO::ARK64;
  if (acc.is_identity() || acc.ZZ.is_zero_mod_p()) {
    F::zero().to_ark(o); F::one().to_ark(o + IO::ARK64); F::zero().to_ark(o + 2 * IO::ARK64);
  } else {
    F::mul(acc.X, acc.ZZ).to_ark(o);
    F::mul(acc.Y, acc.ZZZ).to_ark(o + IO::ARK64);
    acc.ZZ.to_ark(o + 2 * IO::ARK64);
  }
}

// The same Horner pass with THREE LANES PER INSTANCE (curve_lanes.h): the chain of c * nw dependent doublings is pure latency
// (136 of them per Batch::verify instance), one lane per instance leaves all but 64 waves of the chip idle, and spreading each
// doubling's independent products over a lane group more than halves its latency.  21 instances per 64-lane block.
template <class G>
__global__ void __launch_bounds__(64) k_batch_horner_lanes(const uint32_t* __restrict__ wsum, uint64_t* __restrict__ out, uint32_t nw,
                                                           uint32_t c, uint32_t m) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  typedef QTriT<FieldBase<F>> QB;
  typedef LanePoint<QB> LP;
  const int g = QB::group();
  const uint32_t inst = blockIdx.x * 21u + (uint32_t)g;
  if (g >= 21 || inst >= m) return;
  typename LP::Pt acc;
  acc.inf = true;
  for (int w = (int)nw - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < c; k++) LP::dbl(acc);
    const Xyzz<F> v = IO::load_xyzz(wsum + ((size_t)inst * nw + w) * IO::XYZZ_WORDS);     // every lane of the group loads the window sum
    const typename LP::P b = {v.X, v.Y, v.ZZ, v.ZZZ};
    LP::add(acc, b, v.is_identity());
  }
  if (QB::lane() != 0) return;
  uint64_t* o = out + (size_t)inst * 3 * IO::ARK64;
  if (acc.inf || acc.p.ZZ.is_zero_mod_p()) {
    F::zero().to_ark(o); F::one().to_ark(o + IO::ARK64); F::zero().to_ark(o + 2 * IO::ARK64);
  } else {
    F::mul(acc.p.X, acc.p.ZZ).to_ark(o);
    F::mul(acc.p.Y, acc.p.ZZZ).to_ark(o + IO::ARK64);
    acc.p.ZZ.to_ark(o + 2 * IO::ARK64);
  }
}

// G2 of BLS12-377: SIX lanes per instance - the two halves of every Fq2 coordinate in adjacent lanes (QHex377, pairing_lanes.h:
// one signed two-product Montgomery pass per lane and product instead of a three-product Karatsuba), so a lane executes half
// the instructions per doubling: the Horner chain of Batch::verify's key sums is latency and nothing else.  10 instances per
// 64-lane block.  LanePoint runs unchanged on the backend (same formulas, same decisions per group).
template <class G>   // G2_377 only (a template so that every translation unit including this header may hold a copy)
__global__ void __launch_bounds__(64) k_batch_horner_hex(const uint32_t* __restrict__ wsum, uint64_t* __restrict__ out, uint32_t nw, uint32_t c, uint32_t m) {
  typedef PointIO<Fq2> IO;
  typedef QHex377 QB;
  typedef LanePoint<QB> LP;
  constexpr int HW = Fq::WORDS;                      // device words per Fq half (a coordinate is c0 then c1)
  const int g = QB::group(), h = QB::hsel();
  const uint32_t inst = blockIdx.x * 10u + (uint32_t)g;
  if (g >= 10 || inst >= m) return;
  LP::Pt acc;
  acc.inf = true;
  for (int w = (int)nw - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < c; k++) LP::dbl(acc);
    const uint32_t* src = wsum + ((size_t)inst * nw + w) * IO::XYZZ_WORDS + h * HW;   // every lane loads its half of the window sum
    const LP::P b = {Fq::load(src), Fq::load(src + 2 * HW), Fq::load(src + 4 * HW), Fq::load(src + 6 * HW)};
    const int z = b.ZZ.limbs_all_zero() ? 1 : 0;                                       // the identity is stored as exact zeros
    const bool b_inf = (z & __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ 1) << 2, z)) != 0;
    LP::add(acc, b, b_inf);
  }
  if (QB::lane() != 0) return;
  uint64_t* o = out + (size_t)inst * 3 * IO::ARK64 + h * Fq::ARK64;
  if (acc.inf || QB::is_zero_u(acc.p.ZZ)) {
    Fq::zero().to_ark(o);
    (h ? Fq::zero() : Fq::one()).to_ark(o + IO::ARK64);
    Fq::zero().to_ark(o + 2 * IO::ARK64);
  } else {
    QB::mul(acc.p.X, acc.p.ZZ).to_ark(o);
    QB::mul(acc.p.Y, acc.p.ZZZ).to_ark(o + IO::ARK64);
    acc.p.ZZ.to_ark(o + 2 * IO::ARK64);
  }
}
template <class G> struct BatchHornerLanes {
  static void launch(const uint32_t* d_wsum, uint64_t* d_out, uint32_t nw, uint32_t c, uint32_t m, hipStream_t stream) {
    hipLaunchKernelGGL((k_batch_horner_lanes<G>), dim3((m + 20) / 21), dim3(64), 0, stream, d_wsum, d_out, nw, c, m);
  }
};
template <> struct BatchHornerLanes<G2_377> {
  static void launch(const uint32_t* d_wsum, uint64_t* d_out, uint32_t nw, uint32_t c, uint32_t m, hipStream_t stream) {
    hipLaunchKernelGGL((k_batch_horner_hex<G2_377>), dim3((m + 9) / 10), dim3(64), 0, stream, d_wsum, d_out, nw, c, m);
  }
};

// ---- GLV expansion of ONE MSM over BLS12-377 for bases in the prime-order subgroup (msm_bls12_377_g1_subgroup / _g2_subgroup).  G1:
// phi(x, y) = (beta x, y) acts on the subgroup as multiplication by -x^2 (wire.h proves it: the G1 subgroup test); G2: psi acts as [x],
// so psi^2 as [x^2].  With k = k0 + k1 x^2
//   [k]P = [k0]P + [k1] I(P),   I(P) = (beta x, -y) on G1, psi^2(P) on G2,     0 <= k0, k1 < 2^127   (gls.h glv_split_x2).
// n terms with 253-bit scalars become 2 n terms with 127-bit scalars: the same number of bucket additions (8 windows of 16 bits over
// 2 n points instead of 16 over n), HALF the windows - half the buckets to reduce, half the host's Horner chain.  Point i and its image
// sit at i and n + i; a base flagged as the identity gets zero scalars.  Replaces k_convert_bases on this path.
template <class G> struct GlvImage;          // [x^2]P of a subgroup point P, four or fewer field products
template <> struct GlvImage<G1_377> {        // (beta x, -y): phi(x, y) = (beta x, y) = -[x^2](x, y)
  HD static Affine<Fq> of(const Affine<Fq>& P) { return {Fq::mul(P.x, Fq::from_limbs(T377::BETA_GLV)), Fq::wred(Fq::norm(Fq::neg<4, 1>(P.y)))}; }
};
template <> struct GlvImage<G2_377> {        // psi^2(x, y) = (PSI_X^2 x, PSI_Y^2 y): psi acts on G2 as [x] (the GLS expansion below uses psi^j)
  HD static Affine<Fq2> of(const Affine<Fq2>& P) {
    const Fq kx = Fq::from_limbs(T377::PSI_X2), ky = Fq::from_limbs(T377::PSI_Y2);
    return {{Fq::mul(P.x.c0, kx), Fq::mul(P.x.c1, kx)}, {Fq::mul(P.y.c0, ky), Fq::mul(P.y.c1, ky)}};
  }
};
template <class G>   // G1_377 and G2_377
__global__ void __launch_bounds__(256) k_glv_expand(const uint64_t* __restrict__ ark, const uint8_t* __restrict__ inf, const uint32_t* __restrict__ scalars,
                                                    uint32_t n, uint32_t* __restrict__ dev_bases, uint32_t* __restrict__ sc2) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* s = ark + (size_t)i * 2 * IO::ARK64;
  const Affine<F> P = {F::from_ark(s), F::from_ark(s + IO::ARK64)};
  IO::store_affine(dev_bases + (size_t)i * IO::AFF_WORDS, P);
  IO::store_affine(dev_bases + ((size_t)n + i) * IO::AFF_WORDS, GlvImage<G>::of(P));
  uint32_t k[8], k0[4], k1[4];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 a = sp[0], b = sp[1];
  k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z;
  k[7] = b.w & ((1u << (G::SCALAR_BITS - 224)) - 1u);      // bits from Fr::MODULUS_BITS up are ignored, as on the plain path (k_digits) and in ark-ec
  glv_split_x2<8>(k, k0, k1);
  const uint32_t keep = (inf && inf[i]) ? 0u : 0xffffffffu;
  reinterpret_cast<uint4*>(sc2)[i] = uint4{k0[0] & keep, k0[1] & keep, k0[2] & keep, k0[3] & keep};
  reinterpret_cast<uint4*>(sc2)[(size_t)n + i] = uint4{k1[0] & keep, k1[1] & keep, k1[2] & keep, k1[3] & keep};
}
template <class G> struct GlvExpand {
  static constexpr bool AVAILABLE = false;
  static constexpr int BITS = 0;
  static void launch(const uint64_t*, const uint8_t*, const uint32_t*, uint32_t, uint32_t*, uint32_t*, hipStream_t) {}
};
template <class G> struct GlvExpandX2 {      // the two BLS12-377 groups: k = k0 + k1 x^2, both halves below 2^127
  static constexpr bool AVAILABLE = true;
  static constexpr int BITS = 127;
  static void launch(const uint64_t* ark, const uint8_t* inf, const uint32_t* sc, uint32_t n, uint32_t* dev_bases, uint32_t* sc2, hipStream_t st) {
    hipLaunchKernelGGL((k_glv_expand<G>), dim3((n + 255) / 256), dim3(256), 0, st, ark, inf, sc, n, dev_bases, sc2);
  }
};
template <> struct GlvExpand<G1_377> : GlvExpandX2<G1_377> {};
template <> struct GlvExpand<G2_377> : GlvExpandX2<G2_377> {};

// ---- GLS expansion of a batch of G2 instances (BLS12-377): psi = twist^-1 o Frobenius o twist acts on the prime-order subgroup of
// E'(Fq2) as multiplication by the curve parameter x (proved in wire.h, where the same fact is the subgroup test), so
//   [k]P = [d0]P + [d1]psi(P) + [d2]psi^2(P) + [d3]psi^3(P),   k = d0 + d1 x + d2 x^2 + d3 x^3,  0 <= d_j < x < 2^64.
// psi^j(x, y) = (PSI_X^j conj^j(x), PSI_Y^j conj^j(y)) with PSI_X = (-5)^((q-1)/6), PSI_Y = (-5)^((q-1)/4) in Fq: four Fq products per
// image.  One workgroup per instance; instance p of n_p points becomes one of nd n_p points, block j holding psi^j of the originals,
// with 64-bit scalars in 16-byte containers; bases are written in device form (this replaces k_convert_bases).
// Division by x (normalised: its top bit is set) is Knuth's algorithm D in base 2^32 with the two-digit divisor (x >> 32, 1).
// One wave per block (blockIdx.y = which 64 points of the instance) and at most 128 registers: the expansion runs beside the other
// group's accumulate kernel (Batch::verify starts both MSMs at once), whose waves leave less than half a SIMD's register file - a
// 256-thread block of 334-register waves waited for four EMPTY SIMDs of one CU and took 4.5 ms for 0.6 ms of work.
template <class G, int NW, int ND>   // G2_377 only (a template so that every translation unit including this header may hold a copy)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_gls_expand(const uint64_t* __restrict__ ark, const uint8_t* __restrict__ inf, const uint32_t* __restrict__ scalars,
                                                    const uint32_t* __restrict__ offsets, uint32_t* __restrict__ dev_bases,
                                                    uint32_t* __restrict__ sc2, uint8_t* __restrict__ inf2) {
  constexpr int nd = ND;
  typedef PointIO<Fq2> IO;
  const uint32_t inst = blockIdx.x;
  const uint32_t lo = offsets[inst], n = offsets[inst + 1] - lo;
  for (uint32_t t = blockIdx.y * 64u + threadIdx.x; t < n; t += gridDim.y * 64u) {
    const uint64_t* s = ark + (size_t)(lo + t) * 2 * IO::ARK64;
    uint32_t d[4][2];
    gls_digits_base_x<NW, ND>(scalars + (size_t)(lo + t) * 8, d);
    const uint8_t fl = inf ? inf[lo + t] : 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j < nd) {
        const size_t e = (size_t)nd * lo + (size_t)j * n + t;
        uint4 w = {d[j][0], d[j][1], 0u, 0u};
        reinterpret_cast<uint4*>(sc2)[e] = w;
        if (inf2) inf2[e] = fl;
      }
    }
    // the images ONE COORDINATE HALF AT A TIME (round 4): psi^j(x0 + x1 u) = (kx_j x0, +-kx_j x1), so a half of P is read, scaled by the
    // nd - 1 constants and stored before the next is touched - 14 live registers of input instead of the point and its three images
    // (the first form kept them all alive: 1232 B/lane of scratch at the 128 registers this kernel has, 1.1 ms for config 3's 10^6 keys)
    uint32_t* out0 = dev_bases + ((size_t)nd * lo + t) * IO::AFF_WORDS;
#pragma unroll
    for (int h = 0; h < 4; h++) {                        // x.c0, x.c1, y.c0, y.c1
      const Fq v = Fq::from_ark(s + (size_t)h * Fq::ARK64);
      v.store(out0 + h * Fq::WORDS);
      const Fq vn = (h & 1) ? Fq::wred(Fq::norm(Fq::neg<4, 1>(v))) : v;     // conjugation: the u half changes sign for odd j
#pragma unroll
      for (int j = 1; j < 4; j++) {
        if (j < nd) {
          const uint32_t* kc = h < 2 ? (j == 1 ? T377::PSI_X1 : j == 2 ? T377::PSI_X2 : T377::PSI_X3) : (j == 1 ? T377::PSI_Y1 : j == 2 ? T377::PSI_Y2 : T377::PSI_Y3);
          Fq::mul((j & 1) ? vn : v, Fq::from_limbs(kc)).store(out0 + (size_t)j * n * IO::AFF_WORDS + h * Fq::WORDS);
        }
      }
      asm volatile("" ::: "memory");
    }
  }
}
template <class G> struct GlsExpand {
  static constexpr bool AVAILABLE = false;
  static void launch(const uint64_t*, const uint8_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, int, int, uint32_t*, uint32_t*, uint8_t*, hipStream_t) {}
};
template <> struct GlsExpand<G2_377> {
  static constexpr bool AVAILABLE = true;
  // bits = length of the longest scalar: the number of significant words and of digits are compile-time constants of the kernel
  static void launch(const uint64_t* ark, const uint8_t* inf, const uint32_t* sc, const uint32_t* off, uint32_t m, uint32_t max_n, int nd, int bits,
                     uint32_t* dev_bases, uint32_t* sc2, uint8_t* inf2, hipStream_t st) {
    const int nw = bits <= 96 ? 3 : (bits + 31) / 32;
    const dim3 grid(m, (max_n + 63) / 64);
#define CELO_GLS_CASE(NW_, ND_) \
    if (nw == NW_ && nd == ND_) { hipLaunchKernelGGL((k_gls_expand<G2_377, NW_, ND_>), grid, dim3(64), 0, st, ark, inf, sc, off, dev_bases, sc2, inf2); return; }
    CELO_GLS_CASE(3, 2) CELO_GLS_CASE(4, 2) CELO_GLS_CASE(4, 3) CELO_GLS_CASE(5, 3) CELO_GLS_CASE(6, 3) CELO_GLS_CASE(6, 4) CELO_GLS_CASE(7, 4)
#undef CELO_GLS_CASE
    hipLaunchKernelGGL((k_gls_expand<G2_377, 8, 4>), grid, dim3(64), 0, st, ark, inf, sc, off, dev_bases, sc2, inf2);
  }
};

// =====================================================================================================================
// FIXED-BASE MSM: the Groth16 prover hands the SAME Parameters to every proof (crates/epoch-snark/src/api/prover.rs:78,112;
// they are created once, crates/epoch-snark/src/api/setup.rs:63-105), so its queries can carry per-key tables T[j][i] = 2^(cf j) P_i.  Then
//   sum_i k_i P_i = sum_i sum_j d_ij T[j][i]      (d_ij the signed cf-bit digits of k_i)
// and ALL n W digit entries fall into ONE set of 2^(cf - 1) buckets: one bucket reduction instead of W, no Horner chain over the windows,
// and cf is free to grow beyond the 16 bits of the variable-base windows (fewer digits per scalar = fewer additions: 19 instead of 24 for
// the 377-bit scalars at cf = 20).  The pipeline below the digits is the variable-base one, unchanged: the table is handed to it as E = n W
// bases, and the 2^(cf - 1) buckets as NV VIRTUAL windows of M buckets each (bucket b = v M + low) - entry p = j n + i carries its digit in
// the virtual window it belongs to and a zero digit in the others.  M = 2^15 - 1 for cf > 16: the pipeline's 16-bit digit is 15 bits of
// bucket + the sign + the value 0xFFFF for "no digit", and (low = 0x7FFF, negative) IS 0xFFFF - with cf = 16 a negative digit never
// reaches that bucket, with virtual windows it does (caught by the scalar r - 1 at cf = 19), so the last bucket of every virtual window
// stays empty and NV = floor((2^(cf-1) - 1) / M) + 1.  What changes is the end:
//   total = sum_v [ S_v + v M T_v ],   S_v = the window's weighted sum (node + sum_l 2^(15 - l) O_l),   T_v = its plain sum (node),
// one Horner pass of ~15 + log2(NV) doublings on the host (run_device_windows, fx branch): the bits of v M from the top, the plain sums
// of the windows that have the bit added at each step, the levels' O_l joining in from bit 14 down.
struct FixedTable {
  uint32_t* table = nullptr;      // E affine points, device form, entry j n + i = 2^(cf j) P_i
  uint8_t* tinf = nullptr;        // E flags: the entry is the identity (a flagged base; a base whose 2^(cf j) multiple is the identity)
  uint32_t n = 0, W = 0, NV = 0, M = 0;      // M: buckets used per virtual window (the divisor of the bucket index)
  int cf = 0, device = 0;
  size_t bytes = 0;
  float build_ms = 0;
  uint32_t E() const { return n * W; }
};
// T[j] from T[j - 1]: cf doublings and one inversion per point (the inversion is ~20 % of the lane's work: no batching needed for a
// table that is built once per key)
template <class G>
__global__ void __launch_bounds__(128) k_fixed_next(const uint32_t* __restrict__ prev, const uint8_t* __restrict__ pinf, uint32_t* __restrict__ next,
                                                    uint8_t* __restrict__ ninf, uint32_t n, int cf) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> r = {F::zero(), F::zero()};
  uint8_t fl = 1;
  if (!pinf[i]) {
    Xyzz<F> a = Xyzz<F>::from_affine(IO::load_affine(prev + (size_t)i * IO::AFF_WORDS));
    for (int k = 0; k < cf; k++) xyzz_dbl_fn(a);
    if (!a.is_identity() && !a.ZZ.is_zero_mod_p()) {
      const F t = F::inv(F::mul(a.ZZ, a.ZZZ));          // x = X / ZZ, y = Y / ZZZ with one inversion
      r = {F::mul(a.X, F::mul(t, a.ZZZ)), F::mul(a.Y, F::mul(t, a.ZZ))};
      fl = 0;
    }
  }
  IO::store_affine(next + (size_t)i * IO::AFF_WORDS, r);
  ninf[i] = fl;
}
template <class G>
__global__ void __launch_bounds__(256) k_fixed_first_flags(const uint8_t* __restrict__ inf, uint8_t* __restrict__ tinf, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tinf[i] = inf ? (inf[i] ? 1 : 0) : 0;
}
// signed cf-bit digits of the first n_sc scalars (a shorter scalar list leaves the remaining bases out: VariableBaseMSM zips), written
// per virtual window: digits[v E + j n + i] = the low 15 bits of (|d| - 1) | sign << 15 if the entry belongs to v, else the zero digit.
template <int SW, int BITS>
__global__ void __launch_bounds__(256) k_fixed_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ tinf, uint16_t* __restrict__ digits,
                                                      uint32_t n, uint32_t n_sc, int cf, uint32_t W, uint32_t NV, uint32_t M) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t E = (size_t)n * W;
  uint32_t s[SW + 1];
  if (i < n_sc) {
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * SW);
#pragma unroll
    for (int k = 0; k < SW / 4; k++) { const uint4 v = sp[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
  } else {
#pragma unroll
    for (int k = 0; k < SW; k++) s[k] = 0;
  }
  s[SW] = 0;
  if constexpr (BITS < 32 * SW) {          // bits from the scalar length up are not part of the scalar (k_digits, ark-ec)
    s[BITS / 32] &= (1u << (BITS % 32)) - 1u;
#pragma unroll
    for (int k = BITS / 32 + 1; k < SW; k++) s[k] = 0;
  }
  const uint32_t half = 1u << (cf - 1);
  uint32_t carry = 0;
  for (uint32_t j = 0; j < W; j++) {
    const uint32_t bit = j * (uint32_t)cf, wi = bit >> 5, off = bit & 31;
    uint32_t raw = 0;
    if (wi < (uint32_t)SW) {
      uint32_t w0 = 0, w1 = 0;
#pragma unroll
      for (int k = 0; k <= SW; k++) { if ((uint32_t)k == wi) w0 = s[k]; if ((uint32_t)k == wi + 1) w1 = s[k]; }
      raw = (uint32_t)((((uint64_t)w1 << 32) | w0) >> off) & ((1u << cf) - 1u);
    }
    const uint32_t d = raw + carry;
    const uint32_t neg = d > half ? 1u : 0u;
    const uint32_t mag = neg ? (1u << cf) - d : d;
    carry = neg;
    const size_t p = (size_t)j * n + i;
    const bool live = mag != 0 && !tinf[p];
    const uint32_t full = mag - 1u, v = full / M;
    const uint16_t dg = (uint16_t)((full - v * M) | (neg << 15));
    for (uint32_t vv = 0; vv < NV; vv++) digits[(size_t)vv * E + p] = (live && vv == v) ? dg : (uint16_t)0xFFFF;
  }
}

// ---- the same digits COMPACTED BY VIRTUAL WINDOW (round 4).  k_fixed_digits hands the pipeline NV windows of E slots each, all but one of
// an entry's slots holding "no digit": at cf = 20 the two partition passes and the digit kernel move 17 x 4 10^7 x 2 bytes three times and
// the sort costs 1.7 ms where the variable-base sort of as many real entries costs 0.45.  Here every entry gets ONE record (window id,
// digit), the records are placed window by window - window v's entries in slots [0, count_v) of a row of Ep >= max_v count_v slots, the
// rest padded with "no digit" - and the partition pass carries the table index of a slot along (k_part_scatter's remap): the pipeline
// then sees NV windows of Ep ~ E / (NV - 1) slots.  The order of a window's slots is whatever the atomics give; sums do not care.
template <int SW, int BITS>
__global__ void __launch_bounds__(256) k_fixed_digits_c(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ tinf, uint8_t* __restrict__ v8,
                                                        uint16_t* __restrict__ dg16, uint32_t* __restrict__ counts, uint32_t n, uint32_t n_sc, int cf, uint32_t W,
                                                        uint32_t NV, uint32_t M) {
  __shared__ uint32_t lc[128];
  if (threadIdx.x < 128) lc[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint32_t s[SW + 1];
    if (i < n_sc) {
      const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * SW);
#pragma unroll
      for (int k = 0; k < SW / 4; k++) { const uint4 v = sp[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
    } else {
#pragma unroll
      for (int k = 0; k < SW; k++) s[k] = 0;
    }
    s[SW] = 0;
    if constexpr (BITS < 32 * SW) {
      s[BITS / 32] &= (1u << (BITS % 32)) - 1u;
#pragma unroll
      for (int k = BITS / 32 + 1; k < SW; k++) s[k] = 0;
    }
    const uint32_t half = 1u << (cf - 1);
    uint32_t carry = 0;
    for (uint32_t j = 0; j < W; j++) {
      const uint32_t bit = j * (uint32_t)cf, wi = bit >> 5, off = bit & 31;
      uint32_t raw = 0;
      if (wi < (uint32_t)SW) {
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int k = 0; k <= SW; k++) { if ((uint32_t)k == wi) w0 = s[k]; if ((uint32_t)k == wi + 1) w1 = s[k]; }
        raw = (uint32_t)((((uint64_t)w1 << 32) | w0) >> off) & ((1u << cf) - 1u);
      }
      const uint32_t d = raw + carry;
      const uint32_t neg = d > half ? 1u : 0u;
      const uint32_t mag = neg ? (1u << cf) - d : d;
      carry = neg;
      const size_t p = (size_t)j * n + i;
      const bool live = mag != 0 && !tinf[p];
      const uint32_t full = mag - 1u, v = full / M;
      v8[p] = live ? (uint8_t)v : (uint8_t)0xFF;
      dg16[p] = (uint16_t)((full - v * M) | (neg << 15));
      if (live) atomicAdd(&lc[v], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < NV && lc[threadIdx.x]) atomicAdd(counts + threadIdx.x, lc[threadIdx.x]);
}
// record p -> slot of its window's row: ranks within a batch by LDS atomics, one global reservation per (batch, window)
template <class G>   // (a template only so that every translation unit including this header may hold a copy)
__global__ void __launch_bounds__(1024) k_fixed_place(const uint8_t* __restrict__ v8, const uint16_t* __restrict__ dg16, uint32_t* __restrict__ cursor,
                                                      uint16_t* __restrict__ digits, uint32_t* __restrict__ remap, uint32_t E, uint32_t Ep, uint32_t NV) {
  __shared__ uint32_t lc[128], lb[128];
  for (uint32_t p0 = blockIdx.x * 1024u; p0 < E; p0 += gridDim.x * 1024u) {
    if (threadIdx.x < 128) lc[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t p = p0 + threadIdx.x;
    const uint32_t v = p < E ? v8[p] : 0xFFu;
    uint32_t r = 0;
    if (v != 0xFFu) r = atomicAdd(&lc[v], 1u);
    __syncthreads();
    if (threadIdx.x < NV && lc[threadIdx.x]) lb[threadIdx.x] = atomicAdd(cursor + threadIdx.x, lc[threadIdx.x]);
    __syncthreads();
    if (v != 0xFFu) {
      const size_t at = (size_t)v * Ep + lb[v] + r;
      digits[at] = dg16[p];
      remap[at] = p;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- host driver
// the IFMA Horner epilogue (host_ifma.cpp, host_cpu.cpp) exists for the two prime fields
extern "C" int celo_ifma_available();
extern "C" int celo_ifma_horner_377(const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out, int* inf);
extern "C" int celo_ifma_horner_761(const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out, int* inf);
typedef int (*ifma_horner_fn)(const uint64_t*, size_t, const int32_t*, int, uint64_t*, int*);
template <class F> struct IfmaHorner { static constexpr ifma_horner_fn fn = nullptr; };
template <> struct IfmaHorner<Fp<P377>> { static constexpr ifma_horner_fn fn = &celo_ifma_horner_377; };
template <> struct IfmaHorner<Fp<P761>> { static constexpr ifma_horner_fn fn = &celo_ifma_horner_761; };

struct MsmTimings {  // milliseconds, HIP events on the MSM's stream (last call)
  float convert = 0, sort = 0, accumulate = 0, reduce = 0, total = 0;
};

}  // namespace celo
#include "msm_ba.h"
namespace celo {
template <> struct BaCfg<G_761> { static constexpr bool enabled = true; };

// A/B switches and tuning hooks of the pipeline, read from the environment ONCE per process (not per engine, not per call)
struct MsmTuning {
  bool narrow_windows, use_glv, use_gls, gls_force, lane_bitsum, host_threads, seg_occupancy, side_convert, side_convert_all, fx_compact;
  uint32_t seg_min, seg_min_shard, bitsum_lanes_max_shard;
  int seg_halves;          // piece length in half mean-bucket lengths (4 = twice the mean); 0 = not set: the path's own default
  uint32_t bitsum_lanes_max;
  int ba_levels, ba_occ; uint32_t ba_rounds;   // CELO_BA_LEVELS (1..4, default 3), CELO_BA_OCC (waves per SIMD of k_ba_levels: 1 or 2), CELO_BA_ROUNDS (grid = rounds x lanes in flight)
  int batched_affine;      // CELO_BA: 0 (default) = the XYZZ chain everywhere, 1 = batched-affine pre-levels (msm_ba.h) for the groups that enable them (BW6-761)
  uint32_t host_chunks;    // host-pointer entry: index chunks of the pipelined transfer (CELO_HOST_CHUNKS; 0 or 1 = the plain form - celo_amd_msm_set_host_chunks(1), the test hook, is what runs ONE chunk through the pipelined code)
  uint32_t host_head_split, host_tail_split; // ... how often the first / the last of them is cut in halves (CELO_HOST_HEAD_SPLIT, CELO_HOST_TAIL_SPLIT)
  static const MsmTuning& get() {
    static const MsmTuning t = [] {
      MsmTuning v;
      v.narrow_windows = getenv("CELO_NO_NARROW") == nullptr;
      v.use_glv = getenv("CELO_NO_GLV") == nullptr;
      v.use_gls = getenv("CELO_NO_GLS") == nullptr;
      v.gls_force = false;       // the library never applies psi to the plain entry points' arbitrary curve points (the round-3 measurement hook is gone)
      v.lane_bitsum = getenv("CELO_NO_LANE_BITSUM") == nullptr;
      v.host_chunks = getenv("CELO_HOST_CHUNKS") ? (uint32_t)atoi(getenv("CELO_HOST_CHUNKS")) : 0xFFFFFFFFu;   // not set: the group's own default
      v.host_head_split = getenv("CELO_HOST_HEAD_SPLIT") ? (uint32_t)atoi(getenv("CELO_HOST_HEAD_SPLIT")) : 0xFFFFFFFFu;
      v.host_tail_split = getenv("CELO_HOST_TAIL_SPLIT") ? (uint32_t)atoi(getenv("CELO_HOST_TAIL_SPLIT")) : 0xFFFFFFFFu;
      v.fx_compact = getenv("CELO_FX_NO_COMPACT") == nullptr;        // A/B switch: fixed base, digits compacted by virtual window
      v.host_threads = getenv("CELO_NO_HOST_THREADS") == nullptr;
      v.seg_halves = getenv("CELO_SEG_HALVES") ? atoi(getenv("CELO_SEG_HALVES")) : 0;
      v.seg_min = getenv("CELO_SEG_MIN") ? (uint32_t)atoi(getenv("CELO_SEG_MIN")) : 32u;                     // shortest piece of a whole MSM
      v.seg_min_shard = getenv("CELO_SEG_MIN_SHARD") ? (uint32_t)atoi(getenv("CELO_SEG_MIN_SHARD")) : 16u;   // ... of a window shard (8 / 12 / 16 measure alike)
      v.seg_occupancy = getenv("CELO_NO_SEG_OCC") == nullptr;                                                  // A/B switch of the window shards' piece length
      v.side_convert_all = getenv("CELO_SIDE_CONVERT") && atoi(getenv("CELO_SIDE_CONVERT")) == 2;     // 2: whole MSMs too (A/B hook, round 6)
      v.side_convert = getenv("CELO_SIDE_CONVERT") != nullptr;                                                // window shards: base conversion beside the sort
      v.bitsum_lanes_max_shard = getenv("CELO_LANE_BITSUM_MAX_SHARD") ? (uint32_t)atoi(getenv("CELO_LANE_BITSUM_MAX_SHARD")) : 21 * 1024;
      v.batched_affine = getenv("CELO_BA") ? atoi(getenv("CELO_BA")) : 0;      // measured level with the XYZZ chain (profiles/r6_ba_ab.txt): off
      v.ba_levels = getenv("CELO_BA_LEVELS") ? atoi(getenv("CELO_BA_LEVELS")) : 3;
      if (v.ba_levels < 1) v.ba_levels = 1;
      if (v.ba_levels > BA_K_MAX) v.ba_levels = BA_K_MAX;
      v.ba_occ = getenv("CELO_BA_OCC") && atoi(getenv("CELO_BA_OCC")) == 1 ? 1 : 2;
      v.ba_rounds = getenv("CELO_BA_ROUNDS") ? (uint32_t)atoi(getenv("CELO_BA_ROUNDS")) : 2u;
      if (v.ba_rounds < 1) v.ba_rounds = 1;
      v.bitsum_lanes_max = getenv("CELO_LANE_BITSUM_MAX") ? (uint32_t)atoi(getenv("CELO_LANE_BITSUM_MAX")) : 21 * 1024;   // outputs of a launch: one wave of 21 additions per SIMD
      return v;
    }();
    return t;
  }
};

template <class G> class MsmEngine {
 public:
  typedef typename G::F F;
  typedef PointIO<F> IO;
  static constexpr int SW = G::SCALAR_WORDS;

  ~MsmEngine() { release(); }
  void release() {
    if (arena) { (void)hipFree(arena); arena = nullptr; arena_bytes = 0; }
    if (fxs) { (void)hipFree(fxs); fxs = nullptr; fxs_bytes = 0; }
    for (void* p : {(void*)d_in_bases, (void*)d_in_scalars, (void*)d_in_inf})
      if (p) (void)hipFree(p);
    d_in_bases = nullptr; d_in_scalars = nullptr; d_in_inf = nullptr; cap_in = 0;
    if (h_out) { (void)hipHostFree(h_out); h_out = nullptr; }
    if (d_side_out) { (void)hipFree(d_side_out); d_side_out = nullptr; side_out_bytes = 0; }
    if (d_fx_scalars) { (void)hipFree(d_fx_scalars); d_fx_scalars = nullptr; cap_fx = 0; }
    for (int i = 0; i < 6; i++) if (ev[i]) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
    for (int i = 0; i < 2; i++) if (ev_side[i]) { (void)hipEventDestroy(ev_side[i]); ev_side[i] = nullptr; }
    for (hipEvent_t e : ev_copy) if (e) (void)hipEventDestroy(e);
    ev_copy.clear();
  }
  // Measured on the MI355X (sweep over c at n = 2^8 .. 2^20, uniform scalars, all groups): with the log-depth bucket reduction
  // the buckets are cheap and the accumulate lanes are not - few long bucket runs are pure latency - so small and mid-size
  // inputs want MORE buckets than points, and a window size that DIVIDES the scalar length wins by up to 2x because no ragged
  // top window (few, heavy buckets) is left: 253 = 11 * 23 and 377 = 13 * 29.
  //   253-bit scalars (BLS12-377): c = 11 below 2^15 points (0.34 ms at n = 256, was 0.9 with c = 4), 15 below 2^19, then 16
  //   377-bit scalars (BW6-761):   c = 13 below 2^19 points (6.9 ms at 2^18, was 7.1 with c = 14 and 8.6 with 15), then 16
  static int window_bits(size_t n) {
    if (G::SCALAR_BITS > 256) {
      if (n < 512) return 9;
      return n < (size_t(1) << 19) ? 13 : 16;
    }
    if (n < (size_t(1) << 15)) return 11;
    return n < (size_t(1) << 19) ? 15 : 16;
  }
  int force_c = 0;  // test hook / tuning: 0 = auto
  hipStream_t own_stream() { return stream_.get(); }   // this engine's non-blocking stream (host-pointer entry points)
  // big path: mixed window widths (k_digits) for the 16-bit configuration only - the large inputs, where the work is throughput
  // and a ragged top window costs folds and balance (2^20 terms: G1 3.32 -> 3.31, G2 10.85 -> 10.75, BW6-761 19.5 -> 19.0 ms).
  // Small inputs are bound by their longest bucket run, and narrower windows mean longer runs: BW6-761 at 2^14 with 18 x 13 + 12 x 12
  // bits instead of 29 x 13 (+ a carry window) 1.39 -> 1.88 ms, at 2^17 3.58 -> 3.79 ms.  CELO_NO_NARROW=1 is the A/B switch.
  bool narrow_windows = MsmTuning::get().narrow_windows;
  bool narrow_top(int c) const { return narrow_windows && c == 16; }
  // big path, G1 of BLS12-377: the caller vouches for bases in the prime-order subgroup (Signature values, proving-key points): GLV split
  bool big_subgroup_points = false;
  // host-pointer entry (run_host), set by the Groth16 prover's entry points only: a base row x = 0, y = 1 is the identity (k_flag_ark_zero)
  bool ark_zero_identity = false;
  bool use_glv = MsmTuning::get().use_glv;     // A/B switch (CELO_NO_GLV)
  bool last_glv = false;
  // window size for the 2 n points x 127-bit halves of the split (n = the expanded count)
  // (measured, round 3: 127 = 8 x 16 - 1, so c = 16 leaves no ragged top window and wins at every size from 2^14 terms up)
  static int window_bits_glv(size_t) { return 16; }
  // batched path, G2 of BLS12-377: the caller vouches that every base lies in the prime-order subgroup (Batch::verify's public keys:
  // PublicKey values only come from checked deserialisation, secret keys and sums of such), which is what makes psi(P) = [x]P
  bool gls_subgroup_points = false;
  bool use_gls = MsmTuning::get().use_gls;     // A/B switch (CELO_NO_GLS)
  bool gls_force = MsmTuning::get().gls_force;
  int last_gls_digits = 1;
  static constexpr int HOST_HORNER_THREADS = 4;
  bool host_threads = MsmTuning::get().host_threads;   // A/B switch (CELO_NO_HOST_THREADS) of the threaded host epilogue (Fq2 and BW6-761 groups)
  bool lane_horner = true;  // batched path: three lanes per instance in the Horner pass (tuning hook)
  bool lane_bitsum = MsmTuning::get().lane_bitsum;   // big path (CELO_NO_LANE_BITSUM): three lanes per addition in the late levels of the bucket reduction (A/B hook)
  uint32_t BITSUM_LANES_MAX = MsmTuning::get().bitsum_lanes_max;

  // bases/scalars/inf are DEVICE pointers (ark layout); result: Jacobian in ark Montgomery form (3*ARK64 u64) on host.
  // The shape of a call: whether the GLV split is taken, the expanded term count, the scalar length, the window size and count.
  struct Plan { bool glv; uint32_t n; int sbits, c, nw, kn; };
  Plan plan(size_t n_) const {
    Plan p;
    // GLV split (big_subgroup_points: the caller vouches for bases in the prime-order subgroup): 2 n_ terms of sbits-bit scalars
    // (from 2^14 terms: below, the plain path's c = 11 is as fast - measured 0.53 / 0.59 ms at 2^12 / 2^13 either way)
    p.glv = GlvExpand<G>::AVAILABLE && big_subgroup_points && use_glv && n_ >= (size_t(1) << 14);
    p.n = p.glv ? 2u * (uint32_t)n_ : (uint32_t)n_;
    p.sbits = p.glv ? GlvExpand<G>::BITS : G::SCALAR_BITS;
    p.c = force_c ? force_c : (p.glv ? window_bits_glv(p.n) : window_bits(p.n));
    p.nw = (p.sbits + p.c) / p.c;
    p.kn = narrow_top(p.c) ? p.nw * p.c - (p.sbits + 1) : 0;    // the top kn windows are c - 1 bits wide (k_digits)
    return p;
  }
  // first scalar bit of window w (windows of mixed width: the top kn of the nw are c - 1 bits wide)
  static int window_bit(const Plan& p, int w) {
    const int wide = p.nw - p.kn;
    return w < wide ? w * p.c : wide * p.c + (w - wide) * (p.c - 1);
  }
  int run_device(const uint64_t* d_ark_bases, const uint8_t* d_inf, const uint32_t* d_scalars, size_t n_, uint64_t* out_jac,
                 hipStream_t stream) {
    return run_device_windows(d_ark_bases, d_inf, d_scalars, n_, 0, 0, out_jac, nullptr, stream);
  }
  // The same pipeline over the windows [win_lo, win_lo + win_cnt) of the call's plan only (win_cnt = 0: all of them): the WINDOW
  // partition of one MSM over several devices (msm_unit.h msm_multi_windows_impl; SURVEY.md section 8e "alternative partitioning").
  // The result is then the partial sum  sum_{w in range} 2^(bit(w) - bit(win_lo)) S_w  - the caller weighs it by 2^bit(win_lo).
  // out_xyzz (optional, 4 * ARK64 u64: X, Y, ZZ, ZZZ in arkworks limbs, ZZ = 0 for the identity) hands the partial over in the host
  // epilogue's own coordinates, so that the join needs no conversion.
  // fx != nullptr: the FIXED-BASE form (FixedTable above): d_ark_bases / d_inf are unused, n_ = the number of scalars (<= fx->n), the
  // pipeline runs over the table's E entries in NV virtual windows of 2^15 buckets.
  // hin != nullptr: the HOST-POINTER pipeline (round 5; VERDICT r4 item 1 - the call a drop-in caller makes: signature.rs:82-85,
  // public.rs:58-61 hand host slices to multi_scalar_mul).  d_ark_bases / d_inf / d_scalars are then the engine's staging buffers, still
  // EMPTY: scalars (and flags) and bases cross in hin->chunks index chunks on a copy stream; a chunk's digits and sort run over its
  // (chunk, window) virtual windows on a sort stream beside the accumulation of the chunk before, and every chunk is converted and
  // accumulated (k_accumulate_chunk) as soon as it has landed - the PCIe time hides under the accumulation instead of preceding it.
  struct HostIn { const uint64_t* bases; const uint8_t* inf; const uint64_t* scalars; uint32_t chunks, head_split, tail_split; bool ark_zero; };
  static constexpr uint32_t HOST_HEAD_SPLIT_DEFAULT = 1, HOST_TAIL_SPLIT_DEFAULT = 0;      // (host_chunk_plan: runtime.h)
  int run_device_windows(const uint64_t* d_ark_bases, const uint8_t* d_inf, const uint32_t* d_scalars, size_t n_, int win_lo, int win_cnt,
                         uint64_t* out_jac, uint64_t* out_xyzz, hipStream_t stream, const FixedTable* fx = nullptr, const HostIn* hin = nullptr) {
    if (n_ == 0) {
      if (out_jac) write_identity(out_jac);
      if (out_xyzz) memset(out_xyzz, 0, 4 * IO::ARK64 * 8);
      return 0;
    }
    if (n_ >= (size_t(1) << 30)) return 2;
    if (fx && (win_cnt || n_ > fx->n || fx->cf < 16 || fx->cf > 22)) return 2;
    Plan pl = plan(n_);
    // fixed base with several virtual windows: the digits are taken first, compacted by window (k_fixed_digits_c), and the pipeline is
    // sized by the fullest window's row, Ep, instead of by all E entries per window
    uint32_t fx_Ep = 0;
    uint8_t* d_fx_v8 = nullptr; uint16_t* d_fx_dg = nullptr; uint32_t* d_fx_cnt = nullptr;
    if (fx && fx->NV > 2 && fx->NV <= 128 && !win_cnt && n_ <= fx->n && fx->cf >= 16 && fx->cf <= 22 && MsmTuning::get().fx_compact) {
      const size_t E = fx->E();
      const size_t o_dg = (E + 255) & ~size_t(255), o_cnt = o_dg + ((E * 2 + 255) & ~size_t(255)), need = o_cnt + 256 * 4;
      if (need > fxs_bytes) {
        if (fxs) (void)hipFree(fxs);
        fxs = nullptr; fxs_bytes = 0;
        HIP_OK(hipMalloc((void**)&fxs, need + need / 8));
        fxs_bytes = need + need / 8;
      }
      d_fx_v8 = fxs; d_fx_dg = (uint16_t*)(fxs + o_dg); d_fx_cnt = (uint32_t*)(fxs + o_cnt);
      HIP_OK(hipMemsetAsync(d_fx_cnt, 0, 256 * 4, stream));
      hipLaunchKernelGGL((k_fixed_digits_c<SW, G::SCALAR_BITS>), dim3((fx->n + 255) / 256), dim3(256), 0, stream, d_scalars, fx->tinf, d_fx_v8, d_fx_dg, d_fx_cnt, fx->n,
                         (uint32_t)n_, fx->cf, fx->W, fx->NV, fx->M);
      uint32_t h_cnt[128];
      HIP_OK(hipMemcpyAsync(h_cnt, d_fx_cnt, fx->NV * 4, hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      uint32_t mx = 0;
      for (uint32_t v = 0; v < fx->NV; v++) mx = h_cnt[v] > mx ? h_cnt[v] : mx;
      const uint64_t ep = ((uint64_t)mx + 4095) & ~uint64_t(4095);
      if (ep >= 4096 && ep * 2 <= E) fx_Ep = (uint32_t)ep;       // (a window that holds most entries - tiny scalars - gains nothing: the uncompacted form)
    }
    if (fx) { pl.glv = false; pl.n = fx_Ep ? fx_Ep : fx->E(); pl.sbits = G::SCALAR_BITS; pl.c = 16; pl.nw = (int)fx->NV; pl.kn = 0; }
    if (hin && (fx || pl.glv || win_cnt || hin->chunks < 1 || hin->chunks > 64 || !side_stream_.get() || !sort_stream_.get())) return 2;
    const bool glv = pl.glv;
    const uint32_t n = pl.n;
    const int sbits = pl.sbits, c = pl.c, nw_all = pl.nw;
    if (win_cnt < 0 || win_lo < 0 || (win_cnt && win_lo + win_cnt > nw_all)) return 2;
    const int w0 = win_cnt ? win_lo : 0;
    const int nw = win_cnt ? win_cnt : nw_all;      // windows of THIS call: everything below the digits is sized by it
    if ((uint64_t)n * (uint64_t)nw_all >= (uint64_t(1) << 32)) return 2;  // run offsets are 32-bit (n*windows < 2^32: n <= 2^27 at c = 16)
    const uint32_t B = 1u << (c - 1);
    const uint32_t total = (uint32_t)nw * B;
    // the sort's view: ns entries in each of nws windows - the call's own, or (host-pointer pipeline) the K index chunks of cm points
    // times the windows, chunk-major: virtual window k nw + w
    uint32_t K = 0, cm = n;
    uint32_t clen[HOST_CHUNKS_MAX];
    if (hin) K = host_chunk_plan(n, hin->chunks, hin->head_split, hin->tail_split, cm, clen);
    const uint32_t ns = hin ? cm : n, nws = hin ? K * (uint32_t)nw : (uint32_t)nw, vw = hin ? (uint32_t)nw : 0u;
    const uint32_t npad = hin ? K * cm : n;
    if ((uint64_t)npad * (uint64_t)nw_all >= (uint64_t(1) << 32)) return 2;
    const uint32_t total_s = nws * B;
    // piece length: twice the average bucket, within [32, SIZE_BINS-1]
    // (the split's buckets are twice as long - 64 points at 2^20 - and fewer: pieces of 1.5 mean buckets balance its last round better:
    // accumulate 2.64 -> 2.46 ms at 2^20; the plain path is flat between 1.5 and 3)
    const int seg_h = MsmTuning::get().seg_halves ? MsmTuning::get().seg_halves : (glv ? 3 : 4);
    uint32_t SEG = (uint32_t)seg_h * ((fx && !fx_Ep ? n / (uint32_t)nw_all : ns) / B + 1) / 2;      // (fixed base: a virtual window holds E / NV of the entries; compacted: its row)
    uint32_t seg_min = MsmTuning::get().seg_min;
    if (win_cnt && MsmTuning::get().seg_occupancy) {
      // a call that owns FEW windows (a window shard) has fewer additions than the chip has lanes x the usual piece length: a lane is
      // one addition per ~17 us whatever its neighbours do, so pieces of twice the mean bucket leave most SIMDs with nothing after the
      // first round.  Pieces as long as the additions per lane in flight (ACC_LANES) fill one round; the buckets cut in two or three
      // are folded by k_combine_mid_lanes.  Measured at 2^20 terms, 2 of 16 windows: accumulate 0.55 -> see DESIGN.md section 9.
      const uint64_t adds = (uint64_t)n * (uint64_t)nw;
      const uint32_t lanes = (sizeof(F) <= 14 * sizeof(uint32_t)) ? 131072u : 65536u;      // 2 waves (14-limb field) or 1 per SIMD x 64 lanes x 1024 SIMDs
      const uint32_t occ = (uint32_t)((adds + lanes - 1) / lanes);
      if (occ < SEG) { SEG = occ; seg_min = MsmTuning::get().seg_min_shard; }
    }
    if (fx && !MsmTuning::get().seg_halves) {
      // fixed base: few virtual windows hold all n W entries (one at cf = 16: mean bucket 1536) - pieces of twice the mean bucket would be
      // fewer than the chip has lanes; eight rounds of the lanes in flight bound the piece length instead (cf = 16: 72.9 -> see DESIGN.md)
      const uint32_t lanes = (sizeof(F) <= 14 * sizeof(uint32_t)) ? 131072u : 65536u;
      const uint32_t occ = fx->E() / (lanes * 8u) + 1u;
      if (occ < SEG) SEG = occ;
    }
    if (SEG < seg_min) SEG = seg_min;
    if (SEG > SIZE_BINS - 1) SEG = SIZE_BINS - 1;
    const uint32_t PW = B + ns / SEG + 1;       // static piece region per window
    const uint32_t slots = nws * PW;
    const int LB = c - 1;                                          // bucket-index bits (c >= 4)
    const uint32_t res_pts = (uint32_t)(LB + 1) * (uint32_t)nw;    // results: [0] = node(0,0), [l] = O_l, nw points each
    const uint32_t half_pts = (uint32_t)nw * (B / 2 + B / 4);      // most outputs of one launch (the first)

    // ---- workspace arena
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    const size_t o_bases = take(fx ? 0 : (size_t)npad * IO::AFF_WORDS * 4);       // (npad > n: the host-pointer pipeline's virtual indices)
    const size_t o_sc2 = take(glv ? (size_t)n * 16 : 0);
    const size_t o_digits = take((size_t)npad * nw_all * 2);
    const size_t o_remap = take(fx_Ep ? (size_t)n * nw_all * 4 : 0);
    const size_t o_sorted = take((size_t)ns * nws * 4);
    // two-level sort: NBIN bins per window by the low HIB bucket bits, KB2 blocks per window in the partition pass
    const uint32_t HIB = LB < 8 ? 0u : (uint32_t)LB - 8u, NBIN = 1u << HIB;     // bins by the low HIB bucket bits, <= 8 key bits above
    uint32_t KB2 = ns / (64 * NBIN > 4096 ? 64 * NBIN : 4096);
    if (KB2 < 1) KB2 = 1;
    if (KB2 > 64) KB2 = 64;
    const uint32_t chunk2 = (ns + KB2 - 1) / KB2;
    const size_t o_blockcnt = take((size_t)nws * NBIN * KB2 * 4);
    const size_t o_binstart = take((size_t)nws * (NBIN + 1) * 4);
    const size_t o_tileprefix = take((size_t)nws * (NBIN + 1) * 4);
    const size_t o_recidx = take((size_t)ns * nws * 4);
    const size_t o_reckey = take((size_t)ns * nws);
    // zeroed per call, adjacent so that ONE fill covers them: bucket counts, the tiles' run cursors (`starts`), the folded-bucket
    // flags, the piece lengths (unused slots stay 0) and the size bins with their counters
    const size_t o_counts = take((size_t)total_s * 4);
    const size_t o_starts = take((size_t)total_s * 4);
    const size_t o_piecesof = take((size_t)total_s * 4);
    const size_t o_plen = take((size_t)slots * 4);
    constexpr size_t BINS_STRIDE = SIZE_BINS + 64;            // words: the size bins + nwork, nbig, nmid; one set per chunk (host-pointer pipeline)
    const size_t o_bins = take(BINS_STRIDE * 4 * (hin ? K : 1u));
    const size_t o_zero_end = off;
    const size_t o_pfirst = take((size_t)total_s * 4);
    const size_t o_big = take((size_t)total_s * 4);
    const size_t o_mid = take((size_t)total_s * 4);
    const size_t o_pbucket = take(hin ? (size_t)slots * 4 : 0);
    const size_t o_carrier = take(hin ? (size_t)total * IO::XYZZ_WORDS * 4 : 0);
    const size_t o_pstart = take((size_t)slots * 4);
    const size_t o_order = take((size_t)slots * 4);
    const size_t o_partials = take((size_t)slots * IO::XYZZ_WORDS * 4);
    const size_t o_work = take(((size_t)res_pts + 2 * (size_t)half_pts + 64) * IO::XYZZ_WORDS * 4);
    // batched-affine pre-levels (msm_ba.h): the resident variable-base path of the groups that enable them, from mean runs of 8 points up
    const int ba_ovr = batched_affine_override().load();
    const bool use_ba = BaCfg<G>::enabled && !hin && !fx && (ba_ovr >= 0 ? ba_ovr != 0 : MsmTuning::get().batched_affine != 0) && ns / B >= 8;
    const int ba_occ = MsmTuning::get().ba_occ;
    uint32_t ba_lanes = use_ba ? BA_LANES_OCC1 * (uint32_t)ba_occ * MsmTuning::get().ba_rounds : 0;       // the grid: whole rounds of the lanes in flight
    if (use_ba && ba_lanes > (slots + 255) / 256 * 256) ba_lanes = (slots + 255) / 256 * 256;
    const uint32_t ba_pref_slots = use_ba ? ((slots + ba_lanes - 1) / ba_lanes) * (SEG / 2) : 0;            // pairs of a lane's pieces, at most
    const size_t o_ba_pts = take(use_ba ? (size_t)ns * nws * IO::AFF_WORDS * 4 : 0);
    const size_t o_ba_pref = take(use_ba ? (size_t)ba_pref_slots * ba_lanes * F::WORDS * 4 : 0);
    if (ensure(off)) return 1;
    if (res_pts > H_OUT_POINTS) return 2;
    char* A = arena;
    uint32_t* d_bases = fx ? fx->table : (uint32_t*)(A + o_bases);
    uint16_t* d_digits_all = (uint16_t*)(A + o_digits);
    uint16_t* d_digits = d_digits_all + (size_t)w0 * n;          // this call's windows
    uint32_t* d_sorted = (uint32_t*)(A + o_sorted);
    uint32_t* d_blockcnt = (uint32_t*)(A + o_blockcnt);
    uint32_t* d_binstart = (uint32_t*)(A + o_binstart);
    uint32_t* d_recidx = (uint32_t*)(A + o_recidx);
    uint8_t* d_reckey = (uint8_t*)(A + o_reckey);
    uint32_t* d_tileprefix = (uint32_t*)(A + o_tileprefix);
    uint32_t* d_counts = (uint32_t*)(A + o_counts);
    uint32_t* d_starts = (uint32_t*)(A + o_starts);
    uint32_t* d_pfirst = (uint32_t*)(A + o_pfirst);
    uint32_t* d_piecesof = (uint32_t*)(A + o_piecesof);
    uint32_t* d_big = (uint32_t*)(A + o_big);
    uint32_t* d_mid = (uint32_t*)(A + o_mid);
    uint32_t* d_pstart = (uint32_t*)(A + o_pstart);
    uint32_t* d_plen = (uint32_t*)(A + o_plen);
    uint32_t* d_order = (uint32_t*)(A + o_order);
    uint32_t* d_bins = (uint32_t*)(A + o_bins);
    uint32_t* d_nwork = d_bins + SIZE_BINS;
    uint32_t* d_nbig = d_bins + SIZE_BINS + 1;
    uint32_t* d_nmid = d_bins + SIZE_BINS + 2;
    uint32_t* d_partials = (uint32_t*)(A + o_partials);
    uint32_t* d_work = (uint32_t*)(A + o_work);

    uint32_t* d_pbucket = hin ? (uint32_t*)(A + o_pbucket) : nullptr;
    uint32_t* d_carrier = hin ? (uint32_t*)(A + o_carrier) : nullptr;
    HIP_OK(hipEventRecord(ev[0], stream));
    // mean region ns / NBIN: the smallest workgroup whose tile capacity (TILE_EPT entries per lane) holds it with 20 % to spare
    const uint32_t region = ns / NBIN;
    const uint32_t ts_threads = region <= 2048 ? 256u : region <= 4096 ? 512u : 1024u;
    const uint32_t TILE = TILE_EPT * ts_threads, max_tiles = NBIN + ns / TILE + 1;
    uint32_t* d_remap = fx_Ep ? (uint32_t*)(A + o_remap) : nullptr;
    // the two-level sort of the windows [wb, wb + wn) (all of them, or one pass of the host-pointer pipeline)
    auto sort_windows = [&](uint32_t wb, uint32_t wn, hipStream_t st) {
      hipLaunchKernelGGL((k_part_hist<G>), dim3(KB2, wn), dim3(1024), 0, st, d_digits, d_blockcnt, ns, chunk2, NBIN, wb);
      hipLaunchKernelGGL((k_part_scan<G>), dim3(wn), dim3(1024), 0, st, d_blockcnt, d_binstart, d_tileprefix, NBIN, KB2, TILE, wb);
      hipLaunchKernelGGL((k_part_scatter<G>), dim3(KB2, wn), dim3(1024), 0, st, d_digits, d_blockcnt, d_recidx, d_reckey, ns, chunk2, HIB, NBIN, (const uint32_t*)d_remap, vw, wb);
      hipLaunchKernelGGL((k_tile_count<G>), dim3(max_tiles, wn), dim3(ts_threads), 0, st, d_reckey, d_binstart, d_tileprefix, d_counts, ns, B, HIB, NBIN, wb);
      hipLaunchKernelGGL((k_tile_sort<G>), dim3(max_tiles, wn), dim3(ts_threads), 0, st, d_recidx, d_reckey, d_binstart, d_tileprefix, d_counts,
                         d_starts, d_sorted, d_pfirst, d_pstart, d_plen, d_big, d_nbig, d_mid, d_nmid, ns, B, HIB, NBIN, SEG, PW, d_pbucket, vw, wb);
    };
    if (hin) {
      // ---- host-pointer pipeline.  Three streams.  Transfers on the copy stream, in this order (a pageable hipMemcpyAsync holds the
      // calling thread until its bytes have left, so every launch below is issued before the NEXT transfer starts):
      //   scalars (+ flags) of chunk 0 | bases of chunk 0 | scalars of chunk 1 | bases of chunk 1 | ...
      // behind each chunk's scalars, on the SORT stream: digits + two-level sort + longest-first schedule of the chunk's virtual windows
      // (the sort's scratch is indexed by virtual window: passes of different chunks share nothing) - it runs beside the accumulation
      // of the chunk before; behind each chunk's bases and its sort, on the call's stream: conversion + k_accumulate_chunk.  The
      // accumulation is the longer side of every stage from chunk 0 on (2^20 G1 terms: 0.61 ms per quarter against 0.58 ms of
      // transfers), so what the call pays on top of the resident pipeline is the first chunk's transfer.
      hipStream_t cs = side_stream_.get(), ss = sort_stream_.get();
      //  an error return inside the chunk loop must not leave the copy and sort streams reading the caller's host buffers and the arena
      struct Drain { hipStream_t a, b, c; bool armed; ~Drain() { if (armed) { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(c); } } } drain{cs, ss, stream, true};
      if (ev_copy.size() < 3 * (size_t)K + 1) {
        const size_t have = ev_copy.size();
        ev_copy.resize(3 * (size_t)K + 1, nullptr);
        for (size_t i = have; i < ev_copy.size(); i++) HIP_OK(hipEventCreateWithFlags(&ev_copy[i], hipEventDisableTiming));
      }
      hipEvent_t* ev_sc = ev_copy.data();             // [k]: chunk k's scalars are on the device
      hipEvent_t* ev_bs = ev_copy.data() + K;         // [k]: chunk k's bases are
      hipEvent_t* ev_so = ev_copy.data() + 2 * K;     // [k]: chunk k's runs, pieces and schedule are ready
      HIP_OK(hipMemsetAsync(d_counts, 0, o_zero_end - o_counts, stream));       // (the per-call fills run under the first transfer)
      HIP_OK(hipMemsetAsync(d_carrier, 0, (size_t)total * IO::XYZZ_WORDS * 4, stream));
      HIP_OK(hipEventRecord(ev_copy[3 * K], stream));
      HIP_OK(hipStreamWaitEvent(ss, ev_copy[3 * K], 0));
      constexpr size_t PT_BYTES = 2 * (size_t)IO::ARK64 * 8;
      const uint32_t cslots = (uint32_t)nw * PW;
      ArkCoord<IO::ARK64> ark_one;
      F::one().to_ark(ark_one.v);
      if (hin->ark_zero && d_inf != d_in_inf) return 2;       // (the flags are written into the engine's own buffer)
      size_t hlo = 0;                                  // the chunk's first point in the caller's arrays
      for (uint32_t k = 0; k < K; hlo += clen[k], k++) {
        const size_t lo = (size_t)k * cm, cnt = clen[k];      // ... and on the device (virtual index)
        // (the prover's queries - hin->ark_zero: a base row (0, 1) is the identity - need the chunk's bases before its digits: bases first)
        if (hin->ark_zero) {
          HIP_OK(hipMemcpyAsync((char*)d_ark_bases + lo * PT_BYTES, (const char*)hin->bases + hlo * PT_BYTES, cnt * PT_BYTES, hipMemcpyHostToDevice, cs));
          HIP_OK(hipEventRecord(ev_bs[k], cs));
        }
        // scalars -> digits, sort, schedule (sort stream)
        HIP_OK(hipMemcpyAsync((char*)d_scalars + lo * SW * 4, (const char*)hin->scalars + hlo * SW * 4, cnt * SW * 4, hipMemcpyHostToDevice, cs));
        if (hin->inf) HIP_OK(hipMemcpyAsync((char*)d_inf + lo, hin->inf + hlo, cnt, hipMemcpyHostToDevice, cs));
        HIP_OK(hipEventRecord(ev_sc[k], cs));
        HIP_OK(hipStreamWaitEvent(ss, ev_sc[k], 0));
        if (hin->ark_zero) {
          HIP_OK(hipStreamWaitEvent(ss, ev_bs[k], 0));
          hipLaunchKernelGGL((k_flag_ark_zero<G>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ss, d_ark_bases + lo * 2 * IO::ARK64,
                             hin->inf ? d_inf + lo : nullptr, d_in_inf + lo, cnt, ark_one);
        }
        if (launch_digits<SW, G::SCALAR_BITS>(c, d_scalars, d_inf, d_digits_all, (uint32_t)(lo + cnt), ss, cm, (k + 1) * cm, k * cm)) return 3;    // (lanes behind the chunk's last point: "no digit")
        sort_windows(k * (uint32_t)nw, (uint32_t)nw, ss);
        uint32_t* bins_k = d_bins + (size_t)k * BINS_STRIDE;       // longest-first schedule over the chunk's own slots (its virtual windows are adjacent)
        hipLaunchKernelGGL((k_size_hist<G>), dim3(cslots / 256 < 512 ? (cslots + 255) / 256 : 512), dim3(256), 0, ss, d_plen + (size_t)k * cslots, bins_k, cslots);
        hipLaunchKernelGGL((k_size_scan<G>), dim3(1), dim3(1024), 0, ss, bins_k, bins_k + SIZE_BINS);
        hipLaunchKernelGGL((k_size_scatter<G>), dim3((cslots + 4095) / 4096), dim3(1024), 0, ss, d_plen + (size_t)k * cslots, bins_k, d_order + (size_t)k * cslots, cslots);
        HIP_OK(hipEventRecord(ev_so[k], ss));
        if (k == 0) {
          HIP_OK(hipStreamWaitEvent(stream, ev_so[0], 0));
          HIP_OK(hipEventRecord(ev[1], stream));      // ("convert" = chunk 0's scalars, digits, sort and schedule; "sort" is empty on this path;
          HIP_OK(hipEventRecord(ev[2], stream));      //  "accumulate" = everything from here to the last chunk's end)
        }
        // bases -> conversion, accumulation (the call's stream)
        if (!hin->ark_zero) {
          HIP_OK(hipMemcpyAsync((char*)d_ark_bases + lo * PT_BYTES, (const char*)hin->bases + hlo * PT_BYTES, cnt * PT_BYTES, hipMemcpyHostToDevice, cs));
          HIP_OK(hipEventRecord(ev_bs[k], cs));
        }
        HIP_OK(hipStreamWaitEvent(stream, ev_bs[k], 0));
        if (k) HIP_OK(hipStreamWaitEvent(stream, ev_so[k], 0));
        hipLaunchKernelGGL((k_convert_bases<G>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, d_ark_bases + lo * 2 * IO::ARK64, d_bases + lo * IO::AFF_WORDS, cnt);
        hipLaunchKernelGGL((k_accumulate_chunk<G>), dim3((cslots + 255) / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart + (size_t)k * cslots, d_plen + (size_t)k * cslots,
                           d_order + (size_t)k * cslots, bins_k + SIZE_BINS, d_partials + (size_t)k * cslots * IO::XYZZ_WORDS, d_pbucket + (size_t)k * cslots, d_carrier, k ? 1u : 0u);
      }
      drain.armed = false;
    } else {
    // window shards (A/B hook CELO_SIDE_CONVERT): the conversion of ALL n bases is replicated on every shard while its sort shrinks to a
    // handful of latency-bound launches - the two are independent until the accumulation, so the conversion may run on a second stream
    const bool side = (win_cnt || MsmTuning::get().side_convert_all) && !glv && MsmTuning::get().side_convert && side_stream_.get() && ev_side[0];
    if (fx) {
      // nothing to convert: the table is in device form
    } else if (side) {
      HIP_OK(hipEventRecord(ev_side[0], stream));
      HIP_OK(hipStreamWaitEvent(side_stream_.get(), ev_side[0], 0));
      hipLaunchKernelGGL((k_convert_bases<G>), dim3((n + 255) / 256), dim3(256), 0, side_stream_.get(), d_ark_bases, d_bases, (size_t)n);
      HIP_OK(hipEventRecord(ev_side[1], side_stream_.get()));
    } else if (glv) GlvExpand<G>::launch(d_ark_bases, d_inf, d_scalars, (uint32_t)n_, d_bases, (uint32_t*)(A + o_sc2), stream);
    else hipLaunchKernelGGL((k_convert_bases<G>), dim3((n + 255) / 256), dim3(256), 0, stream, d_ark_bases, d_bases, (size_t)n);
    HIP_OK(hipEventRecord(ev[1], stream));
    // ---- sort
    if (fx && fx_Ep) {
      HIP_OK(hipMemsetAsync(d_digits_all, 0xFF, (size_t)n * nw_all * 2, stream));             // every slot "no digit" until a record lands in it
      HIP_OK(hipMemsetAsync(d_fx_cnt + 128, 0, 128 * 4, stream));                               // the rows' cursors
      const uint32_t E32 = (uint32_t)fx->E();
      hipLaunchKernelGGL((k_fixed_place<G>), dim3((E32 + 1023) / 1024 < 4096 ? (E32 + 1023) / 1024 : 4096), dim3(1024), 0, stream, d_fx_v8, d_fx_dg, d_fx_cnt + 128, d_digits_all,
                         d_remap, E32, n, fx->NV);
    } else if (fx) hipLaunchKernelGGL((k_fixed_digits<SW, G::SCALAR_BITS>), dim3((fx->n + 255) / 256), dim3(256), 0, stream, d_scalars, fx->tinf, d_digits_all, fx->n, (uint32_t)n_,
                               fx->cf, fx->W, fx->NV, fx->M);
    else if (glv) { if (launch_digits<4, GlvExpand<G>::BITS>(c, (const uint32_t*)(A + o_sc2), nullptr, d_digits_all, n, stream)) return 3; }
    else if (launch_digits<SW, G::SCALAR_BITS>(c, d_scalars, d_inf, d_digits_all, n, stream)) return 3;
    HIP_OK(hipMemsetAsync(d_counts, 0, o_zero_end - o_counts, stream));
    sort_windows(0, (uint32_t)nw, stream);
    // ---- work items, longest first
    hipLaunchKernelGGL((k_size_hist<G>), dim3(slots / 256 < 512 ? (slots + 255) / 256 : 512), dim3(256), 0, stream, d_plen, d_bins, slots);
    hipLaunchKernelGGL((k_size_scan<G>), dim3(1), dim3(1024), 0, stream, d_bins, d_nwork);
    hipLaunchKernelGGL((k_size_scatter<G>), dim3((slots + 4095) / 4096), dim3(1024), 0, stream, d_plen, d_bins, d_order, slots);
    if (side) HIP_OK(hipStreamWaitEvent(stream, ev_side[1], 0));
    HIP_OK(hipEventRecord(ev[2], stream));
    // ---- accumulate (grid covers every slot; lanes beyond the number of non-empty pieces exit)
    if constexpr (BaCfg<G>::enabled) if (use_ba) {
      uint32_t* d_ba_pts = (uint32_t*)(A + o_ba_pts);
      const int levels = MsmTuning::get().ba_levels;
      if (ba_occ == 2)
        hipLaunchKernelGGL((k_ba_levels<G, 2>), dim3(ba_lanes / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_ba_pts,
                           (uint32_t*)(A + o_ba_pref), ba_lanes, levels, ba_pref_slots);
      else
        hipLaunchKernelGGL((k_ba_levels<G, 1>), dim3(ba_lanes / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_ba_pts,
                           (uint32_t*)(A + o_ba_pref), ba_lanes, levels, ba_pref_slots);
      hipLaunchKernelGGL((k_accumulate_ba<G>), dim3((slots + 255) / 256), dim3(256), 0, stream, d_ba_pts, d_pstart, d_plen, d_order, d_nwork, d_partials, levels);
    }
    if (!use_ba) launch_accumulate<G>(slots, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_partials);
    }
    HIP_OK(hipEventRecord(ev[3], stream));
    // ---- bucket reduction
    // (a window shard cuts EVERY bucket in two or three: one group of lanes per bucket of the call, not 21504 groups striding over them)
    const uint32_t mid_blocks = win_cnt ? (total + 20) / 21 < 16384 ? (total + 20) / 21 : 16384 : 1024;
    const uint32_t cfirst = hin ? 1u : 0u;      // host-pointer pipeline: a bucket's first piece is its carrier and stays out of the folds
    if (lane_bitsum) hipLaunchKernelGGL((k_combine_mid_lanes<G>), dim3(mid_blocks), dim3(64), 0, stream, d_mid, d_nmid, d_counts, d_pfirst, d_partials, d_piecesof, SEG, cfirst);
    else hipLaunchKernelGGL((k_combine_mid<G>), dim3(256), dim3(128), 0, stream, d_mid, d_nmid, d_counts, d_pfirst, d_partials, d_piecesof, SEG, cfirst);
    hipLaunchKernelGGL((k_combine_big<G>), dim3(256), dim3(256), 0, stream, d_big, d_nbig, d_counts, d_pfirst, d_partials, d_piecesof, SEG, cfirst);
    // the leaves of the reduction: the buckets' pieces, or (host-pointer pipeline) the carrier table, one slot per bucket
    const uint32_t* d_leaf = d_partials;
    const uint32_t* d_leaf_counts = d_counts;
    if (hin) {
      hipLaunchKernelGGL((k_merge_carried<G>), dim3((total + 127) / 128), dim3(128), 0, stream, d_counts, d_pfirst, d_partials, d_carrier, total, K, SEG);
      d_leaf = d_carrier; d_leaf_counts = nullptr;
    }
    {
      // work area (in points): [0, res_pts) results, then two launch-alternating halves of half_pts
      struct Arr { uint32_t at, per_window; bool born; int level; };  // `born`: odd list not yet halved (read strided from its level)
      uint32_t half_at[2] = {res_pts, res_pts + half_pts};
      Arr tree = {0, B, true, LB};            // current tree level (level LB = the buckets themselves)
      std::vector<Arr> lists;                 // pending odd lists
      for (int t = 1; t <= LB; t++) {
        BitsumJobs jobs;
        jobs.njobs = 0;
        uint32_t cursor = half_at[t & 1], total_out = 0;
        auto push = [&](uint32_t src, uint32_t outs_per_window, uint32_t mode, int result_slot) -> uint32_t {
          const uint32_t outs = outs_per_window * (uint32_t)nw;
          uint32_t dst;
          if (result_slot >= 0) dst = (uint32_t)result_slot * (uint32_t)nw;
          else { dst = cursor; cursor += outs; }
          const int j = (int)jobs.njobs++;
          total_out += outs;
          jobs.end[j] = total_out; jobs.src[j] = src; jobs.dst[j] = dst; jobs.mode[j] = mode;
          return dst;
        };
        std::vector<Arr> next_lists;
        // the odd list of the current tree level is born now (level >= 2: at least two odd nodes per window)
        if (tree.level >= 2) {
          const uint32_t outs = tree.per_window / 4;
          const uint32_t dst = push(tree.at, outs, tree.level == LB ? 3u : 1u, outs == 1 ? tree.level : -1);
          if (outs > 1) next_lists.push_back({dst, outs, false, tree.level});
        } else {  // level 1: O_1 = node(1, 1)
          push(tree.at, 1, 4u, 1);
        }
        for (const Arr& L : lists) {
          const uint32_t outs = L.per_window / 2;
          const uint32_t dst = push(L.at, outs, 0u, outs == 1 ? L.level : -1);
          if (outs > 1) next_lists.push_back({dst, outs, false, L.level});
        }
        {  // next tree level
          const uint32_t outs = tree.per_window / 2;
          const uint32_t dst = push(tree.at, outs, tree.level == LB ? 2u : 0u, outs == 1 ? 0 : -1);
          tree = {dst, outs, true, tree.level - 1};
        }
        lists.swap(next_lists);
        if (lane_bitsum && total_out <= (win_cnt ? MsmTuning::get().bitsum_lanes_max_shard : BITSUM_LANES_MAX))
          hipLaunchKernelGGL((k_bitsum_lanes<G>), dim3((total_out + 20) / 21), dim3(64), 0, stream, d_leaf, d_leaf_counts, d_pfirst, d_piecesof, SEG, d_work, jobs);
        else
          hipLaunchKernelGGL((k_bitsum<G>), dim3((total_out + 127) / 128), dim3(128), 0, stream, d_leaf, d_leaf_counts, d_pfirst, d_piecesof, SEG,
                             d_work, jobs);
      }
      hipLaunchKernelGGL((k_results_to_ark<G>), dim3((4 * res_pts + 63) / 64), dim3(64), 0, stream, d_work, res_pts);
    }
    HIP_OK(hipEventRecord(ev[4], stream));
    HIP_OK(hipMemcpyAsync(h_out, d_work, (size_t)res_pts * IO::XYZZ_WORDS * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipEventRecord(ev[5], stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipGetLastError());
    (void)hipEventElapsedTime(&tm.convert, ev[0], ev[1]);
    (void)hipEventElapsedTime(&tm.sort, ev[1], ev[2]);
    (void)hipEventElapsedTime(&tm.accumulate, ev[2], ev[3]);
    (void)hipEventElapsedTime(&tm.reduce, ev[3], ev[4]);
    (void)hipEventElapsedTime(&tm.total, ev[0], ev[5]);
    last_c = c; last_nw = nw; last_buckets = total; last_glv = glv;
    // ---- host epilogue: total = sum_w 2^(c w) (node_w + sum_l 2^(LB-l) O_{w,l}): one Horner pass, c doublings and c additions per
    // window (the results arrive as arkworks limbs), as a list of steps: on AVX-512 IFMA where the CPU has it (host_ifma.cpp: the
    // products of one point operation eight at a time, 0.24 -> ~0.1 ms for 253-bit scalars), else - or if that path meets equal or
    // opposite operands, which it does not handle - on 64-bit limbs (host64.h)
    typedef typename HostField<F>::type HF;
    const uint64_t* h64 = reinterpret_cast<const uint64_t*>(h_out);
    constexpr size_t PT64 = (size_t)IO::XYZZ_WORDS / 2;
    horner_steps.clear();
    const int kn = pl.kn;                                       // the top kn of the nw_all windows are c - 1 bits wide (k_digits)
    (void)sbits;
    if (fx) {
      // total = sum_v c_v node_v + sum_l 2^(15 - l) (sum_v O_{v,l}) + sum_v node_v,  c_v = v M: ONE chain over the bit positions t from the top
      // of the largest c_v down to 0 - at step t the accumulator doubles, then takes node_v of every v with bit t of c_v set and, for
      // t <= 14, the O_{v, 15 - t} of every virtual window
      int top = LB - 1;
      while (((uint64_t)(nw - 1) * fx->M) >> (top + 1)) top++;
      for (int t = top; t >= 0; t--) {
        horner_steps.push_back(-1);
        for (int v = 0; v < nw; v++) if ((((uint64_t)v * fx->M) >> t) & 1) horner_steps.push_back(v | HORNER_NODBL);
        if (t <= LB - 1) for (int v = 0; v < nw; v++) horner_steps.push_back(((LB - t) * nw + v) | HORNER_NODBL);
      }
      for (int v = 0; v < nw; v++) horner_steps.push_back(v | HORNER_NODBL);
    } else
    for (int w = nw - 1; w >= 0; w--) {
      horner_steps.push_back(-1);
      for (int l = (w0 + w >= nw_all - kn ? 2 : 1); l <= LB; l++) horner_steps.push_back(l * nw + w);
      horner_steps.push_back(w | HORNER_NODBL);
    }
    auto run_list = [&](const int32_t* steps, int count) {
      if (IfmaHorner<F>::fn && celo_ifma_available()) {
        uint64_t r[4 * IO::ARK64];
        int inf = 0;
        if (IfmaHorner<F>::fn(h64, PT64, steps, count, r, &inf) == 0) return inf ? HXyzz<HF>::identity() : HXyzz<HF>::load(r, IO::ARK64);
      }
      return host64_horner<HF>(h64, PT64, IO::ARK64, steps, count);
    };
    // The pass is linear in its windows: a group of windows run from the identity gives P_j, and the whole is ((P_0 2^d1 + P_1) 2^d2 +
    // P_2) ... with d_j the doublings of group j's steps.  For the fields whose host products are slow - Fq2 (three 6-limb products and
    // their reductions per product: 0.65 ms of every G2 call) and the 12-limb field of BW6-761 (0.6-0.8 ms) - the window groups run on
    // HOST_HORNER_THREADS threads side by side and only the joining doublings stay serial: 0.65 -> 0.3 ms per G2 MSM, more than a
    // quarter of a call below 2^16 terms.  The 6-limb prime field stays on one thread (0.15 ms: the joins would cost what the split saves).
    constexpr int HT = (sizeof(HF) > 6 * 8) ? HOST_HORNER_THREADS : 1;
    HXyzz<HF> total_pt;
    if (HT > 1 && host_threads && (fx ? horner_steps.size() >= 64 : nw >= 2 * HT)) {
      int start[HT + 1], dbls[HT];
      if (fx) {
        // fixed base (late round 4): the chain over the bit positions is linear in them too - a group of consecutive positions run from
        // the identity gives P_j and the join is the same.  Cut BEFORE a doubling step, balancing the additions (the 15 positions that
        // take every virtual window's level sums carry most of them): 0.8 -> 0.4 ms of BW6-761 host work per call at cf = 20.
        int adds_total = 0;
        for (int k = 0; k < (int)horner_steps.size(); k++) adds_total += horner_steps[k] >= 0 ? 1 : 0;
        int g = 0, adds = 0;
        start[0] = 0;
        for (int k = 0; k < (int)horner_steps.size(); k++) {
          if (horner_steps[k] < 0 && g + 1 < HT && k > start[g] && adds * HT >= (g + 1) * adds_total) start[++g] = k;
          adds += horner_steps[k] >= 0 ? 1 : 0;
        }
        while (g + 1 < HT) start[++g] = (int)horner_steps.size();       // (fewer cuts than threads: empty groups, the identity)
        start[HT] = (int)horner_steps.size();
        for (int j = 0; j < HT; j++) {
          dbls[j] = 0;
          for (int k = start[j]; k < start[j + 1]; k++) if (horner_steps[k] < 0 || !(horner_steps[k] & HORNER_NODBL)) dbls[j]++;
        }
      } else {   // window boundaries in the step list: a window's steps end with its NODBL entry
        int wdone = 0, g = 0;
        start[0] = 0;
        for (int k = 0; k < (int)horner_steps.size(); k++) {
          if (horner_steps[k] >= 0 && (horner_steps[k] & HORNER_NODBL)) {
            wdone++;
            if (wdone == (g + 1) * nw / HT && g + 1 < HT) start[++g] = k + 1;
          }
        }
        start[HT] = (int)horner_steps.size();
        for (int j = 0; j < HT; j++) {
          dbls[j] = 0;
          for (int k = start[j]; k < start[j + 1]; k++) if (horner_steps[k] < 0 || !(horner_steps[k] & HORNER_NODBL)) dbls[j]++;
        }
      }
      HXyzz<HF> part[HT];
      std::thread th[HT - 1];
      bool started[HT - 1];
      for (int j = 1; j < HT; j++) {
        started[j - 1] = true;
        try { th[j - 1] = std::thread([&, j] { part[j] = run_list(horner_steps.data() + start[j], start[j + 1] - start[j]); }); }
        catch (const std::system_error&) { started[j - 1] = false; }      // no thread to be had: that group runs here, serially
      }
      part[0] = run_list(horner_steps.data() + start[0], start[1] - start[0]);
      for (int j = 1; j < HT; j++) {
        if (started[j - 1]) th[j - 1].join();
        else part[j] = run_list(horner_steps.data() + start[j], start[j + 1] - start[j]);
      }
      total_pt = part[0];
      for (int j = 1; j < HT; j++) {
        for (int d = 0; d < dbls[j]; d++) total_pt = hxyzz_dbl(total_pt);
        hxyzz_add(total_pt, part[j]);
      }
    } else {
      total_pt = run_list(horner_steps.data(), (int)horner_steps.size());
    }
    if (out_xyzz) {
      if (total_pt.is_identity()) memset(out_xyzz, 0, 4 * IO::ARK64 * 8);
      else { total_pt.X.store(out_xyzz); total_pt.Y.store(out_xyzz + IO::ARK64); total_pt.ZZ.store(out_xyzz + 2 * IO::ARK64); total_pt.ZZZ.store(out_xyzz + 3 * IO::ARK64); }
    }
    if (out_jac) write_host_jacobian(total_pt, out_jac);
    return 0;
  }
  typedef typename HostField<F>::type HostF;
  static void write_host_jacobian(const HXyzz<HostF>& pt, uint64_t* out_jac) {
    if (pt.is_identity()) { write_identity(out_jac); return; }
    (pt.X * pt.ZZ).store(out_jac);                 // (X ZZ, Y ZZZ, ZZ) is a Jacobian representative with Z = ZZ
    (pt.Y * pt.ZZZ).store(out_jac + IO::ARK64);
    pt.ZZ.store(out_jac + 2 * IO::ARK64);
  }

  // host-pointer entry: stages inputs into (cached) device buffers, then run_device
  int run_host(const uint64_t* bases, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t* out_jac, hipStream_t stream) {
    return run_host_windows(bases, inf, scalars, n, 0, 0, out_jac, nullptr, stream);
  }
  int run_host_windows(const uint64_t* bases, const uint8_t* inf, const uint64_t* scalars, size_t n, int win_lo, int win_cnt, uint64_t* out_jac,
                       uint64_t* out_xyzz, hipStream_t stream) {
    if (n == 0) return run_device_windows(nullptr, nullptr, nullptr, 0, win_lo, win_cnt, out_jac, out_xyzz, stream);
    // the pipelined form (run_device_windows' HostIn) from 2^18 terms up, in chunks of at least 2^17 points (the prover's entry points -
    // ark_zero_identity: the flags come from the bases - send a chunk's bases before its scalars); the GLV split (its expansion reads bases
    // and scalars together) and window shards keep the plain form below: three transfers, then the resident pipeline
    const int ovr = host_chunks_override().load();
    // measured on the MI355X box (profiles/r5_host_pointer_*.json; DESIGN.md section 4 "Host-pointer pipeline"): 4 chunks for the 253-bit
    // groups, 8 for BW6-761, the first one cut in halves once
    // ... and from 2^21 terms chunks of about 2^18 points, up to 16 of them (gpurun_out/r5w: G1 2^21 7.43 / 7.18 / 7.19 ms with 4 / 8 / 16 chunks, 2^22
    // 13.85 / 12.80 / 12.34, 2^23 - / 23.79 / 23.01 (32: 23.79), 2^24 51.3 / 46.6 / 43.4 (32: 43.9, 64: 46.0) = 1.05 x resident; BW6-761 2^22 68.4 / 67.8
    // / 70.6 with 8 / 16 / 32; the Fq2 group, whose accumulation is three times the transfer, 2^22 37.7 / 37.7 / 39.1 with 4 / 8 / 16)
    const bool fq2_group = sizeof(F) > 14 * sizeof(uint32_t) && G::SCALAR_BITS <= 256;
    uint32_t by_size = (uint32_t)(n >> (fq2_group ? 19 : 18));
    const uint32_t lo_k = G::SCALAR_BITS > 256 ? 8u : 4u, hi_k = fq2_group ? 8u : 16u;
    by_size = by_size < lo_k ? lo_k : by_size > hi_k ? hi_k : by_size;
    uint32_t chunks = ovr >= 0 ? (uint32_t)(ovr & 0xFF) : MsmTuning::get().host_chunks != 0xFFFFFFFFu ? MsmTuning::get().host_chunks : by_size;
    uint32_t head_split = ovr >= 0 && ((ovr >> 8) & 15) ? (uint32_t)((ovr >> 8) & 15) - 1u : MsmTuning::get().host_head_split != 0xFFFFFFFFu ? MsmTuning::get().host_head_split : HOST_HEAD_SPLIT_DEFAULT;
    uint32_t tail_split = ovr >= 0 && ((ovr >> 12) & 15) ? (uint32_t)((ovr >> 12) & 15) - 1u : MsmTuning::get().host_tail_split != 0xFFFFFFFFu ? MsmTuning::get().host_tail_split : HOST_TAIL_SPLIT_DEFAULT;
    if (chunks > 64) chunks = 64;
    // chunks of at least 2^17 points (2^16 when the count was set by hand - tests): measured (gpurun_out/r5n, G1): 2^17 terms 1.30 ms plain /
    // 1.46 in two chunks, 2^18 1.95 / 1.90 in two / 2.16 in four, 2^19 3.23 / 2.72 in two / 2.55 in four - a chunk's sort pass and the
    // accumulation's last round are fixed costs of ~0.08 ms
    const size_t min_chunk_log = ovr >= 0 ? 16 : 17;
    if (chunks > (n >> min_chunk_log)) chunks = (uint32_t)(n >> min_chunk_log);
    // (the prover runs its four MSMs at once: they hide each other's transfers already, and chunking them costs more than it hides below 2^21
    // rows - tools/bench_prover_host.py, witness-like assignment: 2^19 rows per query 37.8 ms plain / 45.3 pipelined, 2^20 58.0 / 66.8, 2^21 106.6 / 91.2)
    // (the subgroup entry's GLV split reads bases and scalars together and is not pipelined: from 2^19 terms the transfers it waits for cost
    // more than the split saves - tools/bench_host_subgroup.py, G1: 2^18 1.76 ms split / 1.87 pipelined, 2^19 3.08 / 2.68, 2^20 5.88 / 4.18, 2^21
    // 11.6 / 7.07 - so a host-pointer call of that size takes the pipelined plain form; same group element)
    const bool glv_plan = plan(n).glv;
    const bool glv_off = glv_plan && !ark_zero_identity && !win_cnt && n >= (size_t(1) << 19);
    // (the default head split only where its halves keep 2^17 points: gpurun_out/r5z, G1 2^18 terms in two chunks 1.69 ms whole / 1.82 with the
    // first one halved, 2^19 in four 2.55 / 2.69; 2^20 in four 4.14 / 4.04)
    if (!(ovr >= 0 && ((ovr >> 8) & 15)) && MsmTuning::get().host_head_split == 0xFFFFFFFFu && chunks && n / chunks < (size_t(1) << 18)) head_split = 0;
    bool pipelined = chunks >= (ovr >= 0 ? 1u : 2u) && !win_cnt && (!glv_plan || glv_off) && n < (size_t(1) << 30) && (!ark_zero_identity || ovr >= 0 || n >= (size_t(1) << 21));
    size_t need = n;                 // staging capacity in points: the pipelined form addresses chunk k at k cm
    if (pipelined) {
      uint32_t cm, clen[HOST_CHUNKS_MAX];
      need = (size_t)host_chunk_plan(n, chunks, head_split, tail_split, cm, clen) * cm;
      uint64_t nw_run;
      { struct GlvOff { bool& f; bool was; GlvOff(bool& x, bool off) : f(x), was(x) { if (off) f = false; } ~GlvOff() { f = was; } } g(use_glv, glv_plan);
        nw_run = (uint64_t)plan(n).nw; }       // (the pipelined run switches GLV off - the window count checked here is the one it will use)
      if ((uint64_t)need * nw_run >= (uint64_t(1) << 32)) { pipelined = false; need = n; }      // (the holes would overflow the 32-bit run offsets: plain form)
    }
    const size_t n_real = n;
    n = need;
    if (n > cap_in) {
      if (d_in_bases) (void)hipFree(d_in_bases);
      if (d_in_scalars) (void)hipFree(d_in_scalars);
      if (d_in_inf) (void)hipFree(d_in_inf);
      d_in_bases = nullptr; d_in_scalars = nullptr; d_in_inf = nullptr; cap_in = 0;
      HIP_OK(hipMalloc(&d_in_bases, n * 2 * IO::ARK64 * 8));
      HIP_OK(hipMalloc(&d_in_scalars, n * SW * 4));
      HIP_OK(hipMalloc(&d_in_inf, n));
      cap_in = n;
    }
    n = n_real;
    if (pipelined) {
      const HostIn hin = {bases, inf, scalars, chunks, head_split, tail_split, ark_zero_identity};
      struct GlvOff { bool& f; bool was; GlvOff(bool& x, bool off) : f(x), was(x) { if (off) f = false; } ~GlvOff() { f = was; } } glv_guard(use_glv, glv_plan);
      return run_device_windows(d_in_bases, inf || ark_zero_identity ? d_in_inf : nullptr, (const uint32_t*)d_in_scalars, n, 0, 0, out_jac, out_xyzz, stream, nullptr, &hin);
    }
    HIP_OK(hipMemcpyAsync(d_in_bases, bases, n * 2 * IO::ARK64 * 8, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_in_scalars, scalars, n * SW * 4, hipMemcpyHostToDevice, stream));
    if (inf) HIP_OK(hipMemcpyAsync(d_in_inf, inf, n, hipMemcpyHostToDevice, stream));
    if (ark_zero_identity) {       // the prover's queries: rows (0, 1) are arkworks' encoding of the identity
      ArkCoord<IO::ARK64> one;
      F::one().to_ark(one.v);
      hipLaunchKernelGGL((k_flag_ark_zero<G>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_in_bases, inf ? d_in_inf : nullptr, d_in_inf, n, one);
      return run_device_windows(d_in_bases, d_in_inf, (const uint32_t*)d_in_scalars, n, win_lo, win_cnt, out_jac, out_xyzz, stream);
    }
    return run_device_windows(d_in_bases, inf ? d_in_inf : nullptr, (const uint32_t*)d_in_scalars, n, win_lo, win_cnt, out_jac, out_xyzz, stream);
  }

  // ---- fixed-base form (FixedTable): the scalars come from the host (staged into a buffer of their own) or are resident
  int run_fixed(const FixedTable& T, const void* scalars, size_t n_sc, int resident, uint64_t* out_jac, hipStream_t stream) {
    if (n_sc > T.n) n_sc = T.n;                                   // VariableBaseMSM zips bases with scalars: the shorter side decides
    if (n_sc == 0) { write_identity(out_jac); return 0; }
    const uint32_t* d_sc = (const uint32_t*)scalars;
    if (!resident) {
      if (n_sc > cap_fx) {
        if (d_fx_scalars) (void)hipFree(d_fx_scalars);
        d_fx_scalars = nullptr; cap_fx = 0;
        HIP_OK(hipMalloc(&d_fx_scalars, n_sc * SW * 4));
        cap_fx = n_sc;
      }
      HIP_OK(hipMemcpyAsync(d_fx_scalars, scalars, n_sc * SW * 4, hipMemcpyHostToDevice, stream));
      d_sc = (const uint32_t*)d_fx_scalars;
    }
    return run_device_windows(nullptr, nullptr, d_sc, n_sc, 0, 0, out_jac, nullptr, stream, &T);
  }
  // window size of a key's table: buckets ~ entries / 64 within [2^15, 2^19] (measured sweep: DESIGN.md section 4 "Fixed base")
  static int fixed_window_bits(size_t n) {
    int lg = 0;
    while ((size_t(1) << lg) < n * ((G::SCALAR_BITS + 16) / 16)) lg++;
    int cf = lg - 4;      // 2^21 BW6-761 terms: 21 (26.3 ms against 33.3 variable-base; 26.8 at 20); 2^20 G1 terms: 20 (3.08 against 3.3 ms; 4.75 at 21)
    // (21 only where the additions are expensive enough to pay for twice the buckets: the 28-limb fields; measured with the digits compacted
    // by virtual window - before that 20 was the optimum everywhere, profiles/r4_fixed_sweep.txt)
    const int top = sizeof(F) > 14 * sizeof(uint32_t) ? 21 : 20;
    return cf < 16 ? 16 : cf > top ? top : cf;
  }
  // builds T (device memory of the calling thread's device) from n affine bases in arkworks layout; d_* are DEVICE pointers
  static int fixed_build(const uint64_t* d_ark_bases, const uint8_t* d_inf, size_t n_, int cf, FixedTable* T, hipStream_t stream) {
    if (n_ == 0 || n_ >= (size_t(1) << 27)) return 2;
    // the pipeline's 32-bit run offsets bound a table: n W < 2^31 entries and (uncompacted form: every entry in every virtual window)
    // n W NV < 2^32.  With the automatic choice the window steps down until both hold (2^24 BW6-761 terms at the preferred cf = 21
    // are W = 18, NV = 33: 10^10 slots - the key's size the header names for the prover; cf = 19 fits); an explicit cf that does not fit is refused.
    auto fits = [&](int c_) {
      const uint64_t W_ = (uint64_t)((G::SCALAR_BITS + c_) / c_), M_ = c_ == 16 ? 32768u : 32767u, NV_ = ((uint64_t(1) << (c_ - 1)) - 1u) / M_ + 1u;
      return (uint64_t)n_ * W_ < (uint64_t(1) << 31) && (uint64_t)n_ * W_ * NV_ < (uint64_t(1) << 32);
    };
    if (cf == 0) {
      cf = fixed_window_bits(n_);
      while (cf > 16 && !fits(cf)) cf--;
    }
    if (cf < 16 || cf > 22 || !fits(cf)) return 2;
    const uint32_t n = (uint32_t)n_, W = (uint32_t)((G::SCALAR_BITS + cf) / cf);      // W cf >= SCALAR_BITS + 1: room for the signed recoding's carry
    const uint32_t M = cf == 16 ? 32768u : 32767u, NV = ((1u << (cf - 1)) - 1u) / M + 1u;
    const size_t E = (size_t)n * W;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    T->n = n; T->W = W; T->NV = NV; T->M = M; T->cf = cf; T->device = api_device();
    T->bytes = E * IO::AFF_WORDS * 4 + E;
    HIP_OK(hipMalloc(&T->table, E * IO::AFF_WORDS * 4));
    HIP_OK(hipMalloc(&T->tinf, E));
    HIP_OK(hipEventRecord(e0, stream));
    hipLaunchKernelGGL((k_convert_bases<G>), dim3((n + 255) / 256), dim3(256), 0, stream, d_ark_bases, T->table, (size_t)n);
    hipLaunchKernelGGL((k_fixed_first_flags<G>), dim3((n + 255) / 256), dim3(256), 0, stream, d_inf, T->tinf, n);
    for (uint32_t j = 1; j < W; j++)
      hipLaunchKernelGGL((k_fixed_next<G>), dim3((n + 127) / 128), dim3(128), 0, stream, T->table + (size_t)(j - 1) * n * IO::AFF_WORDS, T->tinf + (size_t)(j - 1) * n,
                         T->table + (size_t)j * n * IO::AFF_WORDS, T->tinf + (size_t)j * n, n, cf);
    HIP_OK(hipEventRecord(e1, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipGetLastError());
    (void)hipEventElapsedTime(&T->build_ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
  }

  // ---- batched small MSMs (host pointers).  offsets[m+1]; every instance must have <= 1024 points (larger instances go
  // through run_host one by one).  out: m Jacobian results (arkworks form).
  static constexpr uint32_t BATCH_MAX_N = 1024;
  MsmTimings tm_batch;
  int batch_bits = 0;   // length of the longest scalar of the last batched call
  // bits_hint > 0: the length of the longest scalar of the NEXT batched call, measured by the caller on the same scalars (spares the
  // k_scalar_or round trip: with the chip full of another engine's accumulation that small kernel and its synchronisation waited 16 ms
  // inside batch_verify_strict, and the G1 leg was enqueued only then); measured_bits: what the last call used, before clamping
  int bits_hint = 0, measured_bits = 0;
  int run_batch_host(const uint64_t* bases, const uint8_t* inf, const uint64_t* scalars, const uint32_t* offsets, size_t m,
                     uint64_t* out, hipStream_t stream) {
    return run_batch(bases, inf, scalars, 0, offsets, m, out, nullptr, stream);
  }
  // resident = 0: bases / inf / scalars are HOST pointers (staged into the arena); 1: DEVICE pointers (used in place).
  // out: host buffer for the m Jacobian results, or nullptr to leave them on the device: *d_out_ret then points at them (in this
  // engine's arena, valid until its next call) and the call returns with the work ENQUEUED on `stream`, not finished - the
  // caller chains its consumer behind it (batch verification: normalise + pairing inputs without a host round trip).
  int run_batch(const uint64_t* bases, const uint8_t* inf, const uint64_t* scalars, int resident, const uint32_t* offsets, size_t m,
                uint64_t* out, uint64_t** d_out_ret, hipStream_t stream) {
    measured_bits = 0;                                     // stays 0 on the paths that do not measure (no hint to hand on)
    if (m == 0) return 0;
    const uint32_t total_pts = offsets[m];
    uint32_t max_n = 0;
    for (size_t p = 0; p < m; p++) {
      uint32_t k = offsets[p + 1] - offsets[p];
      if (k > max_n) max_n = k;
    }
    if (max_n > BATCH_MAX_N || total_pts == 0) {
      // An instance beyond the per-workgroup sort (or a call whose instances are all empty): every instance goes through the big
      // pipeline, one after the other (still the GPU; Batch::verify takes any number of signers and accepts an empty batch,
      // crates/bls-crypto/src/bls/batch.rs:44-84).  The big pipeline carves this engine's arena, so in the chained form the m
      // results are collected on the host and put into a buffer of their own; the call is then synchronous, which the chained
      // contract allows (the consumer waits on the stream either way).
      std::vector<uint64_t> hres;
      uint64_t* dst = out;
      if (!out) { hres.resize(m * 3 * IO::ARK64); dst = hres.data(); }
      for (size_t p = 0; p < m; p++) {
        const uint32_t lo = offsets[p], k = offsets[p + 1] - lo;
        const uint8_t* pi = inf ? inf + lo : nullptr;
        const int rc = resident ? run_device(bases + (size_t)lo * 2 * IO::ARK64, pi, (const uint32_t*)scalars + (size_t)lo * SW, k, dst + p * 3 * IO::ARK64, stream)
                                : run_host(bases + (size_t)lo * 2 * IO::ARK64, pi, scalars + (size_t)lo * (SW / 2), k, dst + p * 3 * IO::ARK64, stream);
        if (rc) return rc;
      }
      tm_batch = tm;
      if (!out) {
        const size_t bytes = m * 3 * IO::ARK64 * 8;
        if (bytes > side_out_bytes) {
          if (d_side_out) (void)hipFree(d_side_out);
          d_side_out = nullptr; side_out_bytes = 0;
          HIP_OK(hipMalloc(&d_side_out, bytes));
          side_out_bytes = bytes;
        }
        HIP_OK(hipMemcpyAsync(d_side_out, hres.data(), bytes, hipMemcpyHostToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));
        if (d_out_ret) *d_out_ret = d_side_out;
        side_path = true;
      }
      return 0;
    }
    side_path = false;
    // window size ~ log2(n) - 3 (measured on 4096 x 256, 136-bit exponents: c = 5 beats 6 and 7; the per-(instance,window)
    // running sums and the per-instance Horner are latency-bound, so fewer buckets per window win)
    auto window_for = [&](uint32_t inst_n) {
      int c_ = force_c ? force_c : 3;
      if (!force_c) { while (c_ < 7 && (16u << c_) <= inst_n) c_++; }
      if (c_ > 7) c_ = 7;
      if (c_ < 3) c_ = 3;
      return c_;
    };
    // GLS split (G2 of BLS12-377, subgroup points only: gls_subgroup_points): k = d0 + d1 x + d2 x^2 + d3 x^3 in base x, the curve
    // parameter, and [x]P = psi(P) - so an instance of n points with b-bit scalars becomes one of nd n points psi^j(P) with 64-bit
    // scalars, nd = 2 (b <= 126), 3 (b <= 189: Batch::verify's 136-bit exponents) or 4.  Same group element; what changes is the
    // shape: a quarter to a half of the windows (13 x 5 bits instead of 28) over a larger instance, which takes a wider window
    // (c = 6 for 768 points: 11 windows) - 14 % fewer mixed additions, less than half the per-(instance, window) running sums and
    // a Horner chain of 66 doublings instead of 140.
    const int gls_max = (GlsExpand<G>::AVAILABLE && (gls_subgroup_points || gls_force) && use_gls) ? 4 : 1;
    auto gls_digits = [&](int bits) {
      if (gls_max == 1 || bits <= 64 || bits > G::SCALAR_BITS) return 1;
      const int nd_ = bits <= 126 ? 2 : bits <= 189 ? 3 : 4;
      return (size_t)nd_ * max_n <= BATCH_MAX_N ? nd_ : 1;
    };
    // stage the inputs at the front of the arena, then let the scalars decide the number of windows: bits = length of the longest
    // scalar present.  The arena is sized for the worst layout (full-length scalars, or the largest split that fits) beforehand.
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    const size_t o_in_b = take(resident ? 0 : (size_t)total_pts * 2 * IO::ARK64 * 8), o_in_s = take(resident ? 0 : (size_t)total_pts * SW * 4),
                 o_in_i = take(resident ? 0 : total_pts + 8);
    const size_t o_off = take((m + 1) * 4), o_or = take(64 * 4);
    const size_t front = off;
    auto rest = [&](int c_, int nw_, int nd_, size_t* o) {     // the part of the arena that depends on the layout; returns its end
      size_t save = off;
      off = front;
      const uint32_t B_ = 1u << (c_ - 1);
      const size_t tot = (size_t)nd_ * total_pts;
      const size_t nvw_ = m * (size_t)nw_, nb = nvw_ * B_, en = tot * nw_;
      o[0] = take(tot * IO::AFF_WORDS * 4); o[1] = take(en * 4 + 16);
      o[2] = take(nb * 4); o[3] = take(nb * 4); o[4] = take(nb * 4);
      o[5] = take((size_t)SIZE_BINS * 4 + 256); o[6] = take(nb * IO::XYZZ_WORDS * 4);
      o[7] = take(nvw_ * IO::XYZZ_WORDS * 4); o[8] = take(m * 3 * IO::ARK64 * 8);
      o[9] = take(nd_ > 1 ? tot * 16 : 0); o[10] = take(nd_ > 1 ? tot + 8 : 0); o[11] = take(nd_ > 1 ? (m + 1) * 4 : 0);
      const size_t end = off;
      off = save;
      return end;
    };
    size_t o[12];
    {
      const int c1 = window_for(max_n);
      const int nw1 = (G::SCALAR_BITS + c1) / c1;
      if (m * (size_t)nw1 * (size_t(1) << (c1 - 1)) >= (size_t(1) << 31) || (size_t)total_pts * nw1 >= (size_t(1) << 32)) return 2;
      size_t need = rest(c1, nw1, 1, o);
      for (int nd_ = 2; nd_ <= gls_max; nd_++) {
        if ((size_t)nd_ * max_n > BATCH_MAX_N) break;
        const int c2 = window_for((uint32_t)nd_ * max_n);
        const size_t e2 = rest(c2, (64 + c2) / c2, nd_, o);
        if (e2 > need) need = e2;
      }
      if (ensure(need)) return 1;
    }
    {
      char* A0 = arena;
      if (!resident) HIP_OK(hipMemcpyAsync(A0 + o_in_s, scalars, (size_t)total_pts * SW * 4, hipMemcpyHostToDevice, stream));
      int bits = 1;
      if (bits_hint > 0) bits = bits_hint;          // the caller measured these very scalars already (batch verification: the other leg's engine)
      else {
        HIP_OK(hipMemsetAsync(A0 + o_or, 0, 64 * 4, stream));
        hipLaunchKernelGGL((k_scalar_or<SW>), dim3(2048), dim3(256), 0, stream, resident ? (const uint32_t*)scalars : (const uint32_t*)(A0 + o_in_s),
                           (size_t)total_pts * SW, (uint32_t*)(A0 + o_or));
        uint32_t h_or[SW];
        HIP_OK(hipMemcpyAsync(h_or, A0 + o_or, SW * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (int k = SW - 1; k >= 0; k--) if (h_or[k]) { bits = 32 * k + 32 - __builtin_clz(h_or[k]); break; }
      }
      measured_bits = bits;
      if (bits > G::SCALAR_BITS && gls_digits(bits) == 1) bits = G::SCALAR_BITS;
      batch_bits = bits;
    }
    const int nd = gls_digits(batch_bits);
    const uint32_t eff_max_n = (uint32_t)nd * max_n, eff_total = (uint32_t)nd * total_pts;
    const int eff_bits = nd > 1 ? 64 : (batch_bits > G::SCALAR_BITS ? G::SCALAR_BITS : batch_bits);
    const int c = window_for(eff_max_n);
    const uint32_t B = 1u << (c - 1);
    const int nw = (eff_bits + c) / c;
    const size_t nvw = m * (size_t)nw, nbuckets = nvw * B;
    if (nbuckets >= (size_t(1) << 31) || (size_t)eff_total * nw >= (size_t(1) << 32)) return 2;
    (void)rest(c, nw, nd, o);
    char* A = arena;
    const uint64_t* d_in_b = resident ? bases : (const uint64_t*)(A + o_in_b);
    const uint32_t* d_in_s = resident ? (const uint32_t*)scalars : (const uint32_t*)(A + o_in_s);
    const uint8_t* d_in_i = resident ? inf : (const uint8_t*)(A + o_in_i);
    uint32_t* d_off = (uint32_t*)(A + o_off);
    uint32_t* d_bases = (uint32_t*)(A + o[0]); uint32_t* d_sorted = (uint32_t*)(A + o[1]);
    uint32_t* d_pstart = (uint32_t*)(A + o[2]); uint32_t* d_plen = (uint32_t*)(A + o[3]); uint32_t* d_order = (uint32_t*)(A + o[4]);
    uint32_t* d_bins = (uint32_t*)(A + o[5]); uint32_t* d_nwork = d_bins + SIZE_BINS;
    uint32_t* d_partials = (uint32_t*)(A + o[6]); uint32_t* d_wsum = (uint32_t*)(A + o[7]); uint64_t* d_out = (uint64_t*)(A + o[8]);
    if (!resident) {
      HIP_OK(hipMemcpyAsync(A + o_in_b, bases, (size_t)total_pts * 2 * IO::ARK64 * 8, hipMemcpyHostToDevice, stream));
      if (inf) HIP_OK(hipMemcpyAsync(A + o_in_i, inf, total_pts, hipMemcpyHostToDevice, stream));
    }
    HIP_OK(hipMemcpyAsync(d_off, offsets, (m + 1) * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipEventRecord(ev[0], stream));
    HIP_OK(hipMemsetAsync(d_bins, 0, (size_t)SIZE_BINS * 4 + 256, stream));
    if (nd > 1) {
      uint32_t* d_sc2 = (uint32_t*)(A + o[9]);
      uint8_t* d_inf2 = (uint8_t*)(A + o[10]);
      uint32_t* d_off2 = (uint32_t*)(A + o[11]);
      gls_off.resize(m + 1);
      for (size_t p = 0; p <= m; p++) gls_off[p] = (uint32_t)nd * offsets[p];
      HIP_OK(hipMemcpyAsync(d_off2, gls_off.data(), (m + 1) * 4, hipMemcpyHostToDevice, stream));
      GlsExpand<G>::launch(d_in_b, inf ? d_in_i : nullptr, d_in_s, d_off, (uint32_t)m, max_n, nd, batch_bits, d_bases, d_sc2, inf ? d_inf2 : nullptr, stream);
      HIP_OK(hipEventRecord(ev[1], stream));
      if (launch_batch_sort<4>(c, eff_max_n, d_sc2, inf ? d_inf2 : nullptr, d_off2, d_sorted, d_pstart, d_plen, (uint32_t)m, nw, stream)) return 3;
    } else {
      hipLaunchKernelGGL((k_convert_bases<G>), dim3((total_pts + 255) / 256), dim3(256), 0, stream, d_in_b, d_bases, (size_t)total_pts);
      HIP_OK(hipEventRecord(ev[1], stream));
      if (launch_batch_sort<SW>(c, max_n, d_in_s, inf ? d_in_i : nullptr, d_off, d_sorted, d_pstart, d_plen, (uint32_t)m, nw, stream)) return 3;
    }
    last_gls_digits = nd;
    const uint32_t slots = (uint32_t)nbuckets;
    hipLaunchKernelGGL((k_size_hist<G>), dim3(slots / 256 < 2048 ? (slots + 255) / 256 : 2048), dim3(256), 0, stream, d_plen, d_bins, slots);
    hipLaunchKernelGGL((k_size_scan<G>), dim3(1), dim3(1024), 0, stream, d_bins, d_nwork);
    hipLaunchKernelGGL((k_size_scatter<G>), dim3((slots + 4095) / 4096), dim3(1024), 0, stream, d_plen, d_bins, d_order, slots);
    HIP_OK(hipEventRecord(ev[2], stream));
    launch_accumulate<G>(slots, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_partials);
    HIP_OK(hipEventRecord(ev[3], stream));
    hipLaunchKernelGGL((k_batch_reduce<G>), dim3(((uint32_t)nvw + 127) / 128), dim3(128), 0, stream, d_partials, d_plen, d_wsum, B, (uint32_t)nvw);
    if (lane_horner) BatchHornerLanes<G>::launch(d_wsum, d_out, (uint32_t)nw, (uint32_t)c, (uint32_t)m, stream);
    else hipLaunchKernelGGL((k_batch_horner<G>), dim3(((uint32_t)m + 127) / 128), dim3(128), 0, stream, d_wsum, d_out, (uint32_t)nw, (uint32_t)c, (uint32_t)m);
    HIP_OK(hipEventRecord(ev[4], stream));
    if (out) HIP_OK(hipMemcpyAsync(out, d_out, m * 3 * IO::ARK64 * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipEventRecord(ev[5], stream));
    last_c = c; last_nw = nw; last_buckets = (uint32_t)nbuckets;
    if (d_out_ret) *d_out_ret = d_out;
    if (!out) { HIP_OK(hipGetLastError()); return 0; }     // chained form: the caller synchronises and may call collect_batch_timings()
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipGetLastError());
    collect_batch_timings();
    return 0;
  }
  void collect_batch_timings() {     // after the stream has drained
    if (side_path) { tm = tm_batch; return; }
    (void)hipEventElapsedTime(&tm_batch.convert, ev[0], ev[1]);
    (void)hipEventElapsedTime(&tm_batch.sort, ev[1], ev[2]);
    (void)hipEventElapsedTime(&tm_batch.accumulate, ev[2], ev[3]);
    (void)hipEventElapsedTime(&tm_batch.reduce, ev[3], ev[4]);
    (void)hipEventElapsedTime(&tm_batch.total, ev[0], ev[5]);
    tm = tm_batch;
  }

  MsmTimings tm;
  int last_c = 0, last_nw = 0;
  uint32_t last_buckets = 0;

  static void write_identity(uint64_t* out) {
    // arkworks GroupProjective::zero() = (0, 1, 0); only z == 0 is significant
    Xyzz<F> id = Xyzz<F>::identity();
    write_jacobian(id, out);
  }
  // (X*ZZ, Y*ZZZ, ZZ) is a Jacobian representative of (X/ZZ, Y/ZZZ) with Z = ZZ
  static void write_jacobian(const Xyzz<F>& p, uint64_t* out) {
    if (p.is_identity() || p.ZZ.is_zero_mod_p()) {
      F::zero().to_ark(out);
      F::one().to_ark(out + IO::ARK64);
      F::zero().to_ark(out + 2 * IO::ARK64);
      return;
    }
    F::mul(p.X, p.ZZ).to_ark(out);
    F::mul(p.Y, p.ZZZ).to_ark(out + IO::ARK64);
    p.ZZ.to_ark(out + 2 * IO::ARK64);
  }

 private:
  char* arena = nullptr;  // one device allocation, carved per call (sizes depend on n and the window size)
  size_t arena_bytes = 0;
  uint64_t* d_in_bases = nullptr;
  uint64_t* d_in_scalars = nullptr;
  uint8_t* d_in_inf = nullptr;
  uint32_t* h_out = nullptr;
  uint64_t* d_fx_scalars = nullptr;    // staged scalars of the fixed-base form (run_fixed)
  size_t cap_fx = 0;
  uint8_t* fxs = nullptr;              // fixed base: the per-entry (window id, digit) records and the windows' counts, taken before the arena is laid out
  size_t fxs_bytes = 0;
  uint64_t* d_side_out = nullptr;      // results of a chained batch call that went through the big pipeline (run_batch)
  size_t side_out_bytes = 0;
  bool side_path = false;
  static constexpr size_t H_OUT_POINTS = 17 * 64;   // pinned result buffer: (LB + 1) * windows points; checked per call
  OwnedStream stream_;
  OwnedStream side_stream_;            // window shards: the base conversion beside the sort (CELO_SIDE_CONVERT)
  OwnedStream sort_stream_;            // host-pointer pipeline: digits + sort + schedule of chunk k beside the accumulation of chunk k - 1
  hipEvent_t ev_side[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> ev_copy;     // host-pointer pipeline: per chunk - scalars sent, bases sent, sorted; + one for the per-call fills
  std::vector<int32_t> horner_steps;   // the host epilogue's step list (host64.h), rebuilt per call
  std::vector<uint32_t> gls_off;       // instance offsets of the expanded (GLS) batch
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap_in = 0;

  int ensure(size_t bytes) {
    if (!ev[0]) {
      for (int i = 0; i < 6; i++) HIP_OK(hipEventCreate(&ev[i]));
      for (int i = 0; i < 2; i++) HIP_OK(hipEventCreateWithFlags(&ev_side[i], hipEventDisableTiming));
    }
    if (!h_out) {
      HIP_OK(hipHostMalloc(&h_out, H_OUT_POINTS * IO::XYZZ_WORDS * 4));
    }
    if (bytes > arena_bytes) {
      if (arena) (void)hipFree(arena);
      arena = nullptr; arena_bytes = 0;
      HIP_OK(hipMalloc(&arena, bytes));
      arena_bytes = bytes;
    }
    return 0;
  }

  template <int SWX, int CB, int PT> int launch_batch_sort_cp(const uint32_t* sc, const uint8_t* inf, const uint32_t* off, uint32_t* sorted,
                                                     uint32_t* pstart, uint32_t* plen, uint32_t m, int nw, hipStream_t st) {
    hipLaunchKernelGGL((k_batch_sort<SWX, CB, PT>), dim3(m), dim3(256), 0, st, sc, inf, off, sorted, pstart, plen, nw);
    return 0;
  }
  template <int SWX, int CB> int launch_batch_sort_c(uint32_t max_n, const uint32_t* sc, const uint8_t* inf, const uint32_t* off, uint32_t* sorted,
                                            uint32_t* pstart, uint32_t* plen, uint32_t m, int nw, hipStream_t st) {
    if (max_n <= 256) return launch_batch_sort_cp<SWX, CB, 1>(sc, inf, off, sorted, pstart, plen, m, nw, st);
    if (max_n <= 512) return launch_batch_sort_cp<SWX, CB, 2>(sc, inf, off, sorted, pstart, plen, m, nw, st);
    return launch_batch_sort_cp<SWX, CB, 4>(sc, inf, off, sorted, pstart, plen, m, nw, st);
  }
  template <int SWX> int launch_batch_sort(int c, uint32_t max_n, const uint32_t* sc, const uint8_t* inf, const uint32_t* off, uint32_t* sorted,
                        uint32_t* pstart, uint32_t* plen, uint32_t m, int nw, hipStream_t st) {
    switch (c) {
      case 3: return launch_batch_sort_c<SWX, 3>(max_n, sc, inf, off, sorted, pstart, plen, m, nw, st);
      case 4: return launch_batch_sort_c<SWX, 4>(max_n, sc, inf, off, sorted, pstart, plen, m, nw, st);
      case 5: return launch_batch_sort_c<SWX, 5>(max_n, sc, inf, off, sorted, pstart, plen, m, nw, st);
      case 6: return launch_batch_sort_c<SWX, 6>(max_n, sc, inf, off, sorted, pstart, plen, m, nw, st);
      case 7: return launch_batch_sort_c<SWX, 7>(max_n, sc, inf, off, sorted, pstart, plen, m, nw, st);
      default: return 1;
    }
  }
  template <int SWX, int BITS, int CB> int launch_digits_c(const uint32_t* sc, const uint8_t* inf, uint16_t* digits, uint32_t n, hipStream_t st, uint32_t m, uint32_t npad, uint32_t ibase) {
    constexpr int NW = (BITS + CB) / CB;
    constexpr int KN = NW * CB - (BITS + 1);     // 0 <= KN < CB <= NW for every window size in use
    if constexpr (KN > 0 && KN < NW) {
      if (narrow_top(CB)) { hipLaunchKernelGGL((k_digits<SWX, CB, NW, KN, BITS>), dim3((npad - ibase + 255) / 256), dim3(256), 0, st, sc, inf, digits, n, m, npad, ibase); return 0; }
    }
    hipLaunchKernelGGL((k_digits<SWX, CB, NW, 0, BITS>), dim3((npad - ibase + 255) / 256), dim3(256), 0, st, sc, inf, digits, n, m, npad, ibase);
    return 0;
  }
  // m, npad: the chunked layout (k_digits); 0 = plain
  template <int SWX, int BITS> int launch_digits(int c, const uint32_t* sc, const uint8_t* inf, uint16_t* digits, uint32_t n, hipStream_t st, uint32_t m = 0, uint32_t npad = 0, uint32_t ibase = 0) {
    if (!m) { m = n; npad = n; }
    switch (c) {
      case 4: return launch_digits_c<SWX, BITS, 4>(sc, inf, digits, n, st, m, npad, ibase);
      case 5: return launch_digits_c<SWX, BITS, 5>(sc, inf, digits, n, st, m, npad, ibase);
      case 6: return launch_digits_c<SWX, BITS, 6>(sc, inf, digits, n, st, m, npad, ibase);
      case 7: return launch_digits_c<SWX, BITS, 7>(sc, inf, digits, n, st, m, npad, ibase);
      case 8: return launch_digits_c<SWX, BITS, 8>(sc, inf, digits, n, st, m, npad, ibase);
      case 9: return launch_digits_c<SWX, BITS, 9>(sc, inf, digits, n, st, m, npad, ibase);
      case 10: return launch_digits_c<SWX, BITS, 10>(sc, inf, digits, n, st, m, npad, ibase);
      case 11: return launch_digits_c<SWX, BITS, 11>(sc, inf, digits, n, st, m, npad, ibase);
      case 12: return launch_digits_c<SWX, BITS, 12>(sc, inf, digits, n, st, m, npad, ibase);
      case 13: return launch_digits_c<SWX, BITS, 13>(sc, inf, digits, n, st, m, npad, ibase);
      case 14: return launch_digits_c<SWX, BITS, 14>(sc, inf, digits, n, st, m, npad, ibase);
      case 15: return launch_digits_c<SWX, BITS, 15>(sc, inf, digits, n, st, m, npad, ibase);
      case 16: return launch_digits_c<SWX, BITS, 16>(sc, inf, digits, n, st, m, npad, ibase);
      default: return 1;
    }
  }
};

}  // namespace celo
