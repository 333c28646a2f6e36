// Radix-2 number-theoretic transform for gfx950, a template over the prime field: Fr(BW6-761) = Fq(BLS12-377) (377 bits,
// 2-adicity 46; 14 limbs) for the epoch proof and Fr(BLS12-377) (253 bits, 2-adicity 47; 10 limbs) for the hash-helper proof
// (crates/epoch-snark/src/api/prover.rs:78 and :83-118).
//
// Replaces ark-poly 0.1's Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place (un-vendored dependency,
// Cargo.lock:213-215) as used by ark-groth16's witness map inside create_proof_no_zk, which the reference calls at
// crates/epoch-snark/src/api/prover.rs:78,112 - SURVEY.md section 8f row f3, the step before the prover's MSMs.
// The caller hands over the domain generator (and coset generator / n^-1 when it wants them applied), so no arkworks
// constant is restated here.
//
// Shape (HBM-bound work: 64-B elements, one multiplication per butterfly):
//   k_ntt_load   arkworks Montgomery (48 B) -> 28-bit-limb device form (64 B), optional coset pre-scaling x_i *= g^i
//   k_ntt_tile4  decimation-in-frequency, in place, EIGHT (then six or four) butterfly levels per launch: a workgroup stages
//                1024 elements in LDS (word-planar, 56 KB: two workgroups per CU = two waves per SIMD) and runs radix-4
//                rounds on them, the first reading HBM, the last writing it; 2^20 points = three launches (8 + 8 + 4)
//   k_ntt_pass   the remaining (< 4) levels, up to THREE per launch: a lane loads 8 elements spaced by the lowest level's distance,
//                runs 12 butterflies in registers, stores them back.  Loads and stores are 64-B vectors, consecutive lanes
//                touch consecutive elements.
//   k_ntt_store  bit-reversal gather, optional coset post-scaling x_i *= g^i and final scale, -> arkworks Montgomery
// Twiddles: one table W[k] = omega^k, k < n/2, in device form, built once per (omega, n) from two 1024-entry power
// tables and kept by the engine (32 MB at 2^20: stays in the Infinity Cache across passes).
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include <cstdio>
#include <cstring>
#include "fp.h"
#include "fp_consts.h"
#include "runtime.h"

namespace celo {

typedef Fp<P377> Fr761;   // the scalar field of BW6-761 is the base field of BLS12-377: 14 limbs padded to 16 words, 64-B elements
typedef Fp<P253> Fr377;   // the scalar field of BLS12-377: 10 limbs padded to 12 words, 48-B elements
// every kernel below is a template over the field FR: device elements are FR::WORDS words (uint4 loads / stores), arkworks
// elements FR::ARK64 u64, LDS tiles FR::L word planes

#if defined(__HIPCC__)
// pw[i] = base^(i * stride) for i < 1024 (one lane each: square-and-multiply over the 10 bits of i)
template <class FR>
__global__ void __launch_bounds__(256) k_ntt_powers(const uint32_t* __restrict__ base_dev, uint32_t* __restrict__ pw) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 1024) return;
  const FR b = FR::load(base_dev);
  FR acc = FR::one();
  for (int bit = 9; bit >= 0; bit--) {
    acc = FR::sqr(acc);
    if ((i >> bit) & 1) acc = FR::mul(acc, b);
  }
  FR::wred(acc).store(pw + (size_t)i * FR::WORDS);
}
// out[k] = lo[k & 1023] * hi[k >> 10]   (lo = base^i, hi = base^(1024 i)); used for the twiddle table and coset powers
template <class FR>
__global__ void __launch_bounds__(256) k_ntt_table(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                                   uint32_t* __restrict__ out, uint32_t count) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  FR v = FR::load(lo + (size_t)(k & 1023) * FR::WORDS);
  if (count > 1024) v = FR::mul(v, FR::load(hi + (size_t)(k >> 10) * FR::WORDS));
  FR::wred(v).store(out + (size_t)k * FR::WORDS);
}
template <class FR>
__global__ void __launch_bounds__(256) k_ntt_load(const uint64_t* __restrict__ ark, uint32_t* __restrict__ work, uint32_t n,
                                                  const uint32_t* __restrict__ glo, const uint32_t* __restrict__ ghi) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  FR v = FR::from_ark(ark + (size_t)i * FR::ARK64);
  if (glo) {
    v = FR::mul(v, FR::load(glo + (size_t)(i & 1023) * FR::WORDS));
    if (n > 1024) v = FR::mul(v, FR::load(ghi + (size_t)(i >> 10) * FR::WORDS));
  }
  FR::wred(v).store(work + (size_t)i * FR::WORDS);
}
// Format conversion fused into the first and last butterfly launches: the first reads the caller's arkworks elements (and
// applies the coset pre-scaling), the last writes arkworks elements at the bit-reversed index (coset post-scaling, final scale).
// Null pointers = plain device-form loads / stores on the work array.
struct NttIo {
  const uint64_t* ark_in;    // first launch only
  uint64_t* ark_out;         // last launch only
  const uint32_t* glo;       // coset power tables g^i, g^(1024 i) (pre-scaling with ark_in, post-scaling with ark_out), or null
  const uint32_t* ghi;
  const uint32_t* scale;     // with ark_out, or null
};
template <class FR>
__device__ __forceinline__ FR ntt_coset(const FR& v, const NttIo& io, uint32_t i, uint32_t log_n) {
  FR r = FR::mul(v, FR::load(io.glo + (size_t)(i & 1023) * FR::WORDS));
  if (log_n > 10) r = FR::mul(r, FR::load(io.ghi + (size_t)(i >> 10) * FR::WORDS));
  return r;
}
template <class FR>
__device__ __forceinline__ FR ntt_ld(const uint32_t* __restrict__ work, const NttIo& io, size_t idx, uint32_t log_n) {
  if (!io.ark_in) return FR::load(work + idx * FR::WORDS);
  FR v = FR::from_ark(io.ark_in + idx * FR::ARK64);
  if (io.glo) v = ntt_coset<FR>(v, io, (uint32_t)idx, log_n);
  return FR::wred(v);
}
template <class FR>
__device__ __forceinline__ void ntt_st(uint32_t* __restrict__ work, const NttIo& io, size_t idx, uint32_t log_n, const FR& v) {
  if (!io.ark_out) { v.store(work + idx * FR::WORDS); return; }
  const uint32_t r = __brev((uint32_t)idx) >> (32 - log_n);      // position idx holds output number bitrev(idx)
  FR w = v;
  if (io.glo) w = ntt_coset<FR>(w, io, r, log_n);
  if (io.scale) w = FR::mul(w, FR::load(io.scale));
  w.to_ark(io.ark_out + (size_t)r * FR::ARK64);
}
// one decimation-in-frequency butterfly: (x, y) <- (x + y, (x - y) * omega^k)
template <class FR>
__device__ __forceinline__ void ntt_bf(FR& x, FR& y, uint32_t k, const uint32_t* __restrict__ tw) {
  const FR u = x, v = y;
  x = FR::wred(FR::add(u, v));
  const FR d = FR::template sub<4, 1>(u, v);
  y = FR::mul(d, FR::load(tw + (size_t)k * FR::WORDS));     // tw[0] = 1
}
// R butterfly levels s_hi, s_hi-1, .., s_hi-R+1 (level s pairs i and i + 2^s) on 2^R elements per lane.  The elements are
// named scalars (not an array) so that they stay in VGPRs; local bit b of the element number <-> global level s_lo + b, and the
// twiddle exponent of a butterfly whose upper element has global index i is (i mod 2^s) * n / 2^(s+1).
template <class FR, int R>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) k_ntt_pass(uint32_t* __restrict__ work, const uint32_t* __restrict__ tw,
                                                                                         uint32_t log_n, int s_hi, NttIo io) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = 1u << log_n;
  if (t >= (n >> R)) return;
  const int s_lo = s_hi - R + 1;
  const uint32_t m = 1u << s_lo;
  const uint32_t low = t & (m - 1);
  const size_t i0 = (size_t)(t >> s_lo) * ((size_t)m << R) + low;       // element j of this lane: i0 + j * m
#define NTT_K(j, lvl) ((low + ((uint32_t)((j) & ((1 << (lvl)) - 1)) << s_lo)) << (log_n - 1 - (uint32_t)(s_lo + (lvl))))
#define NTT_LD(j) ntt_ld<FR>(work, io, i0 + (size_t)(j) * m, log_n)
#define NTT_ST(j, v) ntt_st<FR>(work, io, i0 + (size_t)(j) * m, log_n, v)
  if constexpr (R == 3) {
    FR a0 = NTT_LD(0), a1 = NTT_LD(1), a2 = NTT_LD(2), a3 = NTT_LD(3), a4 = NTT_LD(4), a5 = NTT_LD(5), a6 = NTT_LD(6), a7 = NTT_LD(7);
    ntt_bf<FR>(a0, a4, NTT_K(0, 2), tw); ntt_bf<FR>(a1, a5, NTT_K(1, 2), tw); ntt_bf<FR>(a2, a6, NTT_K(2, 2), tw); ntt_bf<FR>(a3, a7, NTT_K(3, 2), tw);
    ntt_bf<FR>(a0, a2, NTT_K(0, 1), tw); ntt_bf<FR>(a1, a3, NTT_K(1, 1), tw); ntt_bf<FR>(a4, a6, NTT_K(4, 1), tw); ntt_bf<FR>(a5, a7, NTT_K(5, 1), tw);
    ntt_bf<FR>(a0, a1, NTT_K(0, 0), tw); ntt_bf<FR>(a2, a3, NTT_K(2, 0), tw); ntt_bf<FR>(a4, a5, NTT_K(4, 0), tw); ntt_bf<FR>(a6, a7, NTT_K(6, 0), tw);
    NTT_ST(0, a0); NTT_ST(1, a1); NTT_ST(2, a2); NTT_ST(3, a3); NTT_ST(4, a4); NTT_ST(5, a5); NTT_ST(6, a6); NTT_ST(7, a7);
  } else if constexpr (R == 2) {
    FR a0 = NTT_LD(0), a1 = NTT_LD(1), a2 = NTT_LD(2), a3 = NTT_LD(3);
    ntt_bf<FR>(a0, a2, NTT_K(0, 1), tw); ntt_bf<FR>(a1, a3, NTT_K(1, 1), tw);
    ntt_bf<FR>(a0, a1, NTT_K(0, 0), tw); ntt_bf<FR>(a2, a3, NTT_K(2, 0), tw);
    NTT_ST(0, a0); NTT_ST(1, a1); NTT_ST(2, a2); NTT_ST(3, a3);
  } else {
    FR a0 = NTT_LD(0), a1 = NTT_LD(1);
    ntt_bf<FR>(a0, a1, NTT_K(0, 0), tw);
    NTT_ST(0, a0); NTT_ST(1, a1);
  }
#undef NTT_LD
#undef NTT_ST
#undef NTT_K
}
// twiddle exponent of the butterfly at sub-level b inside a radix-2^k group whose elements sit at rows r0 + j * 2^L0 (r0 has those
// k row bits clear): (global index of the upper element mod 2^s) * n / 2^(s+1) with s = s_lo + L0 + b
#define NTT_TK(j, b) ((((r0 & ((1u << L0) - 1)) + ((uint32_t)((j) & ((1 << (b)) - 1)) << L0)) << s_lo) + lo) << (log_n - 1 - (uint32_t)(s_lo + L0 + (b)))
// ---- 2*NR levels per launch with 1024-element tiles (56 KB of LDS): two workgroups per CU = two waves per SIMD, which is what
// the multiplier needs to run at its pipe rate (one wave issues a v_mad_u64_u32 every 9.6 cycles, two reach ~5).  NR radix-4
// rounds on local row bits (2NR-1, 2NR-2), .., (1, 0); four elements per lane; a workgroup owns C = 1024 / 4^NR adjacent tiles
// of 4^NR rows (NR = 4: four 256-row tiles, eight levels; NR = 3: sixteen 64-row tiles; NR = 2: sixty-four 16-row tiles).
constexpr int NTT_TILE4_ELEMS = 1024;
template <int NR> __device__ __forceinline__ uint32_t ntt_swz4(uint32_t e) {
  if constexpr (NR == 4) { const uint32_t h = (e >> 6) & 3u; return e ^ (h << 2) ^ (h << 4); }   // 4 columns: rows reach the bank bits
  else return e;                                                                                  // >= 16 columns: lanes are column-contiguous
}
template <class FR, int NR> __device__ __forceinline__ void ntt_lds_put4(uint32_t* lds, uint32_t e, const FR& v) {
  const uint32_t p = ntt_swz4<NR>(e);
#pragma unroll
  for (int k = 0; k < FR::L; k++) lds[k * NTT_TILE4_ELEMS + p] = v.l[k];
}
template <class FR, int NR> __device__ __forceinline__ FR ntt_lds_get4(const uint32_t* lds, uint32_t e) {
  const uint32_t p = ntt_swz4<NR>(e);
  FR v;
#pragma unroll
  for (int k = 0; k < FR::L; k++) v.l[k] = lds[k * NTT_TILE4_ELEMS + p];
  return v;
}
#define NTT_RADIX4(a0, a1, a2, a3)                                        \
  ntt_bf<FR>(a0, a2, NTT_TK(0, 1), tw); ntt_bf<FR>(a1, a3, NTT_TK(1, 1), tw);     \
  ntt_bf<FR>(a0, a1, NTT_TK(0, 0), tw); ntt_bf<FR>(a2, a3, NTT_TK(2, 0), tw);
// round Q of a tile launch (local row bits 2Q+1, 2Q; rows r0 + j * 4^Q): the first round (Q = NR-1) reads HBM, the last (Q = 0)
// writes it, the ones in between exchange through LDS.  A template over Q so that every shift count is a compile-time constant.
template <class FR, int NR, int Q>
__device__ __forceinline__ void ntt_tile_round(uint32_t* lds, uint32_t* __restrict__ work, const uint32_t* __restrict__ tw, const NttIo& io,
                                               uint32_t log_n, int s_lo, uint32_t lo, size_t i0, uint32_t c, uint32_t u) {
  constexpr int LT = 2 * NR, L0 = 2 * Q;
  constexpr uint32_t C = 1024u >> LT, d = 1u << L0;
  const uint32_t r0 = ((u >> L0) << (L0 + 2)) | (u & ((1u << L0) - 1));
  FR a0, a1, a2, a3;
  if constexpr (Q == NR - 1) {
    a0 = ntt_ld<FR>(work, io, i0 + ((size_t)r0 << s_lo), log_n);           a1 = ntt_ld<FR>(work, io, i0 + ((size_t)(r0 + d) << s_lo), log_n);
    a2 = ntt_ld<FR>(work, io, i0 + ((size_t)(r0 + 2 * d) << s_lo), log_n); a3 = ntt_ld<FR>(work, io, i0 + ((size_t)(r0 + 3 * d) << s_lo), log_n);
  } else {
    __syncthreads();
    a0 = ntt_lds_get4<FR, NR>(lds, r0 * C + c);           a1 = ntt_lds_get4<FR, NR>(lds, (r0 + d) * C + c);
    a2 = ntt_lds_get4<FR, NR>(lds, (r0 + 2 * d) * C + c); a3 = ntt_lds_get4<FR, NR>(lds, (r0 + 3 * d) * C + c);
  }
  NTT_RADIX4(a0, a1, a2, a3)
  if constexpr (Q == 0) {
    ntt_st<FR>(work, io, i0 + ((size_t)r0 << s_lo), log_n, a0);           ntt_st<FR>(work, io, i0 + ((size_t)(r0 + d) << s_lo), log_n, a1);
    ntt_st<FR>(work, io, i0 + ((size_t)(r0 + 2 * d) << s_lo), log_n, a2); ntt_st<FR>(work, io, i0 + ((size_t)(r0 + 3 * d) << s_lo), log_n, a3);
  } else {
    ntt_lds_put4<FR, NR>(lds, r0 * C + c, a0);           ntt_lds_put4<FR, NR>(lds, (r0 + d) * C + c, a1);
    ntt_lds_put4<FR, NR>(lds, (r0 + 2 * d) * C + c, a2); ntt_lds_put4<FR, NR>(lds, (r0 + 3 * d) * C + c, a3);
    ntt_tile_round<FR, NR, Q - 1>(lds, work, tw, io, log_n, s_lo, lo, i0, c, u);
  }
}
template <class FR, int NR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_ntt_tile4(uint32_t* __restrict__ work, const uint32_t* __restrict__ tw,
                                                                                              uint32_t log_n, int s_top, NttIo io) {
  extern __shared__ uint32_t lds[];
  constexpr int LT = 2 * NR;                       // levels per launch = log2(rows per tile)
  constexpr uint32_t C = 1024u >> LT;              // tiles per workgroup
  const int s_lo = s_top - (LT - 1);
  const uint32_t c = threadIdx.x & (C - 1), u = threadIdx.x / C;       // u in [0, rows / 4)
  const uint32_t inst = blockIdx.x * C + c;
  const uint32_t lo = inst & ((1u << s_lo) - 1), hi = inst >> s_lo;
  const size_t i0 = ((size_t)hi << (s_top + 1)) + lo;      // row 0 of this lane's tile; row r is element i0 + (r << s_lo)
  ntt_tile_round<FR, NR, NR - 1>(lds, work, tw, io, log_n, s_lo, lo, i0, c, u);
}
#undef NTT_RADIX4
#undef NTT_TK

template <class FR>
__global__ void __launch_bounds__(256) k_ntt_store(const uint32_t* __restrict__ work, uint64_t* __restrict__ ark, uint32_t log_n,
                                                   const uint32_t* __restrict__ glo, const uint32_t* __restrict__ ghi,
                                                   const uint32_t* __restrict__ scale_dev) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = 1u << log_n;
  if (i >= n) return;
  const uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0;
  FR v = FR::load(work + (size_t)r * FR::WORDS);
  if (glo) {
    v = FR::mul(v, FR::load(glo + (size_t)(i & 1023) * FR::WORDS));
    if (n > 1024) v = FR::mul(v, FR::load(ghi + (size_t)(i >> 10) * FR::WORDS));
  }
  if (scale_dev) v = FR::mul(v, FR::load(scale_dev));
  v.to_ark(ark + (size_t)i * FR::ARK64);
}

#define NTT_HIP_OK(x)                                                                                           \
  do {                                                                                                          \
    hipError_t e_ = (x);                                                                                        \
    if (e_ != hipSuccess) {                                                                                     \
      fprintf(stderr, "[celo-amd] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);         \
      return 1;                                                                                                 \
    }                                                                                                           \
  } while (0)

struct NttTimings { float load = 0, passes = 0, store = 0, total = 0; int npasses = 0; };

template <class FR> class NttEngine {
 public:
  ~NttEngine() { release(); }
  void release() {
    for (void** p : {(void**)&d_work, (void**)&d_tw, (void**)&d_small, (void**)&d_io}) if (*p) { (void)hipFree(*p); *p = nullptr; }
    cap_work = cap_tw = cap_io = 0; tw_log_n = 0;
    for (int i = 0; i < 4; i++) if (ev[i]) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
  }
  NttTimings tm;
  hipStream_t own_stream() { return stream_.get(); }
  int max_radix_log2 = 3;   // butterfly levels per register-only pass (tuning hook: 1..3)
  bool use_tiles = true;    // LDS-tiled passes (tuning hook)
  bool fuse_io = true;      // format conversion fused into the first / last butterfly launch (tuning hook)
  // data_dev: n = 2^log_n elements, arkworks Montgomery, DEVICE memory, transformed in place.
  // omega: domain generator of order n (its inverse for an inverse transform).  coset: nullptr or g: x_i *= g^i before
  // (coset_after = 0) or after (1) the transform.  scale: nullptr or a factor applied to every output (n^-1).
  int run_device(uint64_t* data_dev, unsigned log_n, const uint64_t* omega6, const uint64_t* coset6, int coset_after, const uint64_t* scale6,
                 hipStream_t stream) {
    if (log_n > 28) return 2;
    const uint32_t n = 1u << log_n;
    if (prepare(log_n, omega6, coset6, scale6, stream)) return 1;
    uint32_t* glo = coset6 ? small(2) : nullptr;
    uint32_t* ghi = coset6 ? small(3) : nullptr;
    // plan the butterfly launches: (kind, levels): kind 4/3/2 = k_ntt_tile4<kind>, kind 0 = k_ntt_pass<levels>
    int kinds[16], lv[16], np = 0;
    {
      int s = (int)log_n;
      if (use_tiles && n >= (uint32_t)NTT_TILE4_ELEMS) {
        while (s >= 8) { kinds[np] = 4; lv[np++] = 8; s -= 8; }
        if (s >= 6) { kinds[np] = 3; lv[np++] = 6; s -= 6; }
        else if (s >= 4) { kinds[np] = 2; lv[np++] = 4; s -= 4; }
      }
      while (s > 0) { const int r = s >= max_radix_log2 ? max_radix_log2 : s; kinds[np] = 0; lv[np++] = r; s -= r; }
    }
    const bool fused = fuse_io && np >= 2;      // the first launch reads the caller's array, the last one writes it: they must differ
    NTT_HIP_OK(hipEventRecord(ev[0], stream));
    if (!fused) hipLaunchKernelGGL((k_ntt_load<FR>), dim3((n + 255) / 256), dim3(256), 0, stream, data_dev, d_work, n, (coset6 && !coset_after) ? glo : nullptr, ghi);
    NTT_HIP_OK(hipEventRecord(ev[1], stream));
    int s = (int)log_n - 1;
    for (int i = 0; i < np; i++) {
      NttIo io = {nullptr, nullptr, nullptr, nullptr, nullptr};
      if (fused && i == 0) { io.ark_in = data_dev; if (coset6 && !coset_after) { io.glo = glo; io.ghi = ghi; } }
      if (fused && i == np - 1) { io.ark_out = data_dev; if (coset6 && coset_after) { io.glo = glo; io.ghi = ghi; } io.scale = scale6 ? small(4) : nullptr; }
      const uint32_t blocks = n / NTT_TILE4_ELEMS;
      const size_t lds_bytes = (size_t)FR::L * NTT_TILE4_ELEMS * 4;
      if (kinds[i] == 4) hipLaunchKernelGGL((k_ntt_tile4<FR, 4>), dim3(blocks), dim3(256), lds_bytes, stream, d_work, d_tw, log_n, s, io);
      else if (kinds[i] == 3) hipLaunchKernelGGL((k_ntt_tile4<FR, 3>), dim3(blocks), dim3(256), lds_bytes, stream, d_work, d_tw, log_n, s, io);
      else if (kinds[i] == 2) hipLaunchKernelGGL((k_ntt_tile4<FR, 2>), dim3(blocks), dim3(256), lds_bytes, stream, d_work, d_tw, log_n, s, io);
      else {
        const uint32_t threads = n >> lv[i];
        if (lv[i] == 3) hipLaunchKernelGGL((k_ntt_pass<FR, 3>), dim3((threads + 255) / 256), dim3(256), 0, stream, d_work, d_tw, log_n, s, io);
        else if (lv[i] == 2) hipLaunchKernelGGL((k_ntt_pass<FR, 2>), dim3((threads + 255) / 256), dim3(256), 0, stream, d_work, d_tw, log_n, s, io);
        else hipLaunchKernelGGL((k_ntt_pass<FR, 1>), dim3((threads + 255) / 256), dim3(256), 0, stream, d_work, d_tw, log_n, s, io);
      }
      s -= lv[i];
    }
    NTT_HIP_OK(hipEventRecord(ev[2], stream));
    if (!fused) hipLaunchKernelGGL((k_ntt_store<FR>), dim3((n + 255) / 256), dim3(256), 0, stream, d_work, data_dev, log_n, (coset6 && coset_after) ? glo : nullptr, ghi,
                                   scale6 ? small(4) : nullptr);
    NTT_HIP_OK(hipEventRecord(ev[3], stream));
    NTT_HIP_OK(hipStreamSynchronize(stream));
    NTT_HIP_OK(hipGetLastError());
    (void)hipEventElapsedTime(&tm.load, ev[0], ev[1]);
    (void)hipEventElapsedTime(&tm.passes, ev[1], ev[2]);
    (void)hipEventElapsedTime(&tm.store, ev[2], ev[3]);
    (void)hipEventElapsedTime(&tm.total, ev[0], ev[3]);
    tm.npasses = np;
    return 0;
  }
  int run_host(uint64_t* data, unsigned log_n, const uint64_t* omega6, const uint64_t* coset6, int coset_after, const uint64_t* scale6,
               hipStream_t stream) {
    if (log_n > 28) return 2;
    const size_t bytes = ((size_t)FR::ARK64 * 8) << log_n;
    if (bytes > cap_io) {
      if (d_io) (void)hipFree(d_io);
      d_io = nullptr; cap_io = 0;
      NTT_HIP_OK(hipMalloc(&d_io, bytes));
      cap_io = bytes;
    }
    NTT_HIP_OK(hipMemcpyAsync(d_io, data, bytes, hipMemcpyHostToDevice, stream));
    if (int rc = run_device(d_io, log_n, omega6, coset6, coset_after, scale6, stream)) return rc;
    NTT_HIP_OK(hipMemcpy(data, d_io, bytes, hipMemcpyDeviceToHost));
    return 0;
  }

 private:
  uint32_t* d_work = nullptr; size_t cap_work = 0;
  uint32_t* d_tw = nullptr; size_t cap_tw = 0;
  uint32_t* d_small = nullptr;       // 5 x 1024 elements: twiddle lo/hi power tables, coset lo/hi power tables, [4][0] = scale
  uint64_t* d_io = nullptr; size_t cap_io = 0;
  unsigned tw_log_n = 0; uint64_t tw_omega[FR::ARK64] = {};
  OwnedStream stream_;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* small(int i) { return d_small + (size_t)i * 1024 * FR::WORDS; }

  // device-form copy of an arkworks-Montgomery element: done on the host with the same templates
  static void to_dev_words(const uint64_t* ark6, uint32_t* w16) {
    FR v = FR::wred(FR::from_ark(ark6));
    memset(w16, 0, FR::WORDS * 4);
    v.store(w16);
  }
  static FR host_pow1024(FR b) { for (int i = 0; i < 10; i++) b = FR::sqr(b); return FR::wred(b); }
  int upload_pair(const uint64_t* ark6, int slot_lo, int slot_hi, hipStream_t stream) {  // small(slot_lo)[i] = b^i, small(slot_hi)[i] = b^(1024 i)
    uint32_t h[2][FR::WORDS];
    to_dev_words(ark6, h[0]);
    FR b = FR::load(h[0]);
    memset(h[1], 0, sizeof h[1]);
    host_pow1024(b).store(h[1]);
    uint32_t* stage = small(4) + 4 * FR::WORDS;   // scratch slots behind the scale element
    NTT_HIP_OK(hipMemcpyAsync(stage, h, sizeof h, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL((k_ntt_powers<FR>), dim3(4), dim3(256), 0, stream, stage, small(slot_lo));
    hipLaunchKernelGGL((k_ntt_powers<FR>), dim3(4), dim3(256), 0, stream, stage + FR::WORDS, small(slot_hi));
    NTT_HIP_OK(hipStreamSynchronize(stream));     // h lives on this stack frame
    return 0;
  }
  int prepare(unsigned log_n, const uint64_t* omega6, const uint64_t* coset6, const uint64_t* scale6, hipStream_t stream) {
    if (!ev[0]) for (int i = 0; i < 4; i++) NTT_HIP_OK(hipEventCreate(&ev[i]));
    const size_t n = size_t(1) << log_n;
    if (!d_small) NTT_HIP_OK(hipMalloc(&d_small, (size_t)5 * 1024 * FR::WORDS * 4));
    if (n * FR::WORDS * 4 > cap_work) {
      if (d_work) (void)hipFree(d_work);
      d_work = nullptr; cap_work = 0;
      NTT_HIP_OK(hipMalloc(&d_work, n * FR::WORDS * 4));
      cap_work = n * FR::WORDS * 4;
    }
    const size_t tw_count = n > 1 ? n / 2 : 1;
    if (tw_count * FR::WORDS * 4 > cap_tw) {
      if (d_tw) (void)hipFree(d_tw);
      d_tw = nullptr; cap_tw = 0; tw_log_n = 0;
      NTT_HIP_OK(hipMalloc(&d_tw, tw_count * FR::WORDS * 4));
      cap_tw = tw_count * FR::WORDS * 4;
    }
    if (tw_log_n != log_n || memcmp(tw_omega, omega6, sizeof tw_omega) != 0) {
      if (upload_pair(omega6, 0, 1, stream)) return 1;
      hipLaunchKernelGGL((k_ntt_table<FR>), dim3(((uint32_t)tw_count + 255) / 256), dim3(256), 0, stream, small(0), small(1), d_tw, (uint32_t)tw_count);
      tw_log_n = log_n;
      memcpy(tw_omega, omega6, sizeof tw_omega);
    }
    if (coset6 && upload_pair(coset6, 2, 3, stream)) return 1;
    if (scale6) {
      uint32_t h[FR::WORDS];
      to_dev_words(scale6, h);
      NTT_HIP_OK(hipMemcpyAsync(small(4), h, sizeof h, hipMemcpyHostToDevice, stream));
      NTT_HIP_OK(hipStreamSynchronize(stream));
    }
    return 0;
  }
};
#endif  // __HIPCC__

}  // namespace celo
