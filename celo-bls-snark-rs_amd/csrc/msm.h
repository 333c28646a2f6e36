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
//   2 k_digits<count>   signed c-bit digits per scalar; per-(window,bucket) histogram (global atomics)
//   3 k_scan            exclusive prefix sum per window (one workgroup per window, LDS + wave scans)
//   4 k_digits<scatter> counting-sort scatter of (point index | sign) into per-bucket runs
//   5 k_accumulate      one lane per bucket: gathers its run of points, XYZZ mixed adds
//   6 k_reduce_chunks   running-sum over CH consecutive buckets per lane -> (sum, weighted sum)
//   7 k_fixup / k_tree  weighted fix-up by small scalar, then 8-ary tree sums down to <=4 points/window
//   8 host              <=4*NW XYZZ points: Horner over windows (c doublings each) -> Jacobian, ark form
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "curve.h"
#include "fp2.h"

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

// signed-digit recoding; MODE 0 = histogram, MODE 1 = scatter
template <int SW, int CB, int NW, int MODE>
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf,
                                                uint32_t* __restrict__ counters, uint32_t* __restrict__ sorted, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (inf && inf[i]) return;
  uint32_t s[SW + 1];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * SW);
#pragma unroll
  for (int k = 0; k < SW / 4; k++) {
    uint4 v = sp[k];
    s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w;
  }
  s[SW] = 0;
  constexpr uint32_t B = 1u << (CB - 1);
  uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    constexpr int dummy = 0; (void)dummy;
    const int bit = w * CB;
    const int wi = bit >> 5, off = bit & 31;
    uint32_t raw = 0;
    if (wi < SW) {
      uint64_t two = ((uint64_t)s[wi + 1] << 32) | s[wi];
      raw = (uint32_t)(two >> off) & ((1u << CB) - 1);
    }
    uint32_t d = raw + carry;
    uint32_t neg = d > B ? 1u : 0u;
    uint32_t mag = neg ? ((1u << CB) - d) : d;
    carry = neg;
    if (mag != 0) {
      uint32_t slot = (uint32_t)w * B + (mag - 1);
      if (MODE == 0) {
        atomicAdd(&counters[slot], 1u);
      } else {
        uint32_t pos = atomicAdd(&counters[slot], 1u);
        sorted[(size_t)w * n + pos] = i | (neg << 31);
      }
    }
  }
}

// exclusive scan of B counters per window; one 1024-thread workgroup per window.
// counts[] -> starts[] (exclusive prefix) and cursors[] (= starts, consumed by the scatter pass)
template <class G>
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* __restrict__ counts, uint32_t* __restrict__ starts,
                                               uint32_t* __restrict__ cursors, uint32_t B) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t running;
  const uint32_t* c = counts + (size_t)blockIdx.x * B;
  uint32_t* st = starts + (size_t)blockIdx.x * B;
  uint32_t* cu = cursors + (size_t)blockIdx.x * B;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t base = 0; base < B; base += 1024) {
    uint32_t idx = base + threadIdx.x;
    uint32_t v = idx < B ? c[idx] : 0;
    uint32_t x = v;  // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    uint32_t pre = running;
    for (int k = 0; k < wv; k++) pre += wave_tot[k];
    uint32_t excl = pre + x - v;
    if (idx < B) { st[idx] = excl; cu[idx] = excl; }
    __syncthreads();
    if (threadIdx.x == 1023) running = pre + x;
    __syncthreads();
  }
}

// ---- longest-first bucket schedule.  Bucket sizes are far from uniform (the top window of a 253-bit scalar
// only has ~12 significant bits: 4096 buckets of n/4096 points; skewed inputs are worse), and a wave runs as long as
// its largest bucket.  A counting sort of the buckets by size (descending, sizes clamped to SIZE_BINS-1) makes the
// lanes of a wave near-equal and puts the big buckets first so no long tail is left at the end of the launch.
constexpr uint32_t SIZE_BINS = 2048;
template <class G>
__global__ void __launch_bounds__(256) k_size_hist(const uint32_t* __restrict__ counts, uint32_t* __restrict__ bins, uint32_t total) {
  __shared__ uint32_t lh[SIZE_BINS];
  for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 256) lh[i] = 0;
  __syncthreads();
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    uint32_t c = counts[t];
    uint32_t b = SIZE_BINS - 1 - (c < SIZE_BINS ? c : SIZE_BINS - 1);  // descending size
    atomicAdd(&lh[b], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 256)
    if (lh[i]) atomicAdd(&bins[i], lh[i]);
}
template <class G>
__global__ void __launch_bounds__(1024) k_size_scan(uint32_t* __restrict__ bins) {  // in-place exclusive scan of SIZE_BINS counters
  __shared__ uint32_t tmp[SIZE_BINS];
  for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 1024) tmp[i] = bins[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (uint32_t i = 0; i < SIZE_BINS; i++) { uint32_t v = tmp[i]; tmp[i] = run; run += v; }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < SIZE_BINS; i += 1024) bins[i] = tmp[i];
}
template <class G>
__global__ void __launch_bounds__(256) k_size_scatter(const uint32_t* __restrict__ counts, uint32_t* __restrict__ bins,
                                                      uint32_t* __restrict__ order, uint32_t total) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  uint32_t c = counts[t];
  uint32_t b = SIZE_BINS - 1 - (c < SIZE_BINS ? c : SIZE_BINS - 1);
  uint32_t pos = atomicAdd(&bins[b], 1u);
  order[pos] = t;
}

// one lane per bucket: XYZZ sum of its run of (signed) points
template <class G>
__global__ void __launch_bounds__(256) k_accumulate(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ starts, const uint32_t* __restrict__ ends,
                                                    const uint32_t* __restrict__ order, uint32_t* __restrict__ buckets,
                                                    uint32_t B, uint32_t total, uint32_t n) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= total) return;
  uint32_t t = order[tid];
  uint32_t w = t / B;
  const uint32_t* run = sorted + (size_t)w * n;
  uint32_t s = starts[t], e = ends[t];
  Xyzz<F> acc = Xyzz<F>::identity();
  for (uint32_t k = s; k < e; k++) {
    uint32_t v = run[k];
    Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
    if (v >> 31) p = affine_neg(p);
    xyzz_madd(acc, p);
  }
  IO::store_xyzz(buckets + (size_t)t * IO::XYZZ_WORDS, acc);
}

// running sums over CH consecutive buckets (descending): out[2*t] = sum_{j} (j - lo + 1) * B_j, out[2*t+1] = sum_j B_j
template <class G>
__global__ void __launch_bounds__(128) k_reduce_chunks(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ out,
                                                       uint32_t CH, uint32_t nchunks) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunks) return;
  Xyzz<F> running = Xyzz<F>::identity(), acc = Xyzz<F>::identity();
  const uint32_t* b = buckets + (size_t)t * CH * IO::XYZZ_WORDS;
  for (int j = (int)CH - 1; j >= 0; j--) {
    Xyzz<F> v = IO::load_xyzz(b + (size_t)j * IO::XYZZ_WORDS);
    xyzz_add(running, v);
    xyzz_add(acc, running);
  }
  IO::store_xyzz(out + (size_t)(2 * t) * IO::XYZZ_WORDS, acc);
  IO::store_xyzz(out + (size_t)(2 * t + 1) * IO::XYZZ_WORDS, running);
}

// contribution of chunk t (within its window): acc_t + (t_in_window * CH) * running_t
template <class G>
__global__ void __launch_bounds__(128) k_fixup(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t CH,
                                               uint32_t chunks_per_window, uint32_t nchunks) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunks) return;
  Xyzz<F> acc = IO::load_xyzz(in + (size_t)(2 * t) * IO::XYZZ_WORDS);
  Xyzz<F> run = IO::load_xyzz(in + (size_t)(2 * t + 1) * IO::XYZZ_WORDS);
  uint32_t k = (t % chunks_per_window) * CH;
  Xyzz<F> m = xyzz_mul_small(run, k);
  xyzz_add(acc, m);
  IO::store_xyzz(out + (size_t)t * IO::XYZZ_WORDS, acc);
}

// out[t] = sum of in[t*G .. t*G+G-1]
template <class G>
__global__ void __launch_bounds__(128) k_tree(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t grp, uint32_t nout) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nout) return;
  Xyzz<F> acc = IO::load_xyzz(in + (size_t)t * grp * IO::XYZZ_WORDS);
  for (uint32_t j = 1; j < grp; j++) {
    Xyzz<F> v = IO::load_xyzz(in + ((size_t)t * grp + j) * IO::XYZZ_WORDS);
    xyzz_add(acc, v);
  }
  IO::store_xyzz(out + (size_t)t * IO::XYZZ_WORDS, acc);
}

// ---------------------------------------------------------------- host driver
struct MsmTimings {  // milliseconds, HIP events on the MSM's stream (last call)
  float convert = 0, sort = 0, accumulate = 0, reduce = 0, total = 0;
};

template <class G> class MsmEngine {
 public:
  typedef typename G::F F;
  typedef PointIO<F> IO;
  static constexpr int SW = G::SCALAR_WORDS;

  ~MsmEngine() { release(); }
  void release() {
    for (void* p : {(void*)d_bases, (void*)d_sorted, (void*)d_counts, (void*)d_starts, (void*)d_cursors, (void*)d_buckets,
                    (void*)d_tmpA, (void*)d_tmpB, (void*)d_order, (void*)d_bins, (void*)d_in_bases, (void*)d_in_scalars, (void*)d_in_inf})
      if (p) (void)hipFree(p);
    d_bases = d_sorted = d_counts = d_starts = d_cursors = d_buckets = d_tmpA = d_tmpB = d_order = d_bins = nullptr;
    d_in_bases = nullptr; d_in_scalars = nullptr; d_in_inf = nullptr;
    if (h_out) { (void)hipHostFree(h_out); h_out = nullptr; }
    for (int i = 0; i < 6; i++) if (ev[i]) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
    cap_n = 0; cap_in = 0; cap_sorted = 0; cap_total = 0;
  }
  static int window_bits(size_t n) {
    int lg = 0;
    while ((size_t(1) << (lg + 1)) <= n) lg++;
    int c = lg - 4;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return c;
  }
  int force_c = 0;  // test hook / tuning: 0 = auto

  // bases/scalars/inf are DEVICE pointers (ark layout); result: Jacobian in ark Montgomery form (3*ARK64 u64) on host.
  int run_device(const uint64_t* d_ark_bases, const uint8_t* d_inf, const uint32_t* d_scalars, size_t n, uint64_t* out_jac,
                 hipStream_t stream) {
    if (n == 0) { write_identity(out_jac); return 0; }
    if (n >= (size_t(1) << 31)) return 2;
    int c = force_c ? force_c : window_bits(n);
    const int nw = (G::SCALAR_BITS + c) / c;
    const uint32_t B = 1u << (c - 1);
    const uint32_t total = (uint32_t)nw * B;
    if (ensure(n, total)) return 1;
    HIP_OK(hipEventRecord(ev[0], stream));
    hipLaunchKernelGGL((k_convert_bases<G>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_ark_bases, d_bases, n);
    HIP_OK(hipEventRecord(ev[1], stream));
    HIP_OK(hipMemsetAsync(d_counts, 0, (size_t)total * 4, stream));
    if (launch_digits(0, c, d_scalars, d_inf, d_counts, nullptr, (uint32_t)n, stream)) return 3;
    hipLaunchKernelGGL((k_scan<G>), dim3(nw), dim3(1024), 0, stream, d_counts, d_starts, d_cursors, B);
    if (launch_digits(1, c, d_scalars, d_inf, d_cursors, d_sorted, (uint32_t)n, stream)) return 3;
    HIP_OK(hipMemsetAsync(d_bins, 0, SIZE_BINS * 4, stream));
    hipLaunchKernelGGL((k_size_hist<G>), dim3(total / 256 < 512 ? (total + 255) / 256 : 512), dim3(256), 0, stream, d_counts, d_bins, total);
    hipLaunchKernelGGL((k_size_scan<G>), dim3(1), dim3(1024), 0, stream, d_bins);
    hipLaunchKernelGGL((k_size_scatter<G>), dim3((total + 255) / 256), dim3(256), 0, stream, d_counts, d_bins, d_order, total);
    HIP_OK(hipEventRecord(ev[2], stream));
    hipLaunchKernelGGL((k_accumulate<G>), dim3((total + 255) / 256), dim3(256), 0, stream, d_bases, d_sorted, d_starts, d_cursors,
                       d_order, d_buckets, B, total, (uint32_t)n);
    HIP_OK(hipEventRecord(ev[3], stream));
    // bucket reduction
    uint32_t CH = B >= 16 ? 16 : B;
    uint32_t cpw = B / CH, nchunks = cpw * nw;
    hipLaunchKernelGGL((k_reduce_chunks<G>), dim3((nchunks + 127) / 128), dim3(128), 0, stream, d_buckets, d_tmpA, CH, nchunks);
    hipLaunchKernelGGL((k_fixup<G>), dim3((nchunks + 127) / 128), dim3(128), 0, stream, d_tmpA, d_tmpB, CH, cpw, nchunks);
    uint32_t per_window = cpw;
    uint32_t* src = d_tmpB;
    uint32_t* dst = d_tmpA;
    while (per_window > 4) {
      uint32_t grp = per_window >= 8 ? 8 : per_window;
      while (per_window % grp) grp--;  // per_window is a power of two, so grp stays a power of two
      uint32_t nout = per_window / grp * nw;
      hipLaunchKernelGGL((k_tree<G>), dim3((nout + 127) / 128), dim3(128), 0, stream, src, dst, grp, nout);
      per_window /= grp;
      uint32_t* t = src; src = dst; dst = t;
    }
    HIP_OK(hipEventRecord(ev[4], stream));
    size_t out_words = (size_t)per_window * nw * IO::XYZZ_WORDS;
    HIP_OK(hipMemcpyAsync(h_out, src, out_words * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipEventRecord(ev[5], stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipGetLastError());
    (void)hipEventElapsedTime(&tm.convert, ev[0], ev[1]);
    (void)hipEventElapsedTime(&tm.sort, ev[1], ev[2]);
    (void)hipEventElapsedTime(&tm.accumulate, ev[2], ev[3]);
    (void)hipEventElapsedTime(&tm.reduce, ev[3], ev[4]);
    (void)hipEventElapsedTime(&tm.total, ev[0], ev[5]);
    last_c = c; last_nw = nw; last_buckets = total;
    // host epilogue: Horner over windows
    Xyzz<F> total_pt = Xyzz<F>::identity();
    for (int w = nw - 1; w >= 0; w--) {
      for (int k = 0; k < c; k++) total_pt = xyzz_dbl(total_pt);
      for (uint32_t j = 0; j < per_window; j++) {
        Xyzz<F> v = IO::load_xyzz(h_out + ((size_t)w * per_window + j) * IO::XYZZ_WORDS);
        xyzz_add(total_pt, v);
      }
    }
    write_jacobian(total_pt, out_jac);
    return 0;
  }

  // host-pointer entry: stages inputs into (cached) device buffers, then run_device
  int run_host(const uint64_t* bases, const uint8_t* inf, const uint64_t* scalars, size_t n, uint64_t* out_jac, hipStream_t stream) {
    if (n == 0) { write_identity(out_jac); return 0; }
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
    HIP_OK(hipMemcpyAsync(d_in_bases, bases, n * 2 * IO::ARK64 * 8, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_in_scalars, scalars, n * SW * 4, hipMemcpyHostToDevice, stream));
    if (inf) HIP_OK(hipMemcpyAsync(d_in_inf, inf, n, hipMemcpyHostToDevice, stream));
    return run_device(d_in_bases, inf ? d_in_inf : nullptr, (const uint32_t*)d_in_scalars, n, out_jac, stream);
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
  uint32_t *d_bases = nullptr, *d_sorted = nullptr, *d_counts = nullptr, *d_starts = nullptr, *d_cursors = nullptr;
  uint32_t *d_buckets = nullptr, *d_tmpA = nullptr, *d_tmpB = nullptr, *d_order = nullptr, *d_bins = nullptr;
  uint64_t* d_in_bases = nullptr;
  uint64_t* d_in_scalars = nullptr;
  uint8_t* d_in_inf = nullptr;
  uint32_t* h_out = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap_n = 0, cap_in = 0;
  uint32_t cap_total = 0;

  int ensure(size_t n, uint32_t total) {
    if (!ev[0])
      for (int i = 0; i < 6; i++) HIP_OK(hipEventCreate(&ev[i]));
    if (!h_out) HIP_OK(hipHostMalloc(&h_out, (size_t)4 * 128 * IO::XYZZ_WORDS * 4));
    if (!d_bins) HIP_OK(hipMalloc(&d_bins, SIZE_BINS * 4));
    int c = force_c ? force_c : window_bits(n);
    size_t nw = (G::SCALAR_BITS + c) / c;
    if (n > cap_n) {
      if (d_bases) (void)hipFree(d_bases);
      d_bases = nullptr; cap_n = 0;
      HIP_OK(hipMalloc(&d_bases, n * IO::AFF_WORDS * 4));
      cap_n = n;
    }
    if (n * nw > cap_sorted) {
      if (d_sorted) (void)hipFree(d_sorted);
      d_sorted = nullptr; cap_sorted = 0;
      HIP_OK(hipMalloc(&d_sorted, n * nw * 4));
      cap_sorted = n * nw;
    }
    if (total > cap_total) {
      for (void* p : {(void*)d_counts, (void*)d_starts, (void*)d_cursors, (void*)d_buckets, (void*)d_tmpA, (void*)d_tmpB, (void*)d_order})
        if (p) (void)hipFree(p);
      d_counts = d_starts = d_cursors = d_buckets = d_tmpA = d_tmpB = d_order = nullptr; cap_total = 0;
      HIP_OK(hipMalloc(&d_order, (size_t)total * 4));
      HIP_OK(hipMalloc(&d_counts, (size_t)total * 4));
      HIP_OK(hipMalloc(&d_starts, (size_t)total * 4));
      HIP_OK(hipMalloc(&d_cursors, (size_t)total * 4));
      HIP_OK(hipMalloc(&d_buckets, (size_t)total * IO::XYZZ_WORDS * 4));
      HIP_OK(hipMalloc(&d_tmpA, ((size_t)total / 4 + 1024) * IO::XYZZ_WORDS * 4));
      HIP_OK(hipMalloc(&d_tmpB, ((size_t)total / 4 + 1024) * IO::XYZZ_WORDS * 4));
      cap_total = total;
    }
    return 0;
  }
  size_t cap_sorted = 0;

  template <int CB> int launch_digits_c(int mode, const uint32_t* sc, const uint8_t* inf, uint32_t* ctr, uint32_t* sorted, uint32_t n,
                                        hipStream_t st) {
    constexpr int NW = (G::SCALAR_BITS + CB) / CB;
    dim3 g((n + 255) / 256), b(256);
    if (mode == 0) hipLaunchKernelGGL((k_digits<SW, CB, NW, 0>), g, b, 0, st, sc, inf, ctr, sorted, n);
    else hipLaunchKernelGGL((k_digits<SW, CB, NW, 1>), g, b, 0, st, sc, inf, ctr, sorted, n);
    return 0;
  }
  int launch_digits(int mode, int c, const uint32_t* sc, const uint8_t* inf, uint32_t* ctr, uint32_t* sorted, uint32_t n, hipStream_t st) {
    switch (c) {
      case 4: return launch_digits_c<4>(mode, sc, inf, ctr, sorted, n, st);
      case 5: return launch_digits_c<5>(mode, sc, inf, ctr, sorted, n, st);
      case 6: return launch_digits_c<6>(mode, sc, inf, ctr, sorted, n, st);
      case 7: return launch_digits_c<7>(mode, sc, inf, ctr, sorted, n, st);
      case 8: return launch_digits_c<8>(mode, sc, inf, ctr, sorted, n, st);
      case 9: return launch_digits_c<9>(mode, sc, inf, ctr, sorted, n, st);
      case 10: return launch_digits_c<10>(mode, sc, inf, ctr, sorted, n, st);
      case 11: return launch_digits_c<11>(mode, sc, inf, ctr, sorted, n, st);
      case 12: return launch_digits_c<12>(mode, sc, inf, ctr, sorted, n, st);
      case 13: return launch_digits_c<13>(mode, sc, inf, ctr, sorted, n, st);
      case 14: return launch_digits_c<14>(mode, sc, inf, ctr, sorted, n, st);
      case 15: return launch_digits_c<15>(mode, sc, inf, ctr, sorted, n, st);
      case 16: return launch_digits_c<16>(mode, sc, inf, ctr, sorted, n, st);
      default: return 1;
    }
  }
};

}  // namespace celo
