// Translation unit: bulk decoding of compressed BLS12-377 points (see wire.h), one point per lane.
#include <type_traits>
#include "wire.h"
#include <hip/hip_runtime.h>
#include <mutex>
#include "runtime.h"

// two waves per SIMD (scratch instead of AGPRs for what does not fit 256 VGPRs) unless overridden: -DFROW_OCC= for the A/B
#ifndef FROW_OCC
#define FROW_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
namespace celo {
// decode / normalise calls allocate their buffers per call and run on the caller's stream (or the null stream); calls from
// several host threads are serialised per process by this lock (they are bulk calls: one fills the GPU)
static std::mutex wire_mu;

// in: n x 48 (G1) / n x 96 (G2) wire bytes.  out: n x 12 / n x 24 u64, affine (x, y) in arkworks Montgomery limbs (the layout
// the MSM and pairing entry points take), zeros unless status == WIRE_OK.  Control flow is uniform apart from the table scans of the square
// root; the subgroup test (64-bit ladders by the curve parameter x, wire.h) uses the same scalar in every lane.
template <bool G2> __global__ void __launch_bounds__(64) FROW_OCC k_decompress(const uint8_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                      uint8_t* __restrict__ status, uint32_t n, int check, WireConsts k) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if constexpr (G2) {
    Affine<Fq2> p = {Fq2::zero(), Fq2::zero()};
    const WireStatus st = wire_decode_g2(in + (size_t)i * 96, k, check != 0, p);
    uint64_t* o = out + (size_t)i * 24;
    if (st == WIRE_OK) { p.x.c0.to_ark(o); p.x.c1.to_ark(o + 6); p.y.c0.to_ark(o + 12); p.y.c1.to_ark(o + 18); }
    else for (int j = 0; j < 24; j++) o[j] = 0;
    status[i] = st;
  } else {
    Affine<Fq> p = {Fq::zero(), Fq::zero()};
    const WireStatus st = wire_decode_g1(in + (size_t)i * 48, k, check != 0, p);
    uint64_t* o = out + (size_t)i * 12;
    if (st == WIRE_OK) { p.x.to_ark(o); p.y.to_ark(o + 6); }
    else for (int j = 0; j < 12; j++) o[j] = 0;
    status[i] = st;
  }
}

// Jacobian -> affine for n points (ProjectiveCurve::batch_normalization_into_affine, called at crates/bls-crypto/src/bls/
// signature.rs:82 and public.rs:58 before every MSM): Montgomery's trick inside a lane over K consecutive points - one field
// inversion per K points, 3 products per point for the prefix / suffix products, then x = X z^-2, y = Y z^-3.  in: n x 3
// coordinates (arkworks Montgomery limbs, identity = Z == 0); out: n x (x, y), zero rows + inf[i] = 1 for the identity.
// Points that are already affine (Z == 1: everything that came off the wire) take part with z = 1; their x, y come out unchanged.
template <int N, bool REVERSE, class Fn> __device__ __forceinline__ void norm_static_for(Fn&& f) {   // f(integral_constant<j>) for j = 0 .. N-1 (or N-1 .. 0)
  if constexpr (N > 0) {
    if constexpr (REVERSE) { f(std::integral_constant<int, N - 1>{}); norm_static_for<N - 1, true>(f); }
    else { norm_static_for<N - 1, false>(f); f(std::integral_constant<int, N - 1>{}); }
  }
}
template <class F, int K> __global__ void __launch_bounds__(64) FROW_OCC
k_normalize(const uint64_t* __restrict__ jac, uint64_t* __restrict__ out, uint8_t* __restrict__ inf, uint32_t n) {
  constexpr int A = F::ARK64;
  const uint32_t lo = (blockIdx.x * blockDim.x + threadIdx.x) * K;
  if (lo >= n) return;
  const uint32_t cnt = n - lo < (uint32_t)K ? n - lo : (uint32_t)K;
  // Only the prefix products stay in registers (K x 14 / 28 words); z_j is read again on the way back (one conversion product) instead
  // of being kept: with both arrays alive the G1 instance (K = 8) held 224 words of state, the loops were not unrolled and the arrays
  // went to private memory through run-time indices (912 B/lane).
  F pre[K];
  uint32_t idmask = 0;
  F acc = F::one();
  auto load_z = [&](int j) { return F::norm(F::from_ark(jac + ((size_t)lo + j) * 3 * A + 2 * A)); };
  // (unrolled by template recursion: `#pragma unroll` over bodies of this size is refused by the optimizer, and a rolled loop
  // indexes pre[] at run time, i.e. in private memory)
  norm_static_for<K, false>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    pre[j] = acc;
    if ((uint32_t)j < cnt) {
      const F zj = load_z(j);
      if (zj.is_zero_mod_p()) idmask |= 1u << j;
      else acc = F::norm(F::mul(acc, zj));
    }
  });
  F inv = F::norm(F::inv(acc));
  norm_static_for<K, true>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if ((uint32_t)j < cnt) {
      uint64_t* o = out + ((size_t)lo + j) * 2 * A;
      const bool id = (idmask >> j) & 1u;
      inf[lo + j] = id ? 1 : 0;
      if (id) {
        for (int q = 0; q < 2 * A; q++) o[q] = 0;
      } else {
        const F zi = F::norm(F::mul(inv, pre[j]));
        inv = F::norm(F::mul(inv, load_z(j)));
        const uint64_t* src = jac + ((size_t)lo + j) * 3 * A;
        const F zi2 = F::norm(F::sqr(zi));
        F::mul(F::from_ark(src), zi2).to_ark(o);
        F::mul(F::from_ark(src + A), F::norm(F::mul(zi2, zi))).to_ark(o + A);
      }
    }
  });
}

// the constants with the discrete-log tables in device memory: one copy per device, uploaded on first use there
int wire_consts_device(WireConsts& out) {
  static std::mutex mu;
  static WireTables* d_tabs[MAX_DEVICES] = {};
  std::lock_guard<std::mutex> lk(mu);
  WireTables*& d_tab = d_tabs[api_device()];
  out = wire_consts();
  if (!d_tab) {
    if (hipMalloc(&d_tab, sizeof(WireTables)) != hipSuccess) { d_tab = nullptr; return 10; }
    if (hipMemcpy(d_tab, out.tab, sizeof(WireTables), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d_tab); d_tab = nullptr; return 10; }
  }
  out.tab = d_tab;
  return 0;
}
static float g_wire_ms = 0.f;

#define WIRE_TRY(x)                                                                                  \
  do {                                                                                               \
    hipError_t e_ = (x);                                                                             \
    if (e_ != hipSuccess) { fprintf(stderr, "[celo-amd] %s: %s\n", #x, hipGetErrorString(e_)); rc = 10; goto done; } \
  } while (0)

int wire_decompress(int g2, const uint8_t* in, size_t n, int check, uint64_t* out, uint8_t* status, int dev, void* stream_) {
  if (int rc0 = api_enter()) return rc0;
  std::lock_guard<std::mutex> lk(wire_mu);
  if (n == 0) return 0;
  if (!in || !out || !status || n > 0x7fffffffu) return 2;
  WireConsts k;
  if (int rck = wire_consts_device(k)) return rck;
  const size_t ib = g2 ? 96 : 48, ow = g2 ? 24 : 12;
  hipStream_t stream = (hipStream_t)stream_;
  uint8_t *d_in = nullptr, *d_st = nullptr;
  uint64_t* d_out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  if (dev) { d_in = (uint8_t*)in; d_out = out; d_st = status; }
  else {
    WIRE_TRY(hipMalloc(&d_in, n * ib));
    WIRE_TRY(hipMalloc(&d_out, n * ow * 8));
    WIRE_TRY(hipMalloc(&d_st, n));
    WIRE_TRY(hipMemcpyAsync(d_in, in, n * ib, hipMemcpyHostToDevice, stream));
  }
  WIRE_TRY(hipEventCreate(&e0));
  WIRE_TRY(hipEventCreate(&e1));
  WIRE_TRY(hipEventRecord(e0, stream));
  if (g2) hipLaunchKernelGGL((k_decompress<true>), dim3(((uint32_t)n + 63) / 64), dim3(64), 0, stream, d_in, d_out, d_st, (uint32_t)n, check, k);
  else hipLaunchKernelGGL((k_decompress<false>), dim3(((uint32_t)n + 63) / 64), dim3(64), 0, stream, d_in, d_out, d_st, (uint32_t)n, check, k);
  WIRE_TRY(hipGetLastError());
  WIRE_TRY(hipEventRecord(e1, stream));
  if (!dev) {
    WIRE_TRY(hipMemcpyAsync(out, d_out, n * ow * 8, hipMemcpyDeviceToHost, stream));
    WIRE_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, stream));
  }
  WIRE_TRY(hipStreamSynchronize(stream));
  WIRE_TRY(hipEventElapsedTime(&g_wire_ms, e0, e1));
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (!dev) { if (d_in) (void)hipFree(d_in); if (d_out) (void)hipFree(d_out); if (d_st) (void)hipFree(d_st); }
  return rc;
}
int wire_normalize(int g2, const uint64_t* jac, size_t n, uint64_t* out_xy, uint8_t* inf) {
  if (int rc0 = api_enter()) return rc0;
  std::lock_guard<std::mutex> lk(wire_mu);
  if (n == 0) return 0;
  if (!jac || !out_xy || !inf || n > 0x7fffffffu) return 2;
  const size_t cw = g2 ? 12 : 6;
  uint64_t *d_in = nullptr, *d_out = nullptr;
  uint8_t* d_inf = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  WIRE_TRY(hipMalloc(&d_in, n * 3 * cw * 8));
  WIRE_TRY(hipMalloc(&d_out, n * 2 * cw * 8));
  WIRE_TRY(hipMalloc(&d_inf, n));
  WIRE_TRY(hipMemcpyAsync(d_in, jac, n * 3 * cw * 8, hipMemcpyHostToDevice, 0));
  WIRE_TRY(hipEventCreate(&e0));
  WIRE_TRY(hipEventCreate(&e1));
  WIRE_TRY(hipEventRecord(e0, 0));
  if (g2) hipLaunchKernelGGL((k_normalize<Fq2, 4>), dim3(((uint32_t)((n + 3) / 4) + 63) / 64), dim3(64), 0, 0, d_in, d_out, d_inf, (uint32_t)n);
  else hipLaunchKernelGGL((k_normalize<Fq, 8>), dim3(((uint32_t)((n + 7) / 8) + 63) / 64), dim3(64), 0, 0, d_in, d_out, d_inf, (uint32_t)n);
  WIRE_TRY(hipGetLastError());
  WIRE_TRY(hipEventRecord(e1, 0));
  WIRE_TRY(hipMemcpyAsync(out_xy, d_out, n * 2 * cw * 8, hipMemcpyDeviceToHost, 0));
  WIRE_TRY(hipMemcpyAsync(inf, d_inf, n, hipMemcpyDeviceToHost, 0));
  WIRE_TRY(hipStreamSynchronize(0));
  WIRE_TRY(hipEventElapsedTime(&g_wire_ms, e0, e1));
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (d_inf) (void)hipFree(d_inf);
  return rc;
}
float wire_last_ms() { return g_wire_ms; }
}  // namespace celo
