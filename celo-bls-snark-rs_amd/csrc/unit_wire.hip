// Translation unit: bulk decoding of compressed BLS12-377 points (see wire.h), one point per lane.
#include "wire.h"
#include <hip/hip_runtime.h>
#include <mutex>

namespace celo {
std::mutex& api_mutex();
int api_ensure_init();

// in: n x 48 (G1) / n x 96 (G2) wire bytes.  out: n x 12 / n x 24 u64, affine (x, y) in arkworks Montgomery limbs (the layout
// the MSM and pairing entry points take), zeros unless status == WIRE_OK.  Divergence is confined to the Tonelli-Shanks order
// searches; the 253-step subgroup ladder that dominates is uniform (the scalar is r for every lane).
template <bool G2> __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_decompress(const uint8_t* __restrict__ in, uint64_t* __restrict__ out,
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

static float g_wire_ms = 0.f;

#define WIRE_TRY(x)                                                                                  \
  do {                                                                                               \
    hipError_t e_ = (x);                                                                             \
    if (e_ != hipSuccess) { fprintf(stderr, "[celo-amd] %s: %s\n", #x, hipGetErrorString(e_)); rc = 10; goto done; } \
  } while (0)

int wire_decompress(int g2, const uint8_t* in, size_t n, int check, uint64_t* out, uint8_t* status, int dev, void* stream_) {
  std::lock_guard<std::mutex> lk(api_mutex());
  if (int rc0 = api_ensure_init()) return rc0;
  if (n == 0) return 0;
  if (!in || !out || !status || n > 0x7fffffffu) return 2;
  const WireConsts& k = wire_consts();
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
float wire_last_ms() { return g_wire_ms; }
}  // namespace celo
