// Gather bandwidth of point-sized records ("Formula finding"): what a batched-affine pairwise tree over the SORTED
// order would see - records of REC bytes read through an index (a random permutation), a dense write of the same size - against the
// streaming rate of the same bytes.  usage: ubench_gather [log2 records = 24]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
template <int REC16> __global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ src, const uint32_t* __restrict__ idx, uint4* __restrict__ dst, uint32_t n) {
  // REC16 lanes per record: lane q of a record's group moves its q-th 16 bytes
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t r = t / REC16;
  const uint32_t q = (uint32_t)(t % REC16);
  if (r >= n) return;
  dst[r * REC16 + q] = src[(size_t)idx[r] * REC16 + q];
}
template <int REC16> float run(const uint4* src, const uint32_t* idx, uint4* dst, uint32_t n) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const unsigned blocks = (unsigned)(((size_t)n * REC16 + 255) / 256);
  hipLaunchKernelGGL((k_gather<REC16>), dim3(blocks), dim3(256), 0, 0, src, idx, dst, n);
  hipEventRecord(a, 0);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_gather<REC16>), dim3(blocks), dim3(256), 0, 0, src, idx, dst, n);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}
int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 24;
  const uint32_t n = 1u << lg;
  std::vector<uint32_t> ident(n), perm(n);
  std::iota(ident.begin(), ident.end(), 0u);
  perm = ident;
  std::mt19937_64 g(12345);
  std::shuffle(perm.begin(), perm.end(), g);
  uint4 *src, *dst; uint32_t *d_id, *d_pm;
  const size_t bytes = (size_t)n * 128;
  if (hipMalloc((void**)&src, bytes) != hipSuccess || hipMalloc((void**)&dst, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc((void**)&d_id, n * 4); hipMalloc((void**)&d_pm, n * 4);
  hipMemset(src, 1, bytes);
  hipMemcpy(d_id, ident.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_pm, perm.data(), n * 4, hipMemcpyHostToDevice);
  const float s128 = run<8>(src, d_id, dst, n), g128 = run<8>(src, d_pm, dst, n);
  const float s96 = run<6>(src, d_id, dst, n), g96 = run<6>(src, d_pm, dst, n);
  auto gbps = [&](int rec, float ms) { return 2.0 * n * rec / ms / 1e6; };     // read + write
  printf("{\"records\": %u, \"rec128_stream_GBps\": %.0f, \"rec128_gather_GBps\": %.0f, \"rec96_stream_GBps\": %.0f, \"rec96_gather_GBps\": %.0f, "
         "\"rec128_stream_ms\": %.3f, \"rec128_gather_ms\": %.3f}\n", n, gbps(128, s128), gbps(128, g128), gbps(96, s96), gbps(96, g96), s128, g128);
  return 0;
}
