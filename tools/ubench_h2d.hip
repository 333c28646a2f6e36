// Host -> device transfer rates the host-pointer MSM entry can count on (round 5): pageable hipMemcpy, pinned
// hipMemcpyAsync, the host's own memcpy into a pinned buffer on T threads, hipHostRegister of the caller's pages, and the staged
// pipeline (T workers copy sub-blocks into a pinned ring, the submitting thread sends each on as it completes).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/ubench_h2d tools/ubench_h2d.hip -lpthread && tools/ubench_h2d [MiB]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 128;
  const size_t bytes = mib << 20;
  char* pageable = (char*)malloc(bytes);
  for (size_t i = 0; i < bytes; i += 4096) pageable[i] = (char)i;
  char* pinned = nullptr; char* dev = nullptr;
  double t0 = now_ms();
  CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault));
  printf("hipHostMalloc %zu MiB: %.2f ms\n", mib, now_ms() - t0);
  memset(pinned, 1, bytes);
  CK(hipMalloc((void**)&dev, bytes));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  printf("hardware threads: %u\n", std::thread::hardware_concurrency());
  for (int r = 0; r < 4; r++) {
    t0 = now_ms(); CK(hipMemcpy(dev, pageable, bytes, hipMemcpyHostToDevice)); double t = now_ms() - t0;
    printf("pageable hipMemcpy        %8.3f ms  %6.1f GB/s\n", t, bytes / t / 1e6);
  }
  for (int r = 0; r < 4; r++) {
    t0 = now_ms(); CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t = now_ms() - t0;
    printf("pageable hipMemcpyAsync   %8.3f ms  %6.1f GB/s\n", t, bytes / t / 1e6);
  }
  for (int r = 0; r < 4; r++) {
    t0 = now_ms(); CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t = now_ms() - t0;
    printf("pinned hipMemcpyAsync     %8.3f ms  %6.1f GB/s\n", t, bytes / t / 1e6);
  }
  // a FRESH buffer per call (what a caller that builds new Vecs for every MSM hands over): allocation and first touch outside the timing
  for (int r = 0; r < 5; r++) {
    char* fresh = (char*)malloc(bytes);
    for (size_t i = 0; i < bytes; i += 4096) fresh[i] = (char)(i + r);
    t0 = now_ms(); CK(hipMemcpy(dev, fresh, bytes, hipMemcpyHostToDevice)); double t = now_ms() - t0;
    printf("FRESH pageable buffer hipMemcpy   %8.3f ms  %6.1f GB/s\n", t, bytes / t / 1e6);
    free(fresh);
  }
  for (int r = 0; r < 3; r++) {     // ... in four chunks on a stream, as the pipelined entry sends them
    char* fresh = (char*)malloc(bytes);
    for (size_t i = 0; i < bytes; i += 4096) fresh[i] = (char)(i + r);
    t0 = now_ms();
    for (int k = 0; k < 4; k++) CK(hipMemcpyAsync(dev + k * (bytes / 4), fresh + k * (bytes / 4), bytes / 4, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    double t = now_ms() - t0;
    printf("FRESH pageable buffer, 4 async chunks %8.3f ms  %6.1f GB/s\n", t, bytes / t / 1e6);
    free(fresh);
  }
  for (size_t kb : {64, 256, 1024, 4096, 16384}) {
    const size_t blk = kb << 10;
    t0 = now_ms();
    for (size_t o = 0; o < bytes; o += blk) CK(hipMemcpyAsync(dev + o, pinned + o, blk, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    double t = now_ms() - t0;
    printf("pinned async in %6zu KiB blocks  %8.3f ms  %6.1f GB/s\n", kb, t, bytes / t / 1e6);
  }
  for (int r = 0; r < 2; r++) {
    t0 = now_ms(); CK(hipHostRegister(pageable, bytes, hipHostRegisterDefault)); double t1 = now_ms() - t0;
    t0 = now_ms(); CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now_ms() - t0;
    t0 = now_ms(); CK(hipHostUnregister(pageable)); double t3 = now_ms() - t0;
    printf("hipHostRegister %.2f ms, copy %.3f ms (%.1f GB/s), unregister %.2f ms\n", t1, t2, bytes / t2 / 1e6, t3);
  }
  for (unsigned T : {1u, 2u, 4u, 8u, 16u, 32u, 64u}) {
    double best = 1e9;
    for (int r = 0; r < 3; r++) {
      t0 = now_ms();
      std::vector<std::thread> th;
      const size_t per = (bytes / T + 4095) & ~size_t(4095);
      for (unsigned i = 0; i < T; i++) th.emplace_back([=] { size_t o = i * per; if (o < bytes) memcpy(pinned + o, pageable + o, o + per <= bytes ? per : bytes - o); });
      for (auto& t : th) t.join();
      double t = now_ms() - t0; best = t < best ? t : best;
    }
    printf("memcpy pageable->pinned, %2u fresh threads: %8.3f ms  %6.1f GB/s\n", T, best, bytes / best / 1e6);
  }
  // staged pipeline: persistent workers claim sub-blocks, the submitting thread forwards finished blocks in order
  for (unsigned T : {4u, 8u, 16u, 32u}) for (size_t kb : {1024, 4096}) {
    const size_t blk = kb << 10, nblk = bytes / blk;
    double best = 1e9;
    for (int r = 0; r < 3; r++) {
      std::vector<std::atomic<int>> done(nblk);
      for (auto& d : done) d.store(0);
      std::atomic<size_t> next{0};
      t0 = now_ms();
      std::vector<std::thread> th;
      for (unsigned i = 0; i < T; i++) th.emplace_back([&] {
        for (;;) { size_t b = next.fetch_add(1); if (b >= nblk) break; memcpy(pinned + b * blk, pageable + b * blk, blk); done[b].store(1, std::memory_order_release); }
      });
      for (size_t b = 0; b < nblk; b++) {
        while (!done[b].load(std::memory_order_acquire)) {}
        CK(hipMemcpyAsync(dev + b * blk, pinned + b * blk, blk, hipMemcpyHostToDevice, s));
      }
      CK(hipStreamSynchronize(s));
      double t = now_ms() - t0; best = t < best ? t : best;
      for (auto& t2 : th) t2.join();
    }
    printf("staged pipeline %2u workers, %4zu KiB blocks: %8.3f ms  %6.1f GB/s\n", T, kb, best, bytes / best / 1e6);
  }
  return 0;
}
