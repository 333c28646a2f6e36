// Process-wide runtime of the gfx950 library: device binding per host thread and pools of engine instances.
//
// The reference's callers are synchronous, re-entrant and multi-threaded (bls-snark-sys is called from Go/Rust worker threads:
// crates/bls-snark-sys/src/cache.rs:5, signatures.rs:343-400; epoch-snark's prover runs its MSMs from rayon tasks,
// crates/epoch-snark/src/api/prover.rs:78).  So there is no global API lock: every entry point
//   1. api_enter()          binds the calling thread to its device (HIP's current device is per thread),
//   2. EnginePool::lease()  checks out an engine instance of that device (workspace arena, pinned staging, events and its
//                           own non-blocking HIP stream); a new one is created when all are busy (up to MAX_PER_DEVICE,
//                           then callers wait),
//   3. runs, copies its timings to the per-kind "last call" record, and returns the engine.
// Independent calls therefore overlap on the GPU (separate streams) and several devices can be driven from one process:
// celo_amd_use_device() binds a host thread to a device, the msm_*_multi entry points do that internally (one thread per device).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>

namespace celo {

constexpr int MAX_DEVICES = 16;

int api_enter();                 // 0, or 100 (no device).  Applies the thread's device.
int api_device();                // device of the calling thread (after api_enter)
int api_bind_thread(int device); // celo_amd_use_device: 0 or 101

template <class E> class EnginePool {
 public:
  static constexpr int MAX_PER_DEVICE = 8;
  class Lease {
   public:
    Lease(EnginePool* p, int dev, E* e) : pool(p), device(dev), eng(e) {}
    Lease(Lease&& o) noexcept : pool(o.pool), device(o.device), eng(o.eng) { o.eng = nullptr; }
    Lease(const Lease&) = delete;
    ~Lease() { if (eng) pool->give_back(device, eng); }
    E* operator->() { return eng; }
    E& operator*() { return *eng; }
    explicit operator bool() const { return eng != nullptr; }
   private:
    EnginePool* pool; int device; E* eng;
  };
  // the calling thread must have passed api_enter()
  Lease lease() {
    const int dev = api_device();
    std::unique_lock<std::mutex> lk(mu);
    Slot& s = slots[dev];
    for (;;) {
      if (!s.free_list.empty()) { E* e = s.free_list.back(); s.free_list.pop_back(); return Lease(this, dev, e); }
      if (s.created < MAX_PER_DEVICE) {   // engines live for the process
        s.created++;
        lk.unlock();
        E* e = nullptr;
        try { e = new E(); } catch (...) {            // a failed construction must not cost the pool a slot for good
          { std::lock_guard<std::mutex> g(mu); slots[dev].created--; }
          cv.notify_one();
          throw;
        }
        return Lease(this, dev, e);
      }
      cv.wait(lk);
    }
  }
 private:
  struct Slot { std::vector<E*> free_list; int created = 0; };
  Slot slots[MAX_DEVICES];
  std::mutex mu;
  std::condition_variable cv;
  void give_back(int dev, E* e) {
    { std::lock_guard<std::mutex> lk(mu); slots[dev].free_list.push_back(e); }
    cv.notify_one();
  }
};

// celo_amd_msm_set_host_chunks (tests, bench.py's sweep): >= 0 overrides CELO_HOST_CHUNKS for the host-pointer MSM calls that follow
// (msm.h run_host_windows); -1 = the default
inline std::atomic<int>& host_chunks_override() { static std::atomic<int> v{-1}; return v; }
inline std::atomic<int>& batched_affine_override() { static std::atomic<int> v{-1}; return v; }   // -1: CELO_BA / the default (csrc/msm_ba.h)

// The chunks of the host-pointer pipeline: `chunks` index chunks of cm points (a multiple of 1024), the FIRST of them cut in halves
// `head_split` times, smallest piece first - the pipeline is bound by the GPU's work from the moment the first chunk has landed
// (accumulating a chunk takes a little longer than sending the next), so what the call pays beyond the resident pipeline is the
// first chunk's transfer: it is short.  On the device chunk k lives at the virtual index k cm (the staging buffers have holes behind
// short chunks): nothing below the transfers knows chunk lengths.  Returns the number of chunks (<= 80 = 64 + 8 + 8).
constexpr uint32_t HOST_CHUNKS_MAX = 80;
inline uint32_t host_chunk_plan(size_t n, uint32_t chunks, uint32_t head_split, uint32_t tail_split, uint32_t& cm, uint32_t* clen) {
  cm = (uint32_t)(((n + chunks - 1) / chunks + 1023u) & ~size_t(1023));
  const uint32_t base = (uint32_t)((n + cm - 1) / cm);
  uint32_t halves[8], nh = 0, piece = base >= 2 ? cm : (uint32_t)n;      // (one chunk: the whole job, cut by head_split only)
  for (uint32_t t = 0; t < head_split && t < 8 && piece >= (1u << 16); t++) {
    halves[nh] = ((piece + 1) / 2 + 1023u) & ~1023u;
    piece -= halves[nh++];
  }
  uint32_t K = 0;
  clen[K++] = piece;
  while (nh) clen[K++] = halves[--nh];
  if (base < 2) return K;
  for (uint32_t b = 1; b + 1 < base; b++) clen[K++] = cm;
  piece = (uint32_t)(n - (size_t)(base - 1) * cm);           // the last base chunk, largest piece first
  for (uint32_t t = 0; t < tail_split && t < 8 && piece >= (1u << 16); t++) {
    const uint32_t half = ((piece + 1) / 2 + 1023u) & ~1023u;
    clen[K++] = half;
    piece -= half;
  }
  clen[K++] = piece;
  return K;
}

// an engine-owned stream: non-blocking, so that engines of different calls do not serialise through the null stream
struct OwnedStream {
  hipStream_t s = nullptr;
  hipStream_t get() {
    if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
    return s;
  }
};


// ---- chaining engines on the device (batch verification: MSM results -> normalise -> pairing inputs, no host round trip)
// A batched MSM that has been ENQUEUED on its engine's stream: d_out = m Jacobian results (arkworks form) in the engine's
// arena; the engine stays leased until msm_batch_end_*.
// bits: in - the length of the longest scalar if the caller knows it (0: measure); out - what the run used
struct BatchRun { void* lease = nullptr; uint64_t* d_out = nullptr; hipStream_t stream = nullptr; int bits = 0; };
// A pairing engine whose input slots (k pairs in m products) are handed to a producer kernel; pairing_run_staged_* runs the
// check on what the slots hold (stream order) and returns the engine.
struct PairingStage { void* lease = nullptr; uint64_t* d_g1 = nullptr; uint64_t* d_g2 = nullptr; uint8_t* d_i1 = nullptr; uint8_t* d_i2 = nullptr; hipStream_t stream = nullptr; };

}  // namespace celo
