// Translation unit: batched hash-to-G1 over the direct hasher (see hash_direct.h): rounds of candidate counters, then the
// cofactor ladder.
#include "hash_direct.h"
#include "pedersen.h"
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include <cstring>
#include <cstdio>

#include "runtime.h"
// two waves per SIMD (scratch instead of AGPRs for what does not fit 256 VGPRs) unless overridden: -DFROW_OCC= for the A/B
#ifndef FROW_OCC
#define FROW_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
namespace celo {
static std::mutex hash_mu;                 // bulk calls, serialised per process (each fills the GPU; buffers are per call)
int wire_consts_device(WireConsts& out);   // unit_wire.hip

struct HashDom { uint8_t b[8]; };
struct HashIn { const uint8_t* msgs; const uint64_t* msg_off; const uint8_t* extras; const uint64_t* extra_off; };

// One round of try-and-increment for the messages still without a point.  list: their indices (nullptr = all of 0..count);
// each gets CAND adjacent lanes trying the counters base .. base + CAND - 1 side by side (CAND = 1, 2, ... 64 divides the wave:
// many messages -> 1, so every lane's square root is one the serial loop would also have computed; few messages -> up to 16,
// so a round finds a point with probability 1 - 0.58^16 and the call is two launches deep instead of eight).  The lowest
// successful counter of the group wins (ballot), its lane stores the curve point BEFORE the cofactor; a group without success
// appends its message to the next round's list.
__global__ void __launch_bounds__(64) FROW_OCC
k_hash_candidates(HashDom dom, HashIn in, const uint32_t* __restrict__ list, uint32_t count, uint32_t cand_log, uint32_t base, int mode, const EdPoint* __restrict__ gens,
                  uint64_t* __restrict__ cand_xy, uint8_t* __restrict__ attempts, uint32_t* __restrict__ next_list, uint32_t* __restrict__ next_count,
                  WireConsts k) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j = t >> cand_log, s = t & ((1u << cand_log) - 1);
  const bool live = j < count;
  const uint32_t i = live ? (list ? list[j] : j) : 0;
  const uint32_t c = base + s;
  bool ok = false;
  Affine<Fq> p = {Fq::zero(), Fq::zero()};
  if (live && c < 255) {
    const uint8_t* msg = in.msgs + in.msg_off[i];
    const size_t mlen = (size_t)(in.msg_off[i + 1] - in.msg_off[i]);
    const uint8_t* extra = in.extra_off ? in.extras + in.extra_off[i] : nullptr;
    const size_t elen = in.extra_off ? (size_t)(in.extra_off[i + 1] - in.extra_off[i]) : 0;
    ok = tai_candidate(dom.b, msg, mlen, extra, elen, (int)c, k, p, mode, gens);
  }
  const uint64_t mask = __ballot(ok);
  const uint32_t lane = threadIdx.x & 63, g0 = lane & ~((1u << cand_log) - 1);
  const uint64_t gm = (mask >> g0) & (cand_log == 6 ? ~0ull : ((1ull << (1u << cand_log)) - 1));
  if (!live) return;
  if (gm) {
    const uint32_t win = (uint32_t)__builtin_ctzll(gm);
    if (s == win) {
      uint64_t* o = cand_xy + (size_t)i * 12;
      p.x.to_ark(o);
      p.y.to_ark(o + 6);
      attempts[i] = (uint8_t)c;
    }
  } else if (s == 0) {
    if (base + (1u << cand_log) >= 255) attempts[i] = 255;                // every counter tried: the reference errs
    else next_list[atomicAdd(next_count, 1u)] = i;
  }
}
// scale_by_cofactor + normalisation of the winning candidates, one message per lane (uniform: a 124-step ladder and one
// inversion).  A multiple that is the identity (probability ~2^-125 per message) is flagged for the host's serial loop.
__global__ void __launch_bounds__(64) FROW_OCC
k_hash_finish(const uint64_t* __restrict__ cand_xy, const uint8_t* __restrict__ attempts, uint64_t* __restrict__ out, uint8_t* __restrict__ redo, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t* o = out + (size_t)i * 12;
  redo[i] = 0;
  if (attempts[i] == 255) { for (int j = 0; j < 12; j++) o[j] = 0; return; }
  const uint64_t* ci = cand_xy + (size_t)i * 12;
  const Affine<Fq> p = {Fq::norm(Fq::from_ark(ci)), Fq::norm(Fq::from_ark(ci + 6))};
  Affine<Fq> r = {Fq::zero(), Fq::zero()};
  if (tai_finish(p, r)) { r.x.to_ark(o); r.y.to_ark(o + 6); }
  else { redo[i] = 1; for (int j = 0; j < 12; j++) o[j] = 0; }
}

const EdPoint* celo_composite_gens(size_t* count);   // seam_a.hip: the generator table (built once, ChaCha20 stream of the reference)
static EdPoint* g_d_gens_dev[MAX_DEVICES] = {};      // its device copies (11.7 MB each), uploaded on first use (under hash_mu)
#define g_d_gens g_d_gens_dev[api_device()]
static int ensure_device_gens(const EdPoint* h_gens, size_t ngens) {
  if (g_d_gens) return 0;
  if (hipMalloc(&g_d_gens, ngens * sizeof(EdPoint)) != hipSuccess) { g_d_gens = nullptr; return 10; }
  if (hipMemcpy(g_d_gens, h_gens, ngens * sizeof(EdPoint), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(g_d_gens); g_d_gens = nullptr; return 10; }
  return 0;
}
// Device scratch and stream of the bulk hashers, per device, grow-only (under hash_mu).  Round 4: the calls used to hipMalloc / hipFree
// eight buffers and run on the null stream - inside batch_verify_strict, whose hashing thread runs BESIDE the two batch MSMs, the null
// stream's implicit joins and hipFree's device-wide wait held the hashes back until the MSMs had drained (hashes joined at 23.5 of
// 29.9 ms) and the pairing leg was enqueued only then.  A non-blocking stream of their own and one cached allocation: they finish
// under the G2 accumulation.
struct HashScratch { hipStream_t stream = nullptr; uint8_t* buf = nullptr; size_t cap = 0; };
static HashScratch g_hash_scratch[MAX_DEVICES];
static int hash_scratch(size_t need, HashScratch** out) {
  HashScratch& h = g_hash_scratch[api_device()];
  if (!h.stream && hipStreamCreateWithFlags(&h.stream, hipStreamNonBlocking) != hipSuccess) { h.stream = nullptr; return 10; }
  if (need > h.cap) {
    if (h.buf) (void)hipFree(h.buf);
    h.buf = nullptr; h.cap = 0;
    if (hipMalloc((void**)&h.buf, need + need / 4) != hipSuccess) { h.buf = nullptr; return 10; }
    h.cap = need + need / 4;
  }
  *out = &h;
  return 0;
}
static float g_hash_ms = 0.f;
static int g_hash_rounds = 0;

#define HASH_TRY(x)                                                                                  \
  do {                                                                                               \
    hipError_t e_ = (x);                                                                             \
    if (e_ != hipSuccess) { fprintf(stderr, "[celo-amd] %s: %s\n", #x, hipGetErrorString(e_)); rc = 10; goto done; } \
  } while (0)

int hash_to_g1_direct_run(const uint8_t* domain, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* extras, const uint64_t* extra_off,
                          size_t n, uint64_t* out_xy, uint8_t* attempts, int mode) {
  size_t ngens = 0;
  const EdPoint* h_gens = mode == TAI_COMPOSITE ? celo_composite_gens(&ngens) : nullptr;   // before the lock: 0.5 s on first use
  if (int rc0 = api_enter()) return rc0;
  std::lock_guard<std::mutex> lk(hash_mu);
  if (n == 0) return 0;
  if (mode == TAI_COMPOSITE) {
    if (int rcg = ensure_device_gens(h_gens, ngens)) return rcg;
    for (size_t i = 0; i < n; i++)   // counter || extra || message must fit the generator table (the reference panics)
      if ((1 + (msg_off ? msg_off[i + 1] - msg_off[i] : 0) + (extra_off ? extra_off[i + 1] - extra_off[i] : 0)) * 8 > PEDERSEN_MAX_BITS) return 2;
  }
  const EdPoint* d_gens = mode == TAI_COMPOSITE ? g_d_gens : nullptr;
  if (!domain || !msg_off || !out_xy || !attempts || n > 0x3fffffffu) return 2;
  for (size_t i = 0; i < n; i++) {
    if (msg_off[i + 1] < msg_off[i] || (extra_off && extra_off[i + 1] < extra_off[i])) return 2;
  }
  const size_t mb = msg_off[n], eb = extra_off ? extra_off[n] : 0;
  if ((mb && !msgs) || (eb && !extras)) return 2;
  const WireConsts& kh = wire_consts();   // host tables (the serial fallback below)
  WireConsts k;
  if (int rck = wire_consts_device(k)) return rck;
  HashDom dom;
  memcpy(dom.b, domain, 8);
  uint8_t *d_bytes = nullptr, *d_att = nullptr, *d_redo = nullptr;
  uint64_t *d_off = nullptr, *d_out = nullptr, *d_cand = nullptr;
  uint32_t *d_list = nullptr, *d_cnt = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<uint8_t> redo(n);
  int rc = 0;
  HashScratch* hs = nullptr;
  hipStream_t st = nullptr;
  {
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    const size_t o_off = take((n + 1) * 2 * 8), o_out = take(n * 12 * 8), o_cand = take(n * 12 * 8), o_list = take(2 * n * 4), o_cnt = take(256 * 4),
                 o_att = take(n), o_redo = take(n), o_bytes = take(mb + eb + 8);
    if (hash_scratch(off, &hs)) { rc = 10; goto done; }
    st = hs->stream;
    d_off = (uint64_t*)(hs->buf + o_off); d_out = (uint64_t*)(hs->buf + o_out); d_cand = (uint64_t*)(hs->buf + o_cand);
    d_list = (uint32_t*)(hs->buf + o_list); d_cnt = (uint32_t*)(hs->buf + o_cnt); d_att = hs->buf + o_att; d_redo = hs->buf + o_redo; d_bytes = hs->buf + o_bytes;
    if (mb) HASH_TRY(hipMemcpyAsync(d_bytes, msgs, mb, hipMemcpyHostToDevice, st));
    if (eb) HASH_TRY(hipMemcpyAsync(d_bytes + mb, extras, eb, hipMemcpyHostToDevice, st));
    HASH_TRY(hipMemcpyAsync(d_off, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
    if (extra_off) HASH_TRY(hipMemcpyAsync(d_off + n + 1, extra_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
    HASH_TRY(hipMemsetAsync(d_cnt, 0, 256 * 4, st));
    HASH_TRY(hipEventCreate(&e0));
    HASH_TRY(hipEventCreate(&e1));
    HASH_TRY(hipEventRecord(e0, st));
    const HashIn in = {d_bytes, d_off, d_bytes + mb, extra_off ? d_off + n + 1 : nullptr};
    uint32_t count = (uint32_t)n, base = 0;
    int round = 0;
    while (count && base < 255) {
      uint32_t cand_log = 0;
      while (cand_log < 4 && ((size_t)count << (cand_log + 1)) <= (1u << 17)) cand_log++;     // fill ~2^17 lanes, at most 16 counters
      const uint32_t* list = round ? d_list + (size_t)(round & 1) * n : nullptr;
      uint32_t* next = d_list + (size_t)((round + 1) & 1) * n;
      const size_t lanes = (size_t)count << cand_log;
      hipLaunchKernelGGL(k_hash_candidates, dim3((uint32_t)((lanes + 63) / 64)), dim3(64), 0, st, dom, in, list, count, cand_log, base, mode, d_gens, d_cand, d_att,
                         next, d_cnt + round, k);
      HASH_TRY(hipGetLastError());
      HASH_TRY(hipMemcpyAsync(&count, d_cnt + round, 4, hipMemcpyDeviceToHost, st));
      HASH_TRY(hipStreamSynchronize(st));
      base += 1u << cand_log;
      round++;
    }
    g_hash_rounds = round;
    hipLaunchKernelGGL(k_hash_finish, dim3(((uint32_t)n + 63) / 64), dim3(64), 0, st, d_cand, d_att, d_out, d_redo, (uint32_t)n);
    HASH_TRY(hipGetLastError());
    HASH_TRY(hipEventRecord(e1, st));
    HASH_TRY(hipMemcpyAsync(out_xy, d_out, n * 12 * 8, hipMemcpyDeviceToHost, st));
    HASH_TRY(hipMemcpyAsync(attempts, d_att, n, hipMemcpyDeviceToHost, st));
    HASH_TRY(hipMemcpyAsync(redo.data(), d_redo, n, hipMemcpyDeviceToHost, st));
    HASH_TRY(hipStreamSynchronize(st));
    HASH_TRY(hipEventElapsedTime(&g_hash_ms, e0, e1));
    // the counter whose cofactor multiple was the identity is skipped like the reference's loop does: continue serially after it
    for (size_t i = 0; i < n; i++) {
      if (!redo[i]) continue;
      Affine<Fq> p = {Fq::zero(), Fq::zero()};
      int c = 255;
      uint64_t* o = out_xy + i * 12;
      if (hash_to_g1_direct_tai(domain, msgs + msg_off[i], msg_off[i + 1] - msg_off[i], extra_off ? extras + extra_off[i] : nullptr,
                                extra_off ? extra_off[i + 1] - extra_off[i] : 0, kh, p, c, attempts[i] + 1, mode, h_gens)) { p.x.to_ark(o); p.y.to_ark(o + 6); attempts[i] = (uint8_t)c; }
      else { attempts[i] = 255; memset(o, 0, 96); }
    }
  }
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return rc;
}
// ---- bulk Pedersen CRH (pedersen.h): one message per lane; the 52080-generator table (11.7 MB) is uploaded on first use
__global__ void __launch_bounds__(64) FROW_OCC
k_pedersen_crh(const EdPoint* __restrict__ gens, const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ off, uint8_t* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t h[48];
  pedersen_crh(gens, msgs + off[i], (size_t)(off[i + 1] - off[i]), h);
  for (int j = 0; j < 48; j++) out[(size_t)i * 48 + j] = h[j];
}

int pedersen_crh_run(const uint8_t* msgs, const uint64_t* msg_off, size_t n, uint8_t* out48) {
  size_t ngens = 0;
  const EdPoint* gens = celo_composite_gens(&ngens);      // before the lock: building the table takes 0.5 s on first use
  if (int rc0 = api_enter()) return rc0;
  std::lock_guard<std::mutex> lk(hash_mu);
  if (n == 0) return 0;
  if (!msg_off || !out48 || n > 0x7fffffffu) return 2;
  for (size_t i = 0; i < n; i++) {
    if (msg_off[i + 1] < msg_off[i] || (msg_off[i + 1] - msg_off[i]) * 8 > PEDERSEN_MAX_BITS) return 2;   // the reference panics on longer messages
  }
  const size_t mb = msg_off[n];
  if (mb && !msgs) return 2;
  uint8_t *d_bytes = nullptr, *d_out = nullptr;
  uint64_t* d_off = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  if (int rcg = ensure_device_gens(gens, ngens)) return rcg;
  EdPoint* d_gens = g_d_gens;
  HashScratch* hs = nullptr;
  hipStream_t st = nullptr;
  {
    const size_t o_off = 0, o_out = ((n + 1) * 8 + 255) & ~size_t(255), o_bytes = o_out + ((n * 48 + 255) & ~size_t(255));
    if (hash_scratch(o_bytes + mb + 8, &hs)) { rc = 10; goto done; }
    st = hs->stream;
    d_off = (uint64_t*)(hs->buf + o_off); d_out = hs->buf + o_out; d_bytes = hs->buf + o_bytes;
  }
  if (mb) HASH_TRY(hipMemcpyAsync(d_bytes, msgs, mb, hipMemcpyHostToDevice, st));
  HASH_TRY(hipMemcpyAsync(d_off, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  HASH_TRY(hipEventCreate(&e0));
  HASH_TRY(hipEventCreate(&e1));
  HASH_TRY(hipEventRecord(e0, st));
  hipLaunchKernelGGL(k_pedersen_crh, dim3(((uint32_t)n + 63) / 64), dim3(64), 0, st, d_gens, d_bytes, d_off, d_out, (uint32_t)n);
  HASH_TRY(hipGetLastError());
  HASH_TRY(hipEventRecord(e1, st));
  HASH_TRY(hipMemcpyAsync(out48, d_out, n * 48, hipMemcpyDeviceToHost, st));
  HASH_TRY(hipStreamSynchronize(st));
  HASH_TRY(hipEventElapsedTime(&g_hash_ms, e0, e1));
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return rc;
}
float hash_last_ms() { return g_hash_ms; }
int hash_last_rounds() { return g_hash_rounds; }
}  // namespace celo
