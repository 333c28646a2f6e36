// The MSM host driver: tuning switches (MsmTuning), the engine (arena, streams, plan), run_device_windows / run_host_windows and the host epilogue.
// (part of the MSM pipeline: csrc/msm.h includes the pieces in order and carries the overview)
#pragma once

namespace celo {

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
  uint32_t host_head_split, host_tail_split; // ... how often the first / the last of them is cut in halves (celo_amd_msm_set_host_chunks)
  static const MsmTuning& get() {
    static const MsmTuning t = [] {
      MsmTuning v;
      // what a maintainer may want to switch (INTEGRATION.md lists them with the test that covers the non-default) ...
      v.use_glv = true;          // the endomorphism splits of the _subgroup entry points (CELO_NO_GLV / CELO_NO_GLS removed in round 6: the plain entry points ARE the unsplit form)
      v.use_gls = true;
      v.host_chunks = getenv("CELO_HOST_CHUNKS") ? (uint32_t)atoi(getenv("CELO_HOST_CHUNKS")) : 0xFFFFFFFFu;   // not set: the group's own default
      v.host_head_split = 0xFFFFFFFFu;     // the first / last chunk's halvings: celo_amd_msm_set_host_chunks carries them (the environment hooks are gone)
      v.host_tail_split = 0xFFFFFFFFu;
      v.batched_affine = getenv("CELO_BA") ? atoi(getenv("CELO_BA")) : 0;      // measured level with the XYZZ chain (profiles/r6_ba_ab.txt): off
      // ... and the decided A/Bs of rounds 2-6, constants now (their environment hooks - CELO_NO_NARROW, CELO_NO_LANE_BITSUM, CELO_FX_NO_COMPACT,
      // CELO_NO_HOST_THREADS, CELO_SEG_HALVES / _MIN / _MIN_SHARD, CELO_NO_SEG_OCC, CELO_SIDE_CONVERT, CELO_LANE_BITSUM_MAX[_SHARD], CELO_BA_LEVELS /
      // _OCC / _ROUNDS - were removed in round 6: nobody could review a library with 27 switches; the measurements are in DESIGN.md section 4 / 9)
      v.narrow_windows = true;
      v.gls_force = false;       // the library never applies psi to the plain entry points' arbitrary curve points
      v.lane_bitsum = true;
      v.fx_compact = true;
      v.host_threads = true;
      v.seg_halves = 0;          // the path's own default
      v.seg_min = 32u;           // shortest piece of a whole MSM
      v.seg_min_shard = 16u;     // ... of a window shard (8 / 12 / 16 measure alike)
      v.seg_occupancy = true;
      v.side_convert = false;    // base conversion beside the sort: measured level for window shards (round 4) and for whole MSMs (round 6: the time moves from convert to sort)
      v.side_convert_all = false;
      v.ba_levels = 3; v.ba_occ = 2; v.ba_rounds = 2u;    // profiles/r6_ba_ab.txt: the best of the variants measured
      v.bitsum_lanes_max_shard = 21 * 1024;
      v.bitsum_lanes_max = 21 * 1024;   // outputs of a launch: one wave of 21 additions per SIMD
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
  // bits instead of 29 x 13 (+ a carry window) 1.39 -> 1.88 ms, at 2^17 3.58 -> 3.79 ms.  (The round-2 A/B switch CELO_NO_NARROW is gone: MsmTuning::narrow_windows is a constant.)
  bool narrow_windows = MsmTuning::get().narrow_windows;
  bool narrow_top(int c) const { return narrow_windows && c == 16; }
  // big path, G1 of BLS12-377: the caller vouches for bases in the prime-order subgroup (Signature values, proving-key points): GLV split
  bool big_subgroup_points = false;
  // host-pointer entry (run_host), set by the Groth16 prover's entry points only: a base row x = 0, y = 1 is the identity (k_flag_ark_zero)
  bool ark_zero_identity = false;
  bool use_glv = MsmTuning::get().use_glv;     // (a constant since round 6; the pipelined host-pointer form still switches it off per call)
  bool last_glv = false;
  // window size for the 2 n points x 127-bit halves of the split (n = the expanded count)
  // (measured, round 3: 127 = 8 x 16 - 1, so c = 16 leaves no ragged top window and wins at every size from 2^14 terms up)
  static int window_bits_glv(size_t) { return 16; }
  // batched path, G2 of BLS12-377: the caller vouches that every base lies in the prime-order subgroup (Batch::verify's public keys:
  // PublicKey values only come from checked deserialisation, secret keys and sums of such), which is what makes psi(P) = [x]P
  bool gls_subgroup_points = false;
  bool use_gls = MsmTuning::get().use_gls;     // (a constant since round 6)
  bool gls_force = MsmTuning::get().gls_force;
  int last_gls_digits = 1;
  static constexpr int HOST_HORNER_THREADS = 4;
  bool host_threads = MsmTuning::get().host_threads;   // A/B switch (CELO_NO_HOST_THREADS) of the threaded host epilogue (Fq2 and BW6-761 groups)
  bool lane_horner = true;  // batched path: three lanes per instance in the Horner pass (tuning hook)
  bool lane_bitsum = MsmTuning::get().lane_bitsum;   // big path (a constant since round 6): three lanes per addition in the late levels of the bucket reduction (A/B hook)
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
    // window shards (round 4's A/B, MsmTuning::side_convert, a constant false since round 6 - measured level twice): the conversion of ALL n bases is replicated on every shard while its sort shrinks to a
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
    // chunk count by size: about 2^18 points per chunk within [4, 16] ([8, 16] BW6-761; 2^19 within [4, 8] for the Fq2 group) - the sweeps behind
    // these bounds: DESIGN.md section 4 "Host-pointer pipeline: chunk-count measurements"
    const bool fq2_group = sizeof(F) > 14 * sizeof(uint32_t) && G::SCALAR_BITS <= 256;
    uint32_t by_size = (uint32_t)(n >> (fq2_group ? 19 : 18));
    const uint32_t lo_k = G::SCALAR_BITS > 256 ? 8u : 4u, hi_k = fq2_group ? 8u : 16u;
    by_size = by_size < lo_k ? lo_k : by_size > hi_k ? hi_k : by_size;
    uint32_t chunks = ovr >= 0 ? (uint32_t)(ovr & 0xFF) : MsmTuning::get().host_chunks != 0xFFFFFFFFu ? MsmTuning::get().host_chunks : by_size;
    uint32_t head_split = ovr >= 0 && ((ovr >> 8) & 15) ? (uint32_t)((ovr >> 8) & 15) - 1u : MsmTuning::get().host_head_split != 0xFFFFFFFFu ? MsmTuning::get().host_head_split : HOST_HEAD_SPLIT_DEFAULT;
    uint32_t tail_split = ovr >= 0 && ((ovr >> 12) & 15) ? (uint32_t)((ovr >> 12) & 15) - 1u : MsmTuning::get().host_tail_split != 0xFFFFFFFFu ? MsmTuning::get().host_tail_split : HOST_TAIL_SPLIT_DEFAULT;
    if (chunks > 64) chunks = 64;
    // chunks of at least 2^17 points (2^16 when the count was set by hand - tests): a chunk's sort pass and the accumulation's last round are
    // fixed costs of ~0.08 ms (same DESIGN subsection)
    const size_t min_chunk_log = ovr >= 0 ? 16 : 17;
    if (chunks > (n >> min_chunk_log)) chunks = (uint32_t)(n >> min_chunk_log);
    // (the prover's four concurrent MSMs hide each other's transfers: pipelined only from 2^21 rows per query; the subgroup entry's GLV split
    // reads bases and scalars together and is not pipelined: from 2^19 terms a host-pointer call takes the pipelined PLAIN form instead - same
    // group element; both measured, same DESIGN subsection)
    const bool glv_plan = plan(n).glv;
    const bool glv_off = glv_plan && !ark_zero_identity && !win_cnt && n >= (size_t(1) << 19);
    // (the default head split only where its halves keep 2^17 points)
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
  OwnedStream side_stream_;            // the host-pointer pipeline's copy stream (and round 4's side conversion, off)
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
