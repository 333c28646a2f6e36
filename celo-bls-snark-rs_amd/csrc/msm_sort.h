// MSM stage 2-3: signed-digit recoding, the two-level counting sort of the (window, bucket) keys, the pieces of a bucket's run and their longest-first schedule.
// (part of the MSM pipeline: csrc/msm.h includes the pieces in order and carries the overview)
#pragma once

namespace celo {

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

}  // namespace celo
