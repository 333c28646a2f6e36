// MSM stage 5: folding multi-piece buckets (k_combine_mid / _big), the bit-sliced bucket reduction (k_bitsum, k_bitsum_lanes), results to arkworks form.
// (part of the MSM pipeline: csrc/msm.h includes the pieces in order and carries the overview)
#pragma once

namespace celo {

// ---- 5a. buckets cut into 2..16 pieces (e.g. every bucket of a short top window): one lane folds the pieces
// (`first` = 1, chunked pipeline: the first piece is the bucket's carrier and stays out of the fold - the pieces from the second on are
// folded into the second)
template <class G>
__global__ void __launch_bounds__(128) k_combine_mid(const uint32_t* __restrict__ mid, const uint32_t* __restrict__ nmid,
                                                     const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pfirst,
                                                     uint32_t* __restrict__ partials, uint32_t* __restrict__ pieces_of, uint32_t SEG, uint32_t first = 0) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < *nmid; q += gridDim.x * blockDim.x) {
    uint32_t t = mid[q];
    uint32_t pc = (counts[t] + SEG - 1) / SEG - first, pf = pfirst[t] + first;
    Xyzz<F> acc = IO::load_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS);
    for (uint32_t k = 1; k < pc; k++) {
      Xyzz<F> v = IO::load_xyzz(partials + (size_t)(pf + k) * IO::XYZZ_WORDS);
      xyzz_add_fn(acc, v);
    }
    IO::store_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS, acc);
    pieces_of[t] = 1;
  }
}
// the same fold with three lanes per bucket (curve_lanes.h): the folds are a handful of dependent additions on lone waves (the
// 4096 buckets of a short top window, four pieces each), i.e. latency - see k_bitsum_lanes
template <class G>
__global__ void __launch_bounds__(64) k_combine_mid_lanes(const uint32_t* __restrict__ mid, const uint32_t* __restrict__ nmid,
                                                          const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pfirst,
                                                          uint32_t* __restrict__ partials, uint32_t* __restrict__ pieces_of, uint32_t SEG, uint32_t first = 0) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  typedef QTriT<FieldBase<F>> QB;
  typedef LanePoint<QB> LP;
  const int g = QB::group();
  if (g >= 21) return;
  for (uint32_t q = blockIdx.x * 21u + (uint32_t)g; q < *nmid; q += gridDim.x * 21u) {
    const uint32_t t = mid[q];
    const uint32_t pc = (counts[t] + SEG - 1) / SEG - first, pf = pfirst[t] + first;
    const Xyzz<F> a = IO::load_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS);
    typename LP::Pt acc = {{a.X, a.Y, a.ZZ, a.ZZZ}, a.is_identity()};
    for (uint32_t k = 1; k < pc; k++) {
      const Xyzz<F> v = IO::load_xyzz(partials + (size_t)(pf + k) * IO::XYZZ_WORDS);
      const typename LP::P pb = {v.X, v.Y, v.ZZ, v.ZZZ};
      LP::add(acc, pb, v.is_identity());
    }
    if (QB::lane() == 0) {
      const Xyzz<F> r = acc.inf ? Xyzz<F>::identity() : Xyzz<F>{acc.p.X, acc.p.Y, acc.p.ZZ, acc.p.ZZZ};
      IO::store_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS, r);
      pieces_of[t] = 1;
    }
  }
}

// ---- 5b. buckets cut into many pieces (skewed inputs): one workgroup per such bucket folds its pieces into the first
template <class G>
__global__ void __launch_bounds__(256) k_combine_big(const uint32_t* __restrict__ big, const uint32_t* __restrict__ nbig,
                                                     const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pfirst,
                                                     uint32_t* __restrict__ partials, uint32_t* __restrict__ pieces_of, uint32_t SEG, uint32_t first = 0) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  __shared__ uint32_t stage[64 * IO::XYZZ_WORDS];
  for (uint32_t q = blockIdx.x; q < *nbig; q += gridDim.x) {
    uint32_t t = big[q];
    uint32_t pc = (counts[t] + SEG - 1) / SEG - first, pf = pfirst[t] + first;
    Xyzz<F> acc = Xyzz<F>::identity();
    for (uint32_t k = threadIdx.x; k < pc; k += 256) {
      Xyzz<F> v = IO::load_xyzz(partials + (size_t)(pf + k) * IO::XYZZ_WORDS);
      xyzz_add_fn(acc, v);
    }
    // fold 256 -> 64 -> 1 through LDS (64 slots)
    for (uint32_t width = 256; width > 1; width >>= 2) {
      uint32_t q4 = width >> 2;
      for (uint32_t r = 1; r < 4; r++) {
        __syncthreads();
        if (threadIdx.x >= r * q4 && threadIdx.x < (r + 1) * q4) IO::store_xyzz(stage + (threadIdx.x - r * q4) * IO::XYZZ_WORDS, acc);
        __syncthreads();
        if (threadIdx.x < q4) { Xyzz<F> v = IO::load_xyzz(stage + threadIdx.x * IO::XYZZ_WORDS); xyzz_add_fn(acc, v); }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { IO::store_xyzz(partials + (size_t)pf * IO::XYZZ_WORDS, acc); pieces_of[t] = 1; }
    __syncthreads();
  }
}

template <class G> HD Xyzz<typename G::F> load_bucket(const uint32_t* partials, const uint32_t* counts, const uint32_t* pfirst,
                                                       const uint32_t* pieces_of, uint32_t t, uint32_t SEG) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  if (!counts) return IO::load_xyzz(partials + (size_t)t * IO::XYZZ_WORDS);    // chunked pipeline: `partials` is the carrier table, one slot per bucket
  uint32_t c = counts[t];
  if (c == 0) return Xyzz<F>::identity();
  (void)pieces_of; (void)SEG;  // k_combine_mid / k_combine_big have folded multi-piece buckets into their first piece
  return IO::load_xyzz(partials + (size_t)pfirst[t] * IO::XYZZ_WORDS);
}

// ---- bucket reduction: window sum S = sum_b (b + 1) B_b with no scalar multiplication and depth log2(B).
// Binary tree over the bucket index: node(l, p) = sum of the buckets whose top l index bits are p (leaves at level LB).
// Bit k = LB - l of b is set exactly for the leaves under the odd-indexed nodes of level l, so
//   S = node(0, 0) + sum_{l=1..LB} 2^(LB - l) O_l,   O_l = sum_{p odd} node(l, p).
// Launch t builds level LB - t from level LB - t + 1 and halves every pending odd list once; a list is born strided
// (its first halving reads nodes 4i+1 and 4i+3 of its level).  LB launches, 2 point additions of work per bucket (the same
// as a running sum), every addition independent of the others of its launch: the depth of the whole reduction is LB
// additions instead of 16 (running sum) + ~18 (fix-up scalar) + log2 (tree).  The 2^(LB-l) weights are applied by the host
// inside the Horner recombination it runs anyway (one addition per doubling).
struct BitsumJobs {
  static constexpr int MAXJ = 20;
  uint32_t njobs;
  uint32_t end[MAXJ];    // cumulative number of outputs
  uint32_t src[MAXJ];    // point index into the work area (modes 0, 1, 4); unused for the leaf modes
  uint32_t dst[MAXJ];    // point index into the work area
  uint32_t mode[MAXJ];   // 0: in[2i] + in[2i+1]   1: in[4i+1] + in[4i+3]   2, 3: the same on the buckets themselves   4: in[2i+1]
};
template <class G>
__global__ void __launch_bounds__(128) k_bitsum(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ counts,
                                                const uint32_t* __restrict__ pfirst, const uint32_t* __restrict__ pieces_of, uint32_t SEG,
                                                uint32_t* __restrict__ work, BitsumJobs jobs) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= jobs.end[jobs.njobs - 1]) return;
  uint32_t j = 0;
  while (t >= jobs.end[j]) j++;
  const uint32_t i = t - (j ? jobs.end[j - 1] : 0u);
  const uint32_t mode = jobs.mode[j];
  Xyzz<F> a, b;
  if (mode == 2) {
    a = load_bucket<G>(partials, counts, pfirst, pieces_of, 2 * i, SEG);
    b = load_bucket<G>(partials, counts, pfirst, pieces_of, 2 * i + 1, SEG);
  } else if (mode == 3) {
    a = load_bucket<G>(partials, counts, pfirst, pieces_of, 4 * i + 1, SEG);
    b = load_bucket<G>(partials, counts, pfirst, pieces_of, 4 * i + 3, SEG);
  } else {
    const uint32_t* in = work + (size_t)jobs.src[j] * IO::XYZZ_WORDS;
    const size_t ia = mode == 0 ? 2 * (size_t)i : mode == 1 ? 4 * (size_t)i + 1 : 2 * (size_t)i + 1;
    a = IO::load_xyzz(in + ia * IO::XYZZ_WORDS);
    if (mode != 4) b = IO::load_xyzz(in + (ia + (mode == 0 ? 1 : 2)) * IO::XYZZ_WORDS);
  }
  if (mode != 4) xyzz_add(a, b);     // inlined for every field: each launch is one addition deep, its latency is the cost
  IO::store_xyzz(work + ((size_t)jobs.dst[j] + i) * IO::XYZZ_WORDS, a);
}
// The same launch with THREE LANES PER ADDITION (curve_lanes.h): the late levels of the reduction hold fewer additions than the
// chip has SIMDs, each launch costs the latency of one addition on a lone wave (~20 us for G1: 14 dependent-ish products), and
// spreading an addition's independent products over a lane group cuts that chain to 5 product rounds.  Used once a launch has
// few enough outputs that the tripled lane count still leaves every wave a SIMD of its own.
template <class G>
__global__ void __launch_bounds__(64) k_bitsum_lanes(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ counts,
                                                     const uint32_t* __restrict__ pfirst, const uint32_t* __restrict__ pieces_of, uint32_t SEG,
                                                     uint32_t* __restrict__ work, BitsumJobs jobs) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  typedef QTriT<FieldBase<F>> QB;
  typedef LanePoint<QB> LP;
  const int g = QB::group();
  const uint32_t t = blockIdx.x * 21u + (uint32_t)g;
  if (g >= 21 || t >= jobs.end[jobs.njobs - 1]) return;
  uint32_t j = 0;
  while (t >= jobs.end[j]) j++;
  const uint32_t i = t - (j ? jobs.end[j - 1] : 0u);
  const uint32_t mode = jobs.mode[j];
  Xyzz<F> a, b = Xyzz<F>::identity();
  if (mode == 2) {
    a = load_bucket<G>(partials, counts, pfirst, pieces_of, 2 * i, SEG);
    b = load_bucket<G>(partials, counts, pfirst, pieces_of, 2 * i + 1, SEG);
  } else if (mode == 3) {
    a = load_bucket<G>(partials, counts, pfirst, pieces_of, 4 * i + 1, SEG);
    b = load_bucket<G>(partials, counts, pfirst, pieces_of, 4 * i + 3, SEG);
  } else {
    const uint32_t* in = work + (size_t)jobs.src[j] * IO::XYZZ_WORDS;
    const size_t ia = mode == 0 ? 2 * (size_t)i : mode == 1 ? 4 * (size_t)i + 1 : 2 * (size_t)i + 1;
    a = IO::load_xyzz(in + ia * IO::XYZZ_WORDS);
    if (mode != 4) b = IO::load_xyzz(in + (ia + (mode == 0 ? 1 : 2)) * IO::XYZZ_WORDS);
  }
  typename LP::Pt acc = {{a.X, a.Y, a.ZZ, a.ZZZ}, a.is_identity()};
  if (mode != 4) {
    const typename LP::P pb = {b.X, b.Y, b.ZZ, b.ZZZ};
    LP::add(acc, pb, b.is_identity());
  }
  if (QB::lane() != 0) return;
  const Xyzz<F> r = acc.inf ? Xyzz<F>::identity() : Xyzz<F>{acc.p.X, acc.p.Y, acc.p.ZZ, acc.p.ZZZ};
  IO::store_xyzz(work + ((size_t)jobs.dst[j] + i) * IO::XYZZ_WORDS, r);
}
// the final results (node(0,0) and the O_l of every window), in place: device form -> arkworks limbs for the host's 64-bit
// Horner pass (host64.h).  Its own tiny launch: inside k_bitsum the conversion doubled the register count of every level.
template <class G>
__global__ void __launch_bounds__(64) k_results_to_ark(uint32_t* __restrict__ work, uint32_t res_pts) {
  // one lane per COORDINATE (the launch is a single conversion deep); a point's four lanes sit in one workgroup, and every load of
  // the workgroup precedes its stores: the arkworks form is shorter, so a coordinate's output overlaps its neighbour's input
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, pt = t >> 2, q = t & 3;
  const bool live = pt < res_pts;
  uint64_t out[IO::ARK64];
  if (live) F::load(work + (size_t)pt * IO::XYZZ_WORDS + q * IO::FW).to_ark(out);   // exact zeros (the identity's ZZ) stay exact zeros
  __syncthreads();
  if (live) {
    uint64_t* o = reinterpret_cast<uint64_t*>(work + (size_t)pt * IO::XYZZ_WORDS) + q * IO::ARK64;
#pragma unroll
    for (int i = 0; i < IO::ARK64; i++) o[i] = out[i];
  }
}

}  // namespace celo
