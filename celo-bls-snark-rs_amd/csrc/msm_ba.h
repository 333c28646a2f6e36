// Batched-affine pre-levels of the bucket accumulation (round 6; DESIGN.md section 4 "Batched-affine accumulation for the wide fields").
//
// k_accumulate (msm.h) walks a piece's run of points with ONE XYZZ accumulator: a mixed addition per point, 8 products + 2 squarings
// (14 140 multiply-adds over the 28-limb field, at one wave per SIMD with 246 AGPRs of parked state).  An AFFINE addition is
// lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1: 2 products + 1 squaring once 1 / (x2 - x1) is known, and
// Montgomery's trick shares one inversion among B denominators for 3 more products each: 5 products + 1 squaring + 1/B of an inversion
// (~65 products' worth of division steps, csrc/modinv.h) = 9 030 multiply-adds + 100 000 / B.  The trick needs B INDEPENDENT additions per
// inversion - a lane's chain of additions into one accumulator is not - so the run is reduced as a pairwise TREE: level 0 adds the points of
// a run two by two (the leftover of an odd run is copied), level 1 the results, ...  Here every lane owns a stride of pieces of the longest-first
// schedule (a long one, a medium one, ... a short one: the same work in every lane), a few hundred pairs at level 0, and does its own inversion (the division steps are one instruction
// stream whatever the operand): no cross-lane traffic, no barrier, one launch for all levels.  After K levels a piece is 2^-K as long
// and k_accumulate_ba finishes it with the XYZZ chain (7/8 of the additions are gone at K = 3).
//
// Memory: the level results live IN PLACE in one array of affine points indexed like `sorted` (a piece owns the positions of its run):
// level l reads positions cur .. cur + L - 1 and writes its M = ceil(L / 2) results RIGHT-aligned, cur + L - M + j - the back sweep of
// Montgomery's trick runs j = M - 1 .. 0 and position cur + L - M + j >= cur + 2 j is never an input still to be read.  The prefix
// products go to a lane-private column of `pref` ([slot][lane], one 16-byte-padded field element each).
//
// Special pairs (never met on random inputs; the -m gpu suite's repeated-base / negated-pair / witness-like cases meet all of them): equal
// points double (denominator 2 y), opposite points give the identity, the identity as an operand (a level result only: k_digits drops
// infinite bases) is skipped.  The identity is stored as x = y = 0 - not a point of y^2 = x^3 + b, b != 0.  Such pairs put 1 into the batch.
//
// Replaces, for the groups that enable it (BaCfg), the k_accumulate launch of run_device_windows' resident path.  Same group elements, so
// the MSM's affine result is bit-identical (tests/test_msm_gpu.py, test_configs_gpu.py compare with the oracle either way).
#pragma once

namespace celo {

constexpr int BA_K_MAX = 4;    // tree levels before the XYZZ chain: a launch parameter (MsmTuning::ba_levels), at most this
constexpr uint32_t BA_LANES_OCC1 = 1024u * 64u;      // lanes in flight with one wave per SIMD; the grid is ONE round of them (x occupancy)

template <class G> struct BaCfg { static constexpr bool enabled = false; };

// the piece's position and length after `level` levels: (cur, L) -> (cur + L - ceil(L / 2), ceil(L / 2))
HD inline void ba_piece_at(uint32_t s, uint32_t len, int level, uint32_t& cur, uint32_t& L) {
  cur = s; L = len;
  for (int l = 0; l < level; l++) { const uint32_t M = (L + 1) / 2; cur += L - M; L = M; }
}

#define BA_FENCE() asm volatile("" ::: "memory")
template <class F> struct BaOps {
  typedef PointIO<F> IO;
  // kinds of a pair
  enum { ADD = 0, DBL = 1, TAKE_P = 2, TAKE_Q = 3, INF = 4 };
  HD static bool is_inf(const F& x, const F& y) { return x.limbs_all_zero() && y.limbs_all_zero(); }
  // denominator of the pair's slope (what the batch inverts) and the pair's kind.  Generic pairs never read the y coordinates here.
  template <class LoadY1, class LoadY2> HD static int classify(const F& x1, const F& x2, LoadY1 y1f, LoadY2 y2f, F& d, bool may_be_inf) {
    if (may_be_inf) {
      const bool z1 = x1.limbs_all_zero(), z2 = x2.limbs_all_zero();
      if (z1 || z2) {
        const bool i1 = z1 && y1f().limbs_all_zero(), i2 = z2 && y2f().limbs_all_zero();
        if (i1 || i2) { d = F::one(); return i1 ? (i2 ? INF : TAKE_Q) : TAKE_P; }
      }
    }
    d = F::prep(F::template sub<8, 1>(x2, x1));           // [3 -> 1 after prep where products need it, vb <= 4 + 8]
    if (!d.is_zero_mod_p()) return ADD;
    const F y1 = y1f(), y2 = y2f();
    if (F::prep(F::template sub<8, 1>(y2, y1)).is_zero_mod_p() && !y1.is_zero_mod_p()) { d = F::prep(F::dbl(y1)); return DBL; }
    d = F::one();
    return INF;                                           // P + (-P), or a doubled point of order two
  }
  // the affine sum given 1 / d
  HD static Affine<F> finish(int kind, const Affine<F>& p, const Affine<F>& q, const F& dinv) {
    if (kind == TAKE_P) return p;
    if (kind == TAKE_Q) return q;
    if (kind == INF) return {F::zero(), F::zero()};
    F lam;
    if (kind == DBL) {
      const F xx = F::sqr_nn(p.x);
      lam = F::mul_nn(F::prep(F::add(F::add(xx, xx), xx)), dinv);                  // 3 x^2 / (2 y)
    } else {
      lam = F::mul_nn(F::prep(F::template sub<8, 1>(q.y, p.y)), dinv);             // (y2 - y1) / (x2 - x1)
    }
    const F l2 = F::sqr_nn(lam);                                                    // [1, 2]
    const F x3 = F::wred(F::template sub<8, 1>(F::template sub<8, 1>(l2, p.x), q.x));   // [5, 18] -> [1, 3]
    const F t = F::prep(F::template sub<8, 1>(p.x, x3));                            // [3, <= 12]
    const F y3 = F::wred(F::template sub<8, 1>(F::mul_nn(lam, t), p.y));            // [3, 10] -> [1, 3]
    return {x3, y3};
  }
};

// One lane = the pieces order[lane + g * nlanes], g = 0, 1, ... of the longest-first schedule: a long one, a medium one, ..., a short one, so that
// every lane of the grid has about the same number of pairs and the grid can be exactly the lanes the chip holds (one round, no tail).
// `levels` tree levels, each a forward pass (prefix products of the denominators), one inversion, a backward pass (the sums).
// `pts`: affine points by sorted position (level results, in place); `pref`: [slot][lane].
template <class G, int OCC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
k_ba_levels(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ plen,
            const uint32_t* __restrict__ order, const uint32_t* __restrict__ nwork, uint32_t* __restrict__ pts, uint32_t* __restrict__ pref, uint32_t nlanes,
            int levels, uint32_t pref_slots) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  typedef BaOps<F> Ops;
  constexpr int FW = F::WORDS;
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nw_ = *nwork;
  if (lane >= nw_) return;
  uint32_t* const mypref = pref + (size_t)lane * FW;
  const size_t pstride = (size_t)nlanes * FW;
#pragma unroll 1
  for (int level = 0; level < levels; level++) {
    // ---- forward: prefix products of the denominators
    F acc = F::one();
    uint32_t slot = 0;
#pragma unroll 1
    for (uint32_t w = lane; w < nw_; w += nlanes) {
      const uint32_t pid = order[w];
      uint32_t cur, L;
      ba_piece_at(pstart[pid], plen[pid], level, cur, L);
#pragma unroll 1
      for (uint32_t j = 0; j + 1 < L; j += 2) {
        F x1, x2, d;
        const uint32_t *q1, *q2;
        uint32_t v1 = 0, v2 = 0;
        if (level == 0) {
          v1 = sorted[cur + j]; v2 = sorted[cur + j + 1];
          q1 = bases + (size_t)(v1 & 0x7fffffffu) * IO::AFF_WORDS; q2 = bases + (size_t)(v2 & 0x7fffffffu) * IO::AFF_WORDS;
        } else {
          q1 = pts + (size_t)(cur + j) * IO::AFF_WORDS; q2 = q1 + IO::AFF_WORDS;
        }
        x1 = F::load(q1); x2 = F::load(q2);
        (void)Ops::classify(x1, x2,
                            [&]() { F y = F::load(q1 + FW); return (v1 >> 31) ? F::norm(F::template neg<4, 1>(y)) : y; },
                            [&]() { F y = F::load(q2 + FW); return (v2 >> 31) ? F::norm(F::template neg<4, 1>(y)) : y; }, d, level != 0);
        if (slot < pref_slots) acc.store(mypref + (size_t)slot * pstride);       // (the arena holds pref_slots per lane: the host sized them from the piece bound)
        acc = F::mul_nn(acc, d);
        slot++;
      }
    }
    F inv = slot ? F::inv(acc) : F::one();
    // ---- backward: the sums, right-aligned in place.  The lane's pieces in reverse order: the last one first
    uint32_t cnt = (nw_ - lane + nlanes - 1) / nlanes;
#pragma unroll 1
    for (; cnt-- > 0;) {
      const uint32_t pid = order[lane + cnt * nlanes];
      uint32_t cur, L;
      ba_piece_at(pstart[pid], plen[pid], level, cur, L);
      if (L == 0) continue;
      const uint32_t M = (L + 1) / 2, out = cur + L - M;
      if (level == 0 && (L & 1)) {                           // the leftover of an odd run: fetched and signed (later levels: it is in place already)
        const uint32_t v = sorted[cur + L - 1];
        Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
        if (v >> 31) p.y = F::wred(F::template neg<4, 1>(p.y));
        IO::store_affine(pts + (size_t)(out + M - 1) * IO::AFF_WORDS, p);
      }
#pragma unroll 1
      for (uint32_t jj = L / 2; jj-- > 0;) {
        const uint32_t j = 2 * jj;
        const uint32_t *q1, *q2;
        uint32_t v1 = 0, v2 = 0;
        if (level == 0) {
          v1 = sorted[cur + j]; v2 = sorted[cur + j + 1];
          q1 = bases + (size_t)(v1 & 0x7fffffffu) * IO::AFF_WORDS; q2 = bases + (size_t)(v2 & 0x7fffffffu) * IO::AFF_WORDS;
        } else {
          q1 = pts + (size_t)(cur + j) * IO::AFF_WORDS; q2 = q1 + IO::AFF_WORDS;
        }
        // staged so that at most ~230 registers are live (two waves per SIMD): the x coordinates and the batch's running inverse first, the y
        // coordinates only once the two products of Montgomery's trick are done, y1 read again for the last subtraction
        const F x1 = F::load(q1), x2 = F::load(q2);
        auto y1f = [&]() { F y = F::load(q1 + FW); return (v1 >> 31) ? F::norm(F::template neg<4, 1>(y)) : y; };
        auto y2f = [&]() { F y = F::load(q2 + FW); return (v2 >> 31) ? F::norm(F::template neg<4, 1>(y)) : y; };
        F d;
        const int kind = Ops::classify(x1, x2, y1f, y2f, d, level != 0);
        slot--;
        F dinv = F::mul_nn(inv, F::load(mypref + (size_t)slot * pstride));
        inv = F::mul_nn(inv, d);
        BA_FENCE();
        uint32_t* const o = pts + (size_t)(out + jj) * IO::AFF_WORDS;
        if (kind == Ops::ADD || kind == Ops::DBL) {
          F lam;
          if (kind == Ops::DBL) {
            const F xx = F::sqr_nn(x1);
            lam = F::mul_nn(F::prep(F::add(F::add(xx, xx), xx)), dinv);                  // 3 x^2 / (2 y)
          } else {
            lam = F::mul_nn(F::prep(F::template sub<8, 1>(y2f(), y1f())), dinv);         // (y2 - y1) / (x2 - x1)
          }
          BA_FENCE();
          const F x3 = F::wred(F::template sub<8, 1>(F::template sub<8, 1>(F::sqr_nn(lam), x1), x2));   // [5, 18] -> [1, 3]
          const F t = F::prep(F::template sub<8, 1>(x1, x3));                            // [3, <= 12]
          x3.store(o);
          BA_FENCE();
          const F lt = F::mul_nn(lam, t);
          BA_FENCE();
          F::wred(F::template sub<8, 1>(lt, y1f())).store(o + FW);                        // [3, 10] -> [1, 3]
        } else if (kind == Ops::TAKE_P) {
          x1.store(o); y1f().store(o + FW);
        } else if (kind == Ops::TAKE_Q) {
          x2.store(o); y2f().store(o + FW);
        } else {
          F::zero().store(o); F::zero().store(o + FW);
        }
      }
    }
  }
}

// ... and the XYZZ chain over what is left of every piece (one lane per piece, as k_accumulate; the identity among the level results is skipped)
template <class G>
__global__ void __launch_bounds__(256) k_accumulate_ba(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ pstart,
                                                       const uint32_t* __restrict__ plen, const uint32_t* __restrict__ order,
                                                       const uint32_t* __restrict__ nwork, uint32_t* __restrict__ partials, int levels) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= *nwork) return;
  const uint32_t pid = order[tid];
  uint32_t cur, L;
  ba_piece_at(pstart[pid], plen[pid], levels, cur, L);
  Xyzz<F> acc = Xyzz<F>::identity();
#pragma unroll 1
  for (uint32_t k = 0; k < L; k++) {
    const Affine<F> p = IO::load_affine(pts + (size_t)(cur + k) * IO::AFF_WORDS);
    if (BaOps<F>::is_inf(p.x, p.y)) continue;
    xyzz_madd(acc, p);
  }
  IO::store_xyzz(partials + (size_t)pid * IO::XYZZ_WORDS, acc);
}

}  // namespace celo
