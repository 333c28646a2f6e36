// MSM stage 4: the bucket accumulation - one lane per piece (k_accumulate), Fq2 on lane pairs (k_accumulate_pair), the host-pointer pipeline's chunk form (k_accumulate_chunk, k_merge_carried).
// (part of the MSM pipeline: csrc/msm.h includes the pieces in order and carries the overview)
#pragma once

namespace celo {

// ---- 4. one lane per piece: XYZZ sum of its run of (signed) points
// Occupancy A/B (round 2, 2^20 terms): the 28-word fields (G2 of BLS12-377, BW6-761) take 256 VGPRs + ~160 AGPRs = ONE wave per
// SIMD.  Forcing two (-DCELO_ACC_OCC2: 520-744 B/lane of scratch instead of the AGPRs) is SLOWER - G2 8.69 -> 10.0 ms, BW6-761
// 16.0 -> 17.1 ms - because the one-wave kernels already issue an instruction every 5.1-5.3 cycles (the v_mad_u64_u32 rate:
// ~16-19k instructions per mixed addition x 2^20 x windows / 1024 SIMDs): their instruction stream has the independent work a
// second wave would bring.  What is left is the instruction count.
#ifdef CELO_ACC_OCC2
#define ACC_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define ACC_OCC
#endif
template <class G>
__global__ void __launch_bounds__(256) ACC_OCC k_accumulate(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ plen,
                                                    const uint32_t* __restrict__ order, const uint32_t* __restrict__ nwork,
                                                    uint32_t* __restrict__ partials) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= *nwork) return;
  uint32_t pid = order[tid];
  const uint32_t* run = sorted + pstart[pid];
  uint32_t len = plen[pid];
  Xyzz<F> acc = Xyzz<F>::identity();
  if constexpr (sizeof(F) <= 14 * sizeof(uint32_t)) {
    // 14-limb field: the first two points of the run are added as affine + affine (4 products + 2 squares instead of a mixed
    // addition's 8 + 2).  Same-box A/B (tools/ab_cmd_msm.sh, two rounds): G1 accumulate 0.392 / 0.393 -> 0.376 / 0.377 ms at 2^17
    // (runs of 8 points); at 2^20 (runs of 32) it is inside the run-to-run spread: 2.54 / 2.51 -> 2.56 / 2.51 ms.
    auto point = [&](uint32_t k) {
      const uint32_t v = run[k];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      if (v >> 31) p = affine_neg(p);
      return p;
    };
    uint32_t k0 = 0;
    if (len >= 2) { acc = xyzz_add_affine(point(0), point(1)); k0 = 2; }
    // (prefetching the next point, as the one-wave kernels below do, measures nothing here: two waves per SIMD hide the loads)
    for (uint32_t k = k0; k < len; k++) xyzz_madd(acc, point(k));
  } else {
    // the one-wave-per-SIMD kernels of the 28-word fields do not take the affine start: with the second inlined body they lose
    // (G2 8.15 -> 8.39 ms at 2^20, BW6-761 14.52 -> 14.72), and BW6-761 once lost 7 % to a mere restructuring of this loop
#ifdef CELO_ACC_NO_PREFETCH
    for (uint32_t k = 0; k < len; k++) {
      uint32_t v = run[k];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      if (v >> 31) p = affine_neg(p);
      xyzz_madd(acc, p);
    }
#else
    // one wave per SIMD: nothing else hides the two dependent global loads (index, then the point it names) at the head of an
    // iteration - the next point is fetched before the current addition starts (it waits in AGPRs: 169 -> 222 for G2, no scratch).
    // Same-box A/B (tools/ab_cmd_msm.sh): G2 8.32 -> 8.21 ms at 2^20, 1.186 -> 1.148 at 2^17, config 3 27.97 -> 27.71 ms;
    // BW6-761 unchanged (14.70 vs 14.70 ms).  -DCELO_ACC_NO_PREFETCH restores the plain loop.
    if (len) {
      uint32_t v = run[0];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      for (uint32_t k = 0; k < len; k++) {
        const uint32_t vn = run[k + 1 < len ? k + 1 : k];
        const Affine<F> pn = IO::load_affine(bases + (size_t)(vn & 0x7fffffffu) * IO::AFF_WORDS);
        if (v >> 31) p = affine_neg(p);
        xyzz_madd(acc, p);
        p = pn;
        v = vn;
      }
    }
#endif
  }
  IO::store_xyzz(partials + (size_t)pid * IO::XYZZ_WORDS, acc);
}

// ---- 4b. The accumulation over Fq2 on LANE PAIRS (late round 5).  k_accumulate<G2_377> holds a whole Fq2 mixed addition per lane: 256 VGPRs
// + 220 AGPRs, one wave per SIMD, an instruction every 5.1-5.7 cycles where the two-wave G1 kernel issues one every 4.0-4.2 (a lone wave
// cannot issue back to back; DESIGN.md section 4 showed it is not the dependent multiply-add chains).  Here the two halves c0, c1 of
// every Fq2 value sit on two ADJACENT lanes (half = lane & 1): a lane holds half the state (225 VGPRs, no AGPRs: two waves per SIMD) and
// an Fq2 product is QHex377::mul - the six-lane pairings' pair product: one signed two-product Montgomery pass per lane (Fp::mul2s,
// 2 x 196 + 182 multiply-adds: the same count as a half of the one-lane Fq2 product) after one DPP exchange with lane ^ 1.  Squarings and
// Y3 = R t - Y1 PPP are plain pair products here (10 per addition against 6 + 2 squarings + the fused Y3 of curve.h: 11 % more
// multiply-adds).  As register-resident loops (tools/ubench_g2_pair.hip) the pair form runs 2.54 G additions/s against 2.20 G/s; in the
// MSM the gain is 2-3 % alone on the device and 2-5 % beside other kernels (launch_accumulate below): the default; CELO_G2_PAIR=0 restores
// the one-lane kernel.  Same formulas (madd-2008-s, mdbl-2008-s-1), same stored bounds as curve.h (X < 19 p, Y < 7 p, ZZ, ZZZ
// < 3 p, limbs normalised: every hex:: operation carries), same partials layout: the reduction does not know which kernel ran.
// Control flow is PAIR-UNIFORM: both lanes of a pair walk the same piece and take the same branches (the exact-zero tests AND the two
// halves through DPP), so the partner lane is always there for the exchange.
struct PairAcc377 { Fp<P377> X, Y, ZZ, ZZZ; bool inf; };
__device__ __forceinline__ bool pair_both(bool z) {
  const int zi = z ? 1 : 0;
  return (zi & __builtin_amdgcn_mov_dpp(zi, 0xB1, 0xF, 0xF, true)) != 0;
}
// QHex377::mul in two steps: the first operand's exchanged form (X = the even lane's half in both lanes, cs = the odd lane's half times -5 | 1)
struct PairFirst {
  Fp<P377> X; int32_t cs[14];
  __device__ __forceinline__ explicit PairFirst(const Fp<P377>& a) {
    const uint32_t k = QHex377::hsel() ? 1u : 0u - 5u;
#pragma unroll
    for (int i = 0; i < 14; i++) {
      X.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xA0, 0xF, 0xF, true);
      cs[i] = (int32_t)((uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xF5, 0xF, 0xF, true) * k);
    }
  }
};
__device__ __forceinline__ Fp<P377> pair_mul(const PairFirst& f, const Fp<P377>& b, const Fp<P377>& bo) { return Fp<P377>::mul2s(f.X, b, f.cs, bo); }
__device__ __forceinline__ void pair_dbl_affine(PairAcc377& a, const Fp<P377>& px, const Fp<P377>& py) {
  typedef QHex377 Q;
  typedef Fp<P377> H;
  if (pair_both(py.is_zero_mod_p())) { a.inf = true; return; }
  const H U = Q::dbl(py);
  const H V = Q::mul(U, U), W = Q::mul(U, V), S = Q::mul(px, V), xx = Q::mul(px, px);
  const H M = Q::tpl(xx);
  const H X3 = Q::template sub<16>(Q::mul(M, M), Q::dbl(S));
  const H t = Q::template sub<32>(S, X3);
  a.Y = Q::template sub<4>(Q::mul(M, t), Q::mul(W, py));
  a.X = X3; a.ZZ = V; a.ZZZ = W; a.inf = false;
}
template <int V> __device__ __forceinline__ void pair_madd(PairAcc377& a, const Fp<P377>& px, const Fp<P377>& py) {
  typedef QHex377 Q;
  typedef Fp<P377> H;
  if (a.inf) { a.X = px; a.Y = py; a.ZZ = Q::one(); a.ZZZ = Q::one(); a.inf = false; return; }
  const H U2 = Q::mul(px, a.ZZ), S2 = Q::mul(py, a.ZZZ);
  const H Pd = Q::template sub<32>(U2, a.X), R = Q::template sub<16>(S2, a.Y);      // X < 19 p, Y < 7 p
  if (pair_both(Pd.is_zero_mod_p())) {
    if (pair_both(R.is_zero_mod_p())) pair_dbl_affine(a, px, py);
    else a.inf = true;
    return;
  }
  if constexpr (V == 0) {
    const H PP = Q::mul(Pd, Pd), PPP = Q::mul(Pd, PP), Qv = Q::mul(a.X, PP), R2 = Q::mul(R, R);
    const H X3 = Q::template sub<16>(R2, Q::add(Q::add(PPP, Qv), Qv));
    const H t = Q::template sub<32>(Qv, X3);
    a.Y = Q::template sub<4>(Q::mul(R, t), Q::mul(a.Y, PPP));
    a.ZZ = Q::mul(a.ZZ, PP);
    a.ZZZ = Q::mul(a.ZZZ, PPP);
    a.X = X3;
  } else {
    // the exchanged forms of operands that enter several products are built once: Pd and R as first operands (the even lane's half and the
    // scaled odd lane's half: 28 DPP moves + 14 multiplications by -5 | 1 each), PP and PPP as second operands (the partner's half: 14 DPP moves)
    const PairFirst fPd(Pd), fR(R);
    const H PP = pair_mul(fPd, Pd, Q::swap(Pd));
    const H PPo = Q::swap(PP);
    const H PPP = pair_mul(fPd, PP, PPo);
    const H PPPo = Q::swap(PPP);
    const H Qv = pair_mul(PairFirst(a.X), PP, PPo);
    const H R2 = pair_mul(fR, R, Q::swap(R));
    const H X3 = Q::template sub<16>(R2, Q::add(Q::add(PPP, Qv), Qv));
    const H t = Q::template sub<32>(Qv, X3);
    a.Y = Q::template sub<4>(pair_mul(fR, t, Q::swap(t)), pair_mul(PairFirst(a.Y), PPP, PPPo));
    a.ZZ = pair_mul(PairFirst(a.ZZ), PP, PPo);
    a.ZZZ = pair_mul(PairFirst(a.ZZZ), PPP, PPPo);
    a.X = X3;
  }
}
template <class G, int V>      // G = G2_377 (a template so that every translation unit that launches it owns an instantiation); V: see pair_madd
__global__ void __launch_bounds__(256) k_accumulate_pair(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ plen,
                                                            const uint32_t* __restrict__ order, const uint32_t* __restrict__ nwork,
                                                            uint32_t* __restrict__ partials) {
  static_assert(std::is_same<G, G2_377>::value, "lane pairs: Fq2 of BLS12-377");
  typedef Fp<P377> H;
  typedef PointIO<Fp2<P377>> IO;
  constexpr int HW = H::WORDS;                 // words per half; an Fq2 coordinate is c0 | c1
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t piece = tid >> 1, h = tid & 1;
  if (piece >= *nwork) return;
  const uint32_t pid = order[piece];
  const uint32_t* run = sorted + pstart[pid];
  const uint32_t len = plen[pid];
  PairAcc377 acc;
  acc.X = H::zero(); acc.Y = H::zero(); acc.ZZ = H::zero(); acc.ZZZ = H::zero(); acc.inf = true;
  if constexpr (V == 1) {
    for (uint32_t k = 0; k < len; k++) {
      const uint32_t v = run[k];
      const uint32_t* q = bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS + h * HW;
      const H px = H::load(q);
      H py = H::load(q + IO::FW);
      if (v >> 31) py = QHex377::template neg<4>(py);
      pair_madd<V>(acc, px, py);
    }
  } else if (len) {
    // the next point is fetched before the current addition starts: the two dependent loads (index, then the point it names) at the head of an
    // iteration are a larger share of a HALF addition than of a whole one
    uint32_t v = run[0];
    const uint32_t* q = bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS + h * HW;
    H px = H::load(q), py = H::load(q + IO::FW);
    for (uint32_t k = 0; k < len; k++) {
      const uint32_t vn = run[k + 1 < len ? k + 1 : k];
      const uint32_t* qn = bases + (size_t)(vn & 0x7fffffffu) * IO::AFF_WORDS + h * HW;
      const H nx = H::load(qn), ny = H::load(qn + IO::FW);
      if (v >> 31) py = QHex377::template neg<4>(py);
      pair_madd<V>(acc, px, py);
      px = nx; py = ny; v = vn;
    }
  }
  uint32_t* d = partials + (size_t)pid * IO::XYZZ_WORDS + h * HW;
  if (acc.inf) { acc.X = H::zero(); acc.Y = H::zero(); acc.ZZ = H::zero(); acc.ZZZ = H::zero(); }      // the identity is stored as exact zeros
  acc.X.store(d); acc.Y.store(d + IO::FW); acc.ZZ.store(d + 2 * IO::FW); acc.ZZZ.store(d + 3 * IO::FW);
}
// the launch of the accumulation: lane pairs for G2 of BLS12-377 (CELO_G2_PAIR=0: the one-lane kernel), one lane per piece otherwise
template <class G>
inline void launch_accumulate(uint32_t slots, hipStream_t stream, const uint32_t* d_bases, const uint32_t* d_sorted, const uint32_t* d_pstart, const uint32_t* d_plen,
                              const uint32_t* d_order, const uint32_t* d_nwork, uint32_t* d_partials) {
  if constexpr (std::is_same<G, G2_377>::value) {
    // 2 (default): lane pairs with the exchanged operand forms built once; 1: lane pairs, plain products with the next point prefetched;
    // 0: the one-lane kernel.  Same-box A/Bs (profiles/r5_ab_g2_lane_pairs.txt): one G2 MSM of 2^20 terms alone on the device 7.83-7.89 ms
    // one lane, 7.66-7.68 (1), 7.59-7.66 (2) - the pair kernels issue a VALU instruction every 3.95 cycles (the limit) where the one-lane
    // kernel issues one every 4.6, and need 12.5 % more of them (ten pair products of 574 multiply-adds per lane against six products, two
    // squarings and the fused Y3 of curve.h); BESIDE other kernels a two-wave kernel with half the registers shares the device better:
    // config 3 (the key leg beside the signature leg) 24.23-24.29 -> 23.67-23.84 ms (five alternations), config 5 (G2 beside G1 and the
    // pairings) 50.5-52.2 -> 48.9-49.0 ms.  The whole -m gpu suite passes on either.
    static const int pair = getenv("CELO_G2_PAIR") ? atoi(getenv("CELO_G2_PAIR")) : 2;
    if (pair == 2) {
      hipLaunchKernelGGL((k_accumulate_pair<G, 1>), dim3((2 * slots + 255) / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_partials);
      return;
    }
    if (pair) {
      hipLaunchKernelGGL((k_accumulate_pair<G, 0>), dim3((2 * slots + 255) / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_partials);
      return;
    }
  }
  hipLaunchKernelGGL((k_accumulate<G>), dim3((slots + 255) / 256), dim3(256), 0, stream, d_bases, d_sorted, d_pstart, d_plen, d_order, d_nwork, d_partials);
}

// The accumulation of ONE INDEX CHUNK of the host-pointer pipeline (round 5; run_device_windows' HostIn): the bases arrive over PCIe chunk
// by chunk and every chunk is accumulated while the next one is in flight.  The sort ran over (chunk, window) virtual windows, so a
// bucket's points of chunk k are a run of their own, cut into pieces as usual; the lane of a bucket's FIRST piece starts from the
// bucket's carried sum - the value the same bucket reached in the earlier chunks (carrier[(w, b)], zeroes = the identity before chunk 0) -
// and stores it back, so the chunks cost no additions that one pass over all n points would not have spent.  Further pieces of a bucket
// (runs longer than SEG: skewed scalars) go to `partials` as in k_accumulate and are folded into the carrier after the last chunk
// (k_combine_* with first = 1, k_merge_carried).  Chunk launches are stream-ordered: no two lanes ever hold the same carrier.
template <class G>
__global__ void __launch_bounds__(256) ACC_OCC k_accumulate_chunk(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ plen,
                                                    const uint32_t* __restrict__ order, const uint32_t* __restrict__ nwork,
                                                    uint32_t* __restrict__ partials, const uint32_t* __restrict__ pbucket,
                                                    uint32_t* __restrict__ carrier, uint32_t cont) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= *nwork) return;
  uint32_t pid = order[tid];
  const uint32_t* run = sorted + pstart[pid];
  uint32_t len = plen[pid];
  const uint32_t pb = pbucket[pid];
  uint32_t* dst = pb != 0xFFFFFFFFu ? carrier + (size_t)pb * IO::XYZZ_WORDS : partials + (size_t)pid * IO::XYZZ_WORDS;
  const bool carried = cont && pb != 0xFFFFFFFFu;
  Xyzz<F> acc = Xyzz<F>::identity();
  if (carried) acc = IO::load_xyzz(dst);
  if constexpr (sizeof(F) <= 14 * sizeof(uint32_t)) {
    auto point = [&](uint32_t k) {
      const uint32_t v = run[k];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      if (v >> 31) p = affine_neg(p);
      return p;
    };
    uint32_t k0 = 0;
    if (!carried && len >= 2) { acc = xyzz_add_affine(point(0), point(1)); k0 = 2; }
    for (uint32_t k = k0; k < len; k++) xyzz_madd(acc, point(k));
  } else {
    if (len) {
      uint32_t v = run[0];
      Affine<F> p = IO::load_affine(bases + (size_t)(v & 0x7fffffffu) * IO::AFF_WORDS);
      for (uint32_t k = 0; k < len; k++) {
        const uint32_t vn = run[k + 1 < len ? k + 1 : k];
        const Affine<F> pn = IO::load_affine(bases + (size_t)(vn & 0x7fffffffu) * IO::AFF_WORDS);
        if (v >> 31) p = affine_neg(p);
        xyzz_madd(acc, p);
        p = pn;
        v = vn;
      }
    }
  }
  IO::store_xyzz(dst, acc);
}
// after the last chunk: carrier(w, b) += the folded further pieces of bucket (w, b) of every chunk (piece pfirst + 1 of virtual window
// k vw + w, where its run was longer than SEG).  One lane per bucket; for uniform scalars almost no lane has anything to add.
template <class G>
__global__ void __launch_bounds__(128) k_merge_carried(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ pfirst,
                                                       const uint32_t* __restrict__ partials, uint32_t* __restrict__ carrier,
                                                       uint32_t real_total, uint32_t chunks, uint32_t SEG) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= real_total) return;
  bool any = false;
  for (uint32_t k = 0; k < chunks; k++) any |= counts[(size_t)k * real_total + t] > SEG;
  if (!any) return;
  Xyzz<F> acc = IO::load_xyzz(carrier + (size_t)t * IO::XYZZ_WORDS);
  for (uint32_t k = 0; k < chunks; k++) {
    const size_t g = (size_t)k * real_total + t;
    if (counts[g] > SEG) {
      const Xyzz<F> v = IO::load_xyzz(partials + (size_t)(pfirst[g] + 1) * IO::XYZZ_WORDS);
      xyzz_add_fn(acc, v);
    }
  }
  IO::store_xyzz(carrier + (size_t)t * IO::XYZZ_WORDS, acc);
}

}  // namespace celo
