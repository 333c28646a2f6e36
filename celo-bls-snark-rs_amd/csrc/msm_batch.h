// Batched small MSMs (Batch::verify's shape: thousands of independent instances of a few hundred terms) and the GLV / GLS expansions of the BLS12-377 subgroup entry points.
// (part of the MSM pipeline: csrc/msm.h includes the pieces in order and carries the overview)
#pragma once

namespace celo {

// =====================================================================================================================
// Batched small MSMs: m independent instances (instance p owns points [offsets[p], offsets[p+1])), the shape of
// bls-crypto's Batch::verify (crates/bls-crypto/src/bls/batch.rs:69,76: one n-term G2 MSM + one n-term G1 MSM per batch,
// n = number of signers, a few hundred) when bls-snark-sys' batch_verify_strict (crates/bls-snark-sys/src/signatures.rs:358)
// hands over thousands of batches.  Per-instance Pippenger with a small window; the instances are the parallel axis.
//   k_batch_sort      one workgroup per instance: signed digits, per-window LDS counting sort, runs written to HBM
//   k_size_* + k_accumulate (shared with the big path): every (instance, window, bucket) run is a work item, longest first
//   k_batch_reduce    one lane per (instance, window): running sum over its <= 64 buckets
//   k_batch_horner_lanes  three lanes per instance: Horner over the windows, Jacobian result in arkworks form
// OR of every scalar, limb by limb: the batch path sizes its window count by the longest scalar actually present (Batch::verify
// hands over 136-bit exponents in 253-bit containers: 28 windows of 5 bits instead of 51, and no idle lanes in the per-window
// kernels)
// (round 4: 16-byte loads on a grid that covers the chip several times over - the 65 536-lane, 4-byte-load version took 0.43 ms for the
// 32 MB of config 3, on the critical path of both legs of every batch_verify call; limb q of a scalar sits in lane group q / 4)
template <int SW>
__global__ void __launch_bounds__(256) k_scalar_or(const uint32_t* __restrict__ scalars, size_t words, uint32_t* __restrict__ out) {
  static_assert(SW % 4 == 0, "scalars are whole 16-byte groups");
  constexpr size_t G4 = SW / 4;
  const size_t quads = words / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x / G4 * G4;    // a multiple of G4: a lane only ever sees one 16-byte group of the scalar
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4* q = (const uint4*)scalars;
  uint4 acc = {0u, 0u, 0u, 0u};
  for (i = i < stride ? i : quads; i < quads; i += stride) { const uint4 v = q[i]; acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w; }
  __shared__ uint32_t blk[SW];                                      // per block in LDS first: SW global atomics per block, not 4 per lane
  if (threadIdx.x < SW) blk[threadIdx.x] = 0;
  __syncthreads();
  uint32_t* o = blk + 4 * (((size_t)blockIdx.x * blockDim.x + threadIdx.x) % G4);
  if (acc.x) atomicOr(o, acc.x);
  if (acc.y) atomicOr(o + 1, acc.y);
  if (acc.z) atomicOr(o + 2, acc.z);
  if (acc.w) atomicOr(o + 3, acc.w);
  __syncthreads();
  if (threadIdx.x < SW && blk[threadIdx.x]) atomicOr(out + threadIdx.x, blk[threadIdx.x]);
}
template <int SW, int CB, int PT>
__global__ void __launch_bounds__(256) k_batch_sort(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf,
                                                    const uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted,
                                                    uint32_t* __restrict__ pstart, uint32_t* __restrict__ plen, const int NW) {
  constexpr uint32_t B = 1u << (CB - 1);
  static_assert(B <= 64, "batch path supports window sizes up to 7 bits");
  __shared__ uint32_t cnt[B], cur[B];
  const uint32_t inst = blockIdx.x;
  const uint32_t lo = offsets[inst], n = offsets[inst + 1] - lo;
  const size_t entry_base = (size_t)lo * NW;
  uint32_t s[PT][SW + 1];
  uint32_t carry[PT];
  bool live[PT];
#pragma unroll
  for (int q = 0; q < PT; q++) {
    uint32_t i = q * 256 + threadIdx.x;
    live[q] = i < n && !(inf && inf[lo + i]);
    carry[q] = 0;
#pragma unroll
    for (int k = 0; k <= SW; k++) s[q][k] = 0;
    if (i < n) {
      const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)(lo + i) * SW);
#pragma unroll
      for (int k = 0; k < SW / 4; k++) {
        uint4 v = sp[k];
        s[q][4 * k] = v.x; s[q][4 * k + 1] = v.y; s[q][4 * k + 2] = v.z; s[q][4 * k + 3] = v.w;
      }
    }
  }
#pragma unroll 1
  for (int w = 0; w < NW; w++) {
    if (threadIdx.x < B) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mag[PT], neg[PT];
    const int bit = w * CB;
    const int wi = bit >> 5, off = bit & 31;
#pragma unroll
    for (int q = 0; q < PT; q++) {
      uint32_t raw = 0;
      if (wi < SW) {
        // dynamic word index: select from the register array (SW <= 12)
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int k = 0; k <= SW; k++) { if (k == wi) w0 = s[q][k]; if (k == wi + 1) w1 = s[q][k]; }
        uint64_t two = ((uint64_t)w1 << 32) | w0;
        raw = (uint32_t)(two >> off) & ((1u << CB) - 1);
      }
      uint32_t d = raw + carry[q];
      neg[q] = d > B ? 1u : 0u;
      mag[q] = neg[q] ? ((1u << CB) - d) : d;
      carry[q] = neg[q];
      if (!live[q]) mag[q] = 0;
      if (mag[q]) atomicAdd(&cnt[mag[q] - 1], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of B <= 64 counters by the first wave
      uint32_t v = threadIdx.x < B ? cnt[threadIdx.x] : 0, x = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if ((int)threadIdx.x >= o) x += y;
      }
      if (threadIdx.x < B) {
        uint32_t excl = x - v;
        cur[threadIdx.x] = excl;
        size_t bucket = ((size_t)inst * NW + w) * B + threadIdx.x;
        pstart[bucket] = (uint32_t)(entry_base + (size_t)w * n + excl);
        plen[bucket] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PT; q++) {
      if (mag[q]) {
        uint32_t pos = atomicAdd(&cur[mag[q] - 1], 1u);
        sorted[entry_base + (size_t)w * n + pos] = (lo + q * 256 + threadIdx.x) | (neg[q] << 31);
      }
    }
    __syncthreads();
  }
}

// (Round 4, measured and not kept - DESIGN.md section 6: this kernel is NOT latency-bound at config 3's scale.  The G2 leg brings
// 11 windows x 4096 instances = 45 056 running sums of 64 full additions, 2.9 M full additions of Fq2 points in 3.1 ms = 0.93 G/s, 1.6x
// what a mixed addition costs k_accumulate - the 3.1 ms are work.  Six lanes per running sum (LanePoint on the hex backend): 3.46 ms;
// window sums by bit position - chains of 15 additions on 5x the lanes, the Horner pass taking one addition per bit - 3.7 + 1.2 ms
// against 3.1 + 0.6.  Kept from it: the bucket's coordinates are loaded where they are used, curve.h xyzz_add_mem - scratch 1264 -> 260 B.)
template <class G>
__global__ void __launch_bounds__(128) k_batch_reduce(const uint32_t* __restrict__ partials, const uint32_t* __restrict__ plen,
                                                      uint32_t* __restrict__ wsum, uint32_t B, uint32_t nvw) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t vw = blockIdx.x * blockDim.x + threadIdx.x;
  if (vw >= nvw) return;
  Xyzz<F> running = Xyzz<F>::identity(), acc = Xyzz<F>::identity();
  for (int b = (int)B - 1; b >= 0; b--) {
    size_t bucket = (size_t)vw * B + b;
    if (plen[bucket]) xyzz_add_mem(running, partials + bucket * IO::XYZZ_WORDS);   // the bucket's coordinates loaded where they are used (curve.h)
    xyzz_add(acc, running);
  }
  IO::store_xyzz(wsum + (size_t)vw * IO::XYZZ_WORDS, acc);
}

template <class G>
__global__ void __launch_bounds__(128) k_batch_horner(const uint32_t* __restrict__ wsum, uint64_t* __restrict__ out, uint32_t nw,
                                                      uint32_t c, uint32_t m) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  uint32_t inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= m) return;
  Xyzz<F> acc = Xyzz<F>::identity();
  for (int w = (int)nw - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < c; k++) acc = xyzz_dbl(acc);      // inlined: this chain is pure latency (c * nw dependent doublings)
    Xyzz<F> v = IO::load_xyzz(wsum + ((size_t)inst * nw + w) * IO::XYZZ_WORDS);
    xyzz_add(acc, v);
  }
  uint64_t* o = out + (size_t)inst * 3 * IO::ARK64;
  if (acc.is_identity() || acc.ZZ.is_zero_mod_p()) {
    F::zero().to_ark(o); F::one().to_ark(o + IO::ARK64); F::zero().to_ark(o + 2 * IO::ARK64);
  } else {
    F::mul(acc.X, acc.ZZ).to_ark(o);
    F::mul(acc.Y, acc.ZZZ).to_ark(o + IO::ARK64);
    acc.ZZ.to_ark(o + 2 * IO::ARK64);
  }
}

// The same Horner pass with THREE LANES PER INSTANCE (curve_lanes.h): the chain of c * nw dependent doublings is pure latency
// (136 of them per Batch::verify instance), one lane per instance leaves all but 64 waves of the chip idle, and spreading each
// doubling's independent products over a lane group more than halves its latency.  21 instances per 64-lane block.
template <class G>
__global__ void __launch_bounds__(64) k_batch_horner_lanes(const uint32_t* __restrict__ wsum, uint64_t* __restrict__ out, uint32_t nw,
                                                           uint32_t c, uint32_t m) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  typedef QTriT<FieldBase<F>> QB;
  typedef LanePoint<QB> LP;
  const int g = QB::group();
  const uint32_t inst = blockIdx.x * 21u + (uint32_t)g;
  if (g >= 21 || inst >= m) return;
  typename LP::Pt acc;
  acc.inf = true;
  for (int w = (int)nw - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < c; k++) LP::dbl(acc);
    const Xyzz<F> v = IO::load_xyzz(wsum + ((size_t)inst * nw + w) * IO::XYZZ_WORDS);     // every lane of the group loads the window sum
    const typename LP::P b = {v.X, v.Y, v.ZZ, v.ZZZ};
    LP::add(acc, b, v.is_identity());
  }
  if (QB::lane() != 0) return;
  uint64_t* o = out + (size_t)inst * 3 * IO::ARK64;
  if (acc.inf || acc.p.ZZ.is_zero_mod_p()) {
    F::zero().to_ark(o); F::one().to_ark(o + IO::ARK64); F::zero().to_ark(o + 2 * IO::ARK64);
  } else {
    F::mul(acc.p.X, acc.p.ZZ).to_ark(o);
    F::mul(acc.p.Y, acc.p.ZZZ).to_ark(o + IO::ARK64);
    acc.p.ZZ.to_ark(o + 2 * IO::ARK64);
  }
}

// G2 of BLS12-377: SIX lanes per instance - the two halves of every Fq2 coordinate in adjacent lanes (QHex377, pairing_lanes.h:
// one signed two-product Montgomery pass per lane and product instead of a three-product Karatsuba), so a lane executes half
// the instructions per doubling: the Horner chain of Batch::verify's key sums is latency and nothing else.  10 instances per
// 64-lane block.  LanePoint runs unchanged on the backend (same formulas, same decisions per group).
template <class G>   // G2_377 only (a template so that every translation unit including this header may hold a copy)
__global__ void __launch_bounds__(64) k_batch_horner_hex(const uint32_t* __restrict__ wsum, uint64_t* __restrict__ out, uint32_t nw, uint32_t c, uint32_t m) {
  typedef PointIO<Fq2> IO;
  typedef QHex377 QB;
  typedef LanePoint<QB> LP;
  constexpr int HW = Fq::WORDS;                      // device words per Fq half (a coordinate is c0 then c1)
  const int g = QB::group(), h = QB::hsel();
  const uint32_t inst = blockIdx.x * 10u + (uint32_t)g;
  if (g >= 10 || inst >= m) return;
  LP::Pt acc;
  acc.inf = true;
  for (int w = (int)nw - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < c; k++) LP::dbl(acc);
    const uint32_t* src = wsum + ((size_t)inst * nw + w) * IO::XYZZ_WORDS + h * HW;   // every lane loads its half of the window sum
    const LP::P b = {Fq::load(src), Fq::load(src + 2 * HW), Fq::load(src + 4 * HW), Fq::load(src + 6 * HW)};
    const int z = b.ZZ.limbs_all_zero() ? 1 : 0;                                       // the identity is stored as exact zeros
    const bool b_inf = (z & __builtin_amdgcn_ds_bpermute(((int)__lane_id() ^ 1) << 2, z)) != 0;
    LP::add(acc, b, b_inf);
  }
  if (QB::lane() != 0) return;
  uint64_t* o = out + (size_t)inst * 3 * IO::ARK64 + h * Fq::ARK64;
  if (acc.inf || QB::is_zero_u(acc.p.ZZ)) {
    Fq::zero().to_ark(o);
    (h ? Fq::zero() : Fq::one()).to_ark(o + IO::ARK64);
    Fq::zero().to_ark(o + 2 * IO::ARK64);
  } else {
    QB::mul(acc.p.X, acc.p.ZZ).to_ark(o);
    QB::mul(acc.p.Y, acc.p.ZZZ).to_ark(o + IO::ARK64);
    acc.p.ZZ.to_ark(o + 2 * IO::ARK64);
  }
}
template <class G> struct BatchHornerLanes {
  static void launch(const uint32_t* d_wsum, uint64_t* d_out, uint32_t nw, uint32_t c, uint32_t m, hipStream_t stream) {
    hipLaunchKernelGGL((k_batch_horner_lanes<G>), dim3((m + 20) / 21), dim3(64), 0, stream, d_wsum, d_out, nw, c, m);
  }
};
template <> struct BatchHornerLanes<G2_377> {
  static void launch(const uint32_t* d_wsum, uint64_t* d_out, uint32_t nw, uint32_t c, uint32_t m, hipStream_t stream) {
    hipLaunchKernelGGL((k_batch_horner_hex<G2_377>), dim3((m + 9) / 10), dim3(64), 0, stream, d_wsum, d_out, nw, c, m);
  }
};

// ---- GLV expansion of ONE MSM over BLS12-377 for bases in the prime-order subgroup (msm_bls12_377_g1_subgroup / _g2_subgroup).  G1:
// phi(x, y) = (beta x, y) acts on the subgroup as multiplication by -x^2 (wire.h proves it: the G1 subgroup test); G2: psi acts as [x],
// so psi^2 as [x^2].  With k = k0 + k1 x^2
//   [k]P = [k0]P + [k1] I(P),   I(P) = (beta x, -y) on G1, psi^2(P) on G2,     0 <= k0, k1 < 2^127   (gls.h glv_split_x2).
// n terms with 253-bit scalars become 2 n terms with 127-bit scalars: the same number of bucket additions (8 windows of 16 bits over
// 2 n points instead of 16 over n), HALF the windows - half the buckets to reduce, half the host's Horner chain.  Point i and its image
// sit at i and n + i; a base flagged as the identity gets zero scalars.  Replaces k_convert_bases on this path.
template <class G> struct GlvImage;          // [x^2]P of a subgroup point P, four or fewer field products
template <> struct GlvImage<G1_377> {        // (beta x, -y): phi(x, y) = (beta x, y) = -[x^2](x, y)
  HD static Affine<Fq> of(const Affine<Fq>& P) { return {Fq::mul(P.x, Fq::from_limbs(T377::BETA_GLV)), Fq::wred(Fq::norm(Fq::neg<4, 1>(P.y)))}; }
};
template <> struct GlvImage<G2_377> {        // psi^2(x, y) = (PSI_X^2 x, PSI_Y^2 y): psi acts on G2 as [x] (the GLS expansion below uses psi^j)
  HD static Affine<Fq2> of(const Affine<Fq2>& P) {
    const Fq kx = Fq::from_limbs(T377::PSI_X2), ky = Fq::from_limbs(T377::PSI_Y2);
    return {{Fq::mul(P.x.c0, kx), Fq::mul(P.x.c1, kx)}, {Fq::mul(P.y.c0, ky), Fq::mul(P.y.c1, ky)}};
  }
};
template <class G>   // G1_377 and G2_377
__global__ void __launch_bounds__(256) k_glv_expand(const uint64_t* __restrict__ ark, const uint8_t* __restrict__ inf, const uint32_t* __restrict__ scalars,
                                                    uint32_t n, uint32_t* __restrict__ dev_bases, uint32_t* __restrict__ sc2) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* s = ark + (size_t)i * 2 * IO::ARK64;
  const Affine<F> P = {F::from_ark(s), F::from_ark(s + IO::ARK64)};
  IO::store_affine(dev_bases + (size_t)i * IO::AFF_WORDS, P);
  IO::store_affine(dev_bases + ((size_t)n + i) * IO::AFF_WORDS, GlvImage<G>::of(P));
  uint32_t k[8], k0[4], k1[4];
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 a = sp[0], b = sp[1];
  k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z;
  k[7] = b.w & ((1u << (G::SCALAR_BITS - 224)) - 1u);      // bits from Fr::MODULUS_BITS up are ignored, as on the plain path (k_digits) and in ark-ec
  glv_split_x2<8>(k, k0, k1);
  const uint32_t keep = (inf && inf[i]) ? 0u : 0xffffffffu;
  reinterpret_cast<uint4*>(sc2)[i] = uint4{k0[0] & keep, k0[1] & keep, k0[2] & keep, k0[3] & keep};
  reinterpret_cast<uint4*>(sc2)[(size_t)n + i] = uint4{k1[0] & keep, k1[1] & keep, k1[2] & keep, k1[3] & keep};
}
template <class G> struct GlvExpand {
  static constexpr bool AVAILABLE = false;
  static constexpr int BITS = 0;
  static void launch(const uint64_t*, const uint8_t*, const uint32_t*, uint32_t, uint32_t*, uint32_t*, hipStream_t) {}
};
template <class G> struct GlvExpandX2 {      // the two BLS12-377 groups: k = k0 + k1 x^2, both halves below 2^127
  static constexpr bool AVAILABLE = true;
  static constexpr int BITS = 127;
  static void launch(const uint64_t* ark, const uint8_t* inf, const uint32_t* sc, uint32_t n, uint32_t* dev_bases, uint32_t* sc2, hipStream_t st) {
    hipLaunchKernelGGL((k_glv_expand<G>), dim3((n + 255) / 256), dim3(256), 0, st, ark, inf, sc, n, dev_bases, sc2);
  }
};
template <> struct GlvExpand<G1_377> : GlvExpandX2<G1_377> {};
template <> struct GlvExpand<G2_377> : GlvExpandX2<G2_377> {};

// ---- GLS expansion of a batch of G2 instances (BLS12-377): psi = twist^-1 o Frobenius o twist acts on the prime-order subgroup of
// E'(Fq2) as multiplication by the curve parameter x (proved in wire.h, where the same fact is the subgroup test), so
//   [k]P = [d0]P + [d1]psi(P) + [d2]psi^2(P) + [d3]psi^3(P),   k = d0 + d1 x + d2 x^2 + d3 x^3,  0 <= d_j < x < 2^64.
// psi^j(x, y) = (PSI_X^j conj^j(x), PSI_Y^j conj^j(y)) with PSI_X = (-5)^((q-1)/6), PSI_Y = (-5)^((q-1)/4) in Fq: four Fq products per
// image.  One workgroup per instance; instance p of n_p points becomes one of nd n_p points, block j holding psi^j of the originals,
// with 64-bit scalars in 16-byte containers; bases are written in device form (this replaces k_convert_bases).
// Division by x (normalised: its top bit is set) is Knuth's algorithm D in base 2^32 with the two-digit divisor (x >> 32, 1).
// One wave per block (blockIdx.y = which 64 points of the instance) and at most 128 registers: the expansion runs beside the other
// group's accumulate kernel (Batch::verify starts both MSMs at once), whose waves leave less than half a SIMD's register file - a
// 256-thread block of 334-register waves waited for four EMPTY SIMDs of one CU and took 4.5 ms for 0.6 ms of work.
template <class G, int NW, int ND>   // G2_377 only (a template so that every translation unit including this header may hold a copy)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_gls_expand(const uint64_t* __restrict__ ark, const uint8_t* __restrict__ inf, const uint32_t* __restrict__ scalars,
                                                    const uint32_t* __restrict__ offsets, uint32_t* __restrict__ dev_bases,
                                                    uint32_t* __restrict__ sc2, uint8_t* __restrict__ inf2) {
  constexpr int nd = ND;
  typedef PointIO<Fq2> IO;
  const uint32_t inst = blockIdx.x;
  const uint32_t lo = offsets[inst], n = offsets[inst + 1] - lo;
  for (uint32_t t = blockIdx.y * 64u + threadIdx.x; t < n; t += gridDim.y * 64u) {
    const uint64_t* s = ark + (size_t)(lo + t) * 2 * IO::ARK64;
    uint32_t d[4][2];
    gls_digits_base_x<NW, ND>(scalars + (size_t)(lo + t) * 8, d);
    const uint8_t fl = inf ? inf[lo + t] : 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j < nd) {
        const size_t e = (size_t)nd * lo + (size_t)j * n + t;
        uint4 w = {d[j][0], d[j][1], 0u, 0u};
        reinterpret_cast<uint4*>(sc2)[e] = w;
        if (inf2) inf2[e] = fl;
      }
    }
    // the images ONE COORDINATE HALF AT A TIME (round 4): psi^j(x0 + x1 u) = (kx_j x0, +-kx_j x1), so a half of P is read, scaled by the
    // nd - 1 constants and stored before the next is touched - 14 live registers of input instead of the point and its three images
    // (the first form kept them all alive: 1232 B/lane of scratch at the 128 registers this kernel has, 1.1 ms for config 3's 10^6 keys)
    uint32_t* out0 = dev_bases + ((size_t)nd * lo + t) * IO::AFF_WORDS;
#pragma unroll
    for (int h = 0; h < 4; h++) {                        // x.c0, x.c1, y.c0, y.c1
      const Fq v = Fq::from_ark(s + (size_t)h * Fq::ARK64);
      v.store(out0 + h * Fq::WORDS);
      const Fq vn = (h & 1) ? Fq::wred(Fq::norm(Fq::neg<4, 1>(v))) : v;     // conjugation: the u half changes sign for odd j
#pragma unroll
      for (int j = 1; j < 4; j++) {
        if (j < nd) {
          const uint32_t* kc = h < 2 ? (j == 1 ? T377::PSI_X1 : j == 2 ? T377::PSI_X2 : T377::PSI_X3) : (j == 1 ? T377::PSI_Y1 : j == 2 ? T377::PSI_Y2 : T377::PSI_Y3);
          Fq::mul((j & 1) ? vn : v, Fq::from_limbs(kc)).store(out0 + (size_t)j * n * IO::AFF_WORDS + h * Fq::WORDS);
        }
      }
      asm volatile("" ::: "memory");
    }
  }
}
template <class G> struct GlsExpand {
  static constexpr bool AVAILABLE = false;
  static void launch(const uint64_t*, const uint8_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, int, int, uint32_t*, uint32_t*, uint8_t*, hipStream_t) {}
};
template <> struct GlsExpand<G2_377> {
  static constexpr bool AVAILABLE = true;
  // bits = length of the longest scalar: the number of significant words and of digits are compile-time constants of the kernel
  static void launch(const uint64_t* ark, const uint8_t* inf, const uint32_t* sc, const uint32_t* off, uint32_t m, uint32_t max_n, int nd, int bits,
                     uint32_t* dev_bases, uint32_t* sc2, uint8_t* inf2, hipStream_t st) {
    const int nw = bits <= 96 ? 3 : (bits + 31) / 32;
    const dim3 grid(m, (max_n + 63) / 64);
#define CELO_GLS_CASE(NW_, ND_) \
    if (nw == NW_ && nd == ND_) { hipLaunchKernelGGL((k_gls_expand<G2_377, NW_, ND_>), grid, dim3(64), 0, st, ark, inf, sc, off, dev_bases, sc2, inf2); return; }
    CELO_GLS_CASE(3, 2) CELO_GLS_CASE(4, 2) CELO_GLS_CASE(4, 3) CELO_GLS_CASE(5, 3) CELO_GLS_CASE(6, 3) CELO_GLS_CASE(6, 4) CELO_GLS_CASE(7, 4)
#undef CELO_GLS_CASE
    hipLaunchKernelGGL((k_gls_expand<G2_377, 8, 4>), grid, dim3(64), 0, st, ark, inf, sc, off, dev_bases, sc2, inf2);
  }
};

}  // namespace celo
