// Translation unit: the LATENCY path of the BLS12-377 pairing check - verify / verify_pop / verify_sig of ONE signature
// (crates/bls-crypto/src/bls/public.rs:71-120, reached from verify_signature / verify_pop of bls-snark-sys), two or three pairs,
// or of the few that concurrent callers bring at once (capi.hip's combined launch): up to 768 products of <= 3 pairs, a block each.  The throughput kernels give a pair (or a product) to one six-lane group from start to finish; a lone product then costs
// one wave walking 63 Miller iterations and a 5-ladder final exponentiation with every multiplication of an Fq12 operation in
// sequence (4.8 ms through the FFI, twice one host core).  Same treatment as unit_pairing761_wide.hip, on the six-lane backend:
//   * a super-group of three six-lane groups (18 lanes) holds every Fq12 value three times; group r computes the r-th of the
//     independent products of an operation (the three Fq6 products of an Fq12 product, the two of a squaring, the two Fq2
//     products of a cyclotomic squaring, the three of the sparse line product) - one instruction stream, operands selected per
//     group by mask arithmetic - and the results cross by ds_bpermute;
//   * k377_wide_miller: wave 0 walks the point steps R <- 2R (+ Q) of all pairs and writes the 69 lines per pair to global memory;
//     three more waves run the accumulator updates f <- f^2 * line by ITERATION RANGE (F_n = F_h^(2^(n-h)) * G with G the same
//     recurrence over iterations h .. n-1 from 1; cuts 0 / 28 / 48 / 63 balance the chains: 413 product rounds instead of 950);
//   * k377_wide_final: the product of the partial values, then ark-ec's final exponentiation chain on the side-by-side operations.
// Exact arithmetic: the same field elements as the throughput path (GT values against the oracle, tests/test_pairing_gpu.py).
#include "pairing_lanes_kernels.h"

namespace celo {
namespace {
typedef LPH377 LP;
typedef LP::Tow Tow;
typedef LP::Pair Pair;
typedef LP::QB QB;
typedef QB::V V;
typedef Tow::E12 E12;
constexpr int W377 = lanes_gt_words<LP>();
constexpr int CONSUMERS = 3;
constexpr int LINE_WORDS = 3 * 6 * 16;           // per step and pair: c0, c1, c2 as the group's six lanes hold them (16-word slots)
constexpr int LINE_STEPS = 69;                   // 63 doublings + 6 additions (x = 0x8508c00000000001)
constexpr int SUPER = 3;                         // super-groups per wave

__device__ __forceinline__ int sub3() { return QB::group() % 3; }
template <int R> __device__ __forceinline__ V from_sub(const V& x) { return QB::from_addr(x, ((int)__lane_id() + 6 * (R - sub3())) << 2); }
__device__ __forceinline__ V pick3(const V& a0, const V& a1, const V& a2) {
  const int q = sub3();
  const uint32_t m0 = QB::lane_mask(q == 0), m1 = QB::lane_mask(q == 1);
  V r;
#pragma unroll
  for (int i = 0; i < QB::NWORDS; i++) {
    const uint32_t t = (a1.l[i] & m1) | (a2.l[i] & ~m1);
    r.l[i] = (a0.l[i] & m0) | (t & ~m0);
  }
  return r;
}
__device__ __forceinline__ E12 mul12_w3(const E12& x, const E12& y) {              // Tow::mul12_inl
  const V p = Tow::mul6(pick3(x.a, x.b, QB::add(x.a, x.b)), pick3(y.a, y.b, QB::add(y.a, y.b)));
  const V v0 = from_sub<0>(p), v1 = from_sub<1>(p), t = from_sub<2>(p);
  E12 r;
  r.b = QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(t, v0), v1));
  r.a = QB::wred(QB::add_l(v0, Tow::template mul_by_gen_k_l<4>(v1)));
  return r;
}
__device__ __forceinline__ E12 sqr12_w3(const E12& x) {                            // Tow::sqr12 (sub-group 2 repeats product 0)
  const V s2 = QB::lred(QB::add_l(x.a, Tow::template mul_by_gen_k_l<4>(x.b)));
  const V p = Tow::mul6(pick3(x.a, QB::add(x.a, x.b), x.a), pick3(x.b, s2, x.b));
  const V ab = from_sub<0>(p), t = from_sub<1>(p);
  const V c0 = QB::template sub_l<64>(QB::template sub_l<4>(t, ab), Tow::template mul_by_gen_k<4>(ab));
  return {QB::wred(c0), QB::lred(QB::dbl_l(ab))};
}
__device__ __forceinline__ E12 cyclo_w3(const E12& f) {                            // Tow::cyclotomic_sqr_inl: its two Fq2 product rounds side by side
  const V x = QB::template sel<1>(QB::template perm<QP(0, 0, 1)>(f.b), QB::template perm<QP(0, 0, 1)>(f.a));
  const V y = QB::template sel<1>(QB::template perm<QP(1, 2, 2)>(f.a), QB::template perm<QP(1, 1, 2)>(f.b));
  const V p = QB::mul(pick3(x, QB::add(x, y), x), pick3(y, QB::add(QB::template mul_nr_k_l<4>(y), x), y));
  const V tmp = from_sub<0>(p), m = from_sub<1>(p);
  const V o0 = QB::lred(QB::template sub_l<64>(QB::template sub_l<4>(m, tmp), QB::template mul_nr_k<4>(tmp)));
  const V ut = QB::template perm<QP(2, 0, 1)>(tmp);
  const V u = QB::dbl_l(QB::template sel<0>(QB::template mul_nr_k<4>(ut), ut));
  E12 z;
  z.a = QB::wred(QB::add_l(QB::dbl_l(QB::template sub_l<4>(o0, f.a)), o0));
  z.b = QB::wred(QB::add_l(QB::dbl_l(QB::add_l(u, f.b)), u));
  return z;
}
// f *= line at P (Pair::ell = one scaling + mul_by_034): A = f.a s0, b = f.b (s3 + s4 v), e = (f.a + f.b)((s0 + s3) + s4 v) as ONE
// mul6_by_01 with per-group operands (A's second coefficient is zero: a wasted product, no divergence)
__device__ __forceinline__ void ell_w3(E12& f, const Pair::Line& l, const LP::F& px, const LP::F& py) {
  const V sc = QB::mul_fp(QB::template sel<0>(l.c0, l.c1), QB::pickf(py, px, px));
  const V s0 = QB::template bcast<0>(sc), s3 = QB::template bcast<1>(sc), s4 = l.c2;
  const V p = Tow::mul6_by_01(pick3(f.a, f.b, QB::add(f.a, f.b)), pick3(s0, s3, QB::add(s0, s3)), pick3(QB::zero(), s4, s4));
  const V A = from_sub<0>(p), b = from_sub<1>(p), e = from_sub<2>(p);
  f.b = QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(e, A), b));
  f.a = QB::lred(QB::add_l(A, Tow::template mul_by_gen_k_l<4>(b)));
}
__device__ __forceinline__ void store12_w3(uint32_t* p, const E12& f) { if (sub3() == 0) LP::store12(p, f); }
__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void line_store(uint32_t* p, const Pair::Line& l) {
  const int q = QB::sub();
  l.c0.store(p + q * 16); l.c1.store(p + (6 + q) * 16); l.c2.store(p + (12 + q) * 16);
}
__device__ __forceinline__ Pair::Line line_load(const uint32_t* p) {
  const int q = QB::sub();
  return {Fq::load(p + q * 16), Fq::load(p + (6 + q) * 16), Fq::load(p + (12 + q) * 16)};
}
__device__ __forceinline__ int x_bit(int it) { return (int)((T377::X >> (62 - it)) & 1); }           // iteration it = 0 .. 62

// One block per product (its pairs: offsets[p] .. offsets[p + 1], at most SUPER of them).  kt = pairs of the whole call;
// f_out: CONSUMERS * kt partial Miller values ((c, pair) at c * kt + pair); lines: step-major, (step, pair) at step * kt + pair.
__global__ void __launch_bounds__(64 * (1 + CONSUMERS)) LANES_OCC
k377_wide_miller(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1, const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
                 const uint32_t* __restrict__ offsets, uint32_t* __restrict__ f_out, uint32_t kt, uint32_t* __restrict__ lines) {
  constexpr int N = 63;
  constexpr int CUT[CONSUMERS + 1] = {0, 28, 48, N};
  __shared__ uint32_t produced;
  if (threadIdx.x == 0) produced = 0;
  __syncthreads();
  const int wave = (int)(threadIdx.x >> 6), g = QB::group();
  const uint32_t first = offsets[blockIdx.x], k = offsets[blockIdx.x + 1] - first;     // this product's pairs
  if (g >= LP::GROUPS) return;
  if (wave == 0) {                                                            // the point steps: one six-lane group per pair
    if ((uint32_t)g >= k) return;
    const uint32_t i = first + (uint32_t)g;
    const V Qc = LP::load_q(g2 + (size_t)i * LP::G2W);
    V Rc = QB::template sel<2>(QB::one(), Qc);
    uint32_t s = 0;
#pragma unroll 1
    for (int it = 0; it < N; it++) {
      Pair::Line l;
      Pair::double_step(Rc, l);
      line_store(lines + ((size_t)s * kt + i) * LINE_WORDS, l);
      lds_st(&produced, ++s);
      if (x_bit(it)) {
        Pair::add_step(Rc, Qc, l);
        line_store(lines + ((size_t)s * kt + i) * LINE_WORDS, l);
        lds_st(&produced, ++s);
      }
    }
    return;
  }
  if (g >= 3 * SUPER || (uint32_t)(g / 3) >= k) return;
  const uint32_t i = first + (uint32_t)(g / 3);                               // this super-group's pair
  const bool dead = (inf1 && inf1[i]) || (inf2 && inf2[i]);
  const LP::F px = LP::load_p(g1 + (size_t)i * LP::G1W, 0), py = LP::load_p(g1 + (size_t)i * LP::G1W, 1);
  const int c = wave - 1, lo = CUT[c], hi = CUT[c + 1];
  uint32_t s = 0;
  for (int it = 0; it < lo; it++) s += 1 + x_bit(it);
  E12 f = Tow::one12();
#pragma unroll 1
  for (int it = lo; it < hi; it++) {
    if (it != lo) f = sqr12_w3(f);
    const int steps = 1 + x_bit(it);
    for (int q = 0; q < steps; q++) {
      while (lds_ld(&produced) <= s) __builtin_amdgcn_s_sleep(4);
      const Pair::Line l = line_load(lines + ((size_t)s * kt + i) * LINE_WORDS);
      ell_w3(f, l, px, py);
      s++;
    }
  }
#pragma unroll 1
  for (int q = hi; q < N; q++) f = sqr12_w3(f);
  if (dead) f = Tow::one12();
  store12_w3(f_out + (size_t)(c * kt + i) * W377, f);
}

__device__ __attribute__((noinline)) E12 exp_by_x_w3(const E12& f) {
  E12 acc = f;
#pragma unroll 1
  for (int i = 62; i >= 0; i--) {
    acc = cyclo_w3(acc);
    if ((T377::X >> i) & 1) acc = mul12_w3(acc, f);
  }
  return acc;
}
// one block per product: the product of its pairs' partial Miller values, then (do_fe) ark-ec's bls12 final exponentiation
// (pairing_lanes.h final_exponentiation_t) on the side-by-side operations
__global__ void __launch_bounds__(64) LANES_OCC k377_wide_final(const uint32_t* __restrict__ f_in, const uint32_t* __restrict__ offsets, uint32_t kt, int do_fe,
                                                                uint8_t* __restrict__ is_one, uint64_t* __restrict__ gt_ark) {
  if (QB::group() >= 3) return;                                              // one super-group
  const uint32_t p = blockIdx.x, lo = offsets[p], hi = offsets[p + 1];
  E12 f = Tow::one12();
  bool first = true;
  for (uint32_t i = lo; i < hi; i++)
    for (int c = 0; c < CONSUMERS; c++) {
      const E12 v = LP::load12(f_in + (size_t)(c * kt + i) * W377);
      f = first ? v : mul12_w3(f, v);
      first = false;
    }
  if (do_fe) {
    E12 f2 = Tow::inv12(f);
    E12 r = mul12_w3(Tow::conj12(f), f2);
    f2 = r;
    r = mul12_w3(Pair::template frob12<2>(r), f2);
    E12 y0 = Tow::conj12(cyclo_w3(r));
    E12 y5 = exp_by_x_w3(r);
    E12 y1 = cyclo_w3(y5);
    E12 y3 = mul12_w3(y0, y5);
    y0 = exp_by_x_w3(y3);
    E12 y2 = exp_by_x_w3(y0);
    E12 y4 = mul12_w3(exp_by_x_w3(y2), y1);
    y1 = exp_by_x_w3(y4);
    y3 = Tow::conj12(y3);
    y1 = mul12_w3(mul12_w3(y1, y3), r);
    y3 = Tow::conj12(r);
    y0 = Pair::template frob12<3>(mul12_w3(y0, r));
    y4 = Pair::template frob12<1>(mul12_w3(y4, y3));
    y5 = Pair::template frob12<2>(mul12_w3(y5, y2));
    y5 = mul12_w3(mul12_w3(y5, y0), y4);
    f = mul12_w3(y5, y1);
  }
  const bool one = Tow::is_one12(f);
  if (sub3() != 0) return;
  if (is_one && LP::writer()) is_one[p] = one ? 1 : 0;
  if (gt_ark) LP::to_ark12(f, gt_ark + (size_t)p * 72);
}
// The final exponentiation of a few thousand products (round 4: Batch::verify's 4096 verdicts, the exposed tail of config 3): the same
// chain on the side-by-side operations, THREE products per wave (super-group sg of a block takes product 3 * blockIdx.x + sg; groups
// 9 of a wave idles).  k_final_exp_slots gives a product one six-lane group - 205 waves for 2048 products, one per SIMD on a fifth of the
// chip, every Fq12 product 6 rounds and every cyclotomic squaring 2 in sequence; here they are 2 and 1 on 683 waves (used up to 3072
// products, while every wave still has a SIMD to itself: unit_pairing_lf.hip).
__device__ __forceinline__ E12 final_exp_w3(const E12& f_in) {
  E12 f2 = Tow::inv12(f_in);
  E12 r = mul12_w3(Tow::conj12(f_in), f2);
  f2 = r;
  r = mul12_w3(Pair::template frob12<2>(r), f2);
  E12 y0 = Tow::conj12(cyclo_w3(r));
  E12 y5 = exp_by_x_w3(r);
  E12 y1 = cyclo_w3(y5);
  E12 y3 = mul12_w3(y0, y5);
  y0 = exp_by_x_w3(y3);
  E12 y2 = exp_by_x_w3(y0);
  E12 y4 = mul12_w3(exp_by_x_w3(y2), y1);
  y1 = exp_by_x_w3(y4);
  y3 = Tow::conj12(y3);
  y1 = mul12_w3(mul12_w3(y1, y3), r);
  y3 = Tow::conj12(r);
  y0 = Pair::template frob12<3>(mul12_w3(y0, r));
  y4 = Pair::template frob12<1>(mul12_w3(y4, y3));
  y5 = Pair::template frob12<2>(mul12_w3(y5, y2));
  y5 = mul12_w3(mul12_w3(y5, y0), y4);
  return mul12_w3(y5, y1);
}
__global__ void __launch_bounds__(64) LANES_OCC k377_w3_final_products(const uint32_t* __restrict__ prod, uint8_t* __restrict__ is_one, uint64_t* __restrict__ gt_ark,
                                                                       uint32_t m) {
  const int g = QB::group();
  if (g >= 3 * SUPER) return;
  const uint32_t p = blockIdx.x * (uint32_t)SUPER + (uint32_t)(g / 3);
  const bool live = p < m;
  const E12 f = final_exp_w3(LP::load12(prod + (size_t)(live ? p : 0) * W377));          // a surplus super-group walks product 0 and stores nothing
  const bool one = Tow::is_one12(f);
  if (!live || sub3() != 0) return;
  if (is_one && LP::writer()) is_one[p] = one ? 1 : 0;
  if (gt_ark) LP::to_ark12(f, gt_ark + (size_t)p * 72);
}
// ---- the same with TWO lane groups per product, five products per wave (late round 4): 3073 ... 5120 products, where three groups per
// product would give SIMDs a second wave.  A cyclotomic squaring's two Fq2 product rounds still run side by side (1 round: they are 70 % of
// the chain's rounds), an Fq12 product takes two Fq6 steps instead of three (groups 0 | 1: x.a y.a | x.b y.b, then both the Karatsuba term).
__device__ __forceinline__ int sub2() { return QB::group() & 1; }
template <int R> __device__ __forceinline__ V from_sub2(const V& x) { return QB::from_addr(x, ((int)__lane_id() + 6 * (R - sub2())) << 2); }
__device__ __forceinline__ V pick2(const V& a0, const V& a1) { return QB::choose(sub2() == 0, a0, a1); }
__device__ __forceinline__ E12 mul12_w2(const E12& x, const E12& y) {
  const V p = Tow::mul6(pick2(x.a, x.b), pick2(y.a, y.b));
  const V t = Tow::mul6(QB::add(x.a, x.b), QB::add(y.a, y.b));
  const V v0 = from_sub2<0>(p), v1 = from_sub2<1>(p);
  E12 r;
  r.b = QB::wred(QB::template sub_l<4>(QB::template sub_l<4>(t, v0), v1));
  r.a = QB::wred(QB::add_l(v0, Tow::template mul_by_gen_k_l<4>(v1)));
  return r;
}
__device__ __forceinline__ E12 cyclo_w2(const E12& f) {
  const V x = QB::template sel<1>(QB::template perm<QP(0, 0, 1)>(f.b), QB::template perm<QP(0, 0, 1)>(f.a));
  const V y = QB::template sel<1>(QB::template perm<QP(1, 2, 2)>(f.a), QB::template perm<QP(1, 1, 2)>(f.b));
  const V p = QB::mul(pick2(x, QB::add(x, y)), pick2(y, QB::add(QB::template mul_nr_k_l<4>(y), x)));
  const V tmp = from_sub2<0>(p), m = from_sub2<1>(p);
  const V o0 = QB::lred(QB::template sub_l<64>(QB::template sub_l<4>(m, tmp), QB::template mul_nr_k<4>(tmp)));
  const V ut = QB::template perm<QP(2, 0, 1)>(tmp);
  const V u = QB::dbl_l(QB::template sel<0>(QB::template mul_nr_k<4>(ut), ut));
  E12 z;
  z.a = QB::wred(QB::add_l(QB::dbl_l(QB::template sub_l<4>(o0, f.a)), o0));
  z.b = QB::wred(QB::add_l(QB::dbl_l(QB::add_l(u, f.b)), u));
  return z;
}
__device__ __attribute__((noinline)) E12 exp_by_x_w2(const E12& f) {
  E12 acc = f;
#pragma unroll 1
  for (int i = 62; i >= 0; i--) {
    acc = cyclo_w2(acc);
    if ((T377::X >> i) & 1) acc = mul12_w2(acc, f);
  }
  return acc;
}
__device__ __forceinline__ E12 final_exp_w2(const E12& f_in) {
  E12 f2 = Tow::inv12(f_in);
  E12 r = mul12_w2(Tow::conj12(f_in), f2);
  f2 = r;
  r = mul12_w2(Pair::template frob12<2>(r), f2);
  E12 y0 = Tow::conj12(cyclo_w2(r));
  E12 y5 = exp_by_x_w2(r);
  E12 y1 = cyclo_w2(y5);
  E12 y3 = mul12_w2(y0, y5);
  y0 = exp_by_x_w2(y3);
  E12 y2 = exp_by_x_w2(y0);
  E12 y4 = mul12_w2(exp_by_x_w2(y2), y1);
  y1 = exp_by_x_w2(y4);
  y3 = Tow::conj12(y3);
  y1 = mul12_w2(mul12_w2(y1, y3), r);
  y3 = Tow::conj12(r);
  y0 = Pair::template frob12<3>(mul12_w2(y0, r));
  y4 = Pair::template frob12<1>(mul12_w2(y4, y3));
  y5 = Pair::template frob12<2>(mul12_w2(y5, y2));
  y5 = mul12_w2(mul12_w2(y5, y0), y4);
  return mul12_w2(y5, y1);
}
__global__ void __launch_bounds__(64) LANES_OCC k377_w2_final_products(const uint32_t* __restrict__ prod, uint8_t* __restrict__ is_one, uint64_t* __restrict__ gt_ark,
                                                                       uint32_t m) {
  const int g = QB::group();
  if (g >= 10) return;
  const uint32_t p = blockIdx.x * 5u + (uint32_t)(g >> 1);
  const bool live = p < m;
  const E12 f = final_exp_w2(LP::load12(prod + (size_t)(live ? p : 0) * W377));
  const bool one = Tow::is_one12(f);
  if (!live || sub2() != 0) return;
  if (is_one && LP::writer()) is_one[p] = one ? 1 : 0;
  if (gt_ark) LP::to_ark12(f, gt_ark + (size_t)p * 72);
}
}  // namespace

// m products (GT-shaped Miller values at prod, six-lane device form) -> verdicts and / or GT values; enqueued on `s`
void final_exp_w3_377(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, hipStream_t s) {
  hipLaunchKernelGGL(k377_w3_final_products, dim3((m + SUPER - 1) / SUPER), dim3(64), 0, s, prod, is_one, gt, m);
}
void final_exp_w2_377(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, hipStream_t s) {
  hipLaunchKernelGGL(k377_w2_final_products, dim3((m + 4) / 5), dim3(64), 0, s, prod, is_one, gt, m);
}

size_t wide_lines_words_377(uint32_t kt) { return (size_t)LINE_STEPS * kt * LINE_WORDS; }
// m products of <= 3 pairs each (kt pairs in all, d_off: their offsets on the device), everything enqueued on `s`.
// d_f: room for CONSUMERS * kt GT values; d_lines: wide_lines_words_377(kt) words.
int wide_products_377(const uint64_t* d_g1, const uint8_t* d_i1, const uint64_t* d_g2, const uint8_t* d_i2, const uint32_t* d_off, uint32_t m, uint32_t kt, uint32_t* d_f,
                      uint32_t* d_lines, uint8_t* d_one, uint64_t* d_gt, int do_fe, hipStream_t s) {
  hipLaunchKernelGGL(k377_wide_miller, dim3(m), dim3(64 * (1 + CONSUMERS)), 0, s, d_g1, d_i1, d_g2, d_i2, d_off, d_f, kt, d_lines);
  hipLaunchKernelGGL(k377_wide_final, dim3(m), dim3(64), 0, s, d_f, d_off, kt, do_fe, d_one, d_gt);
  return 0;
}
}  // namespace celo
