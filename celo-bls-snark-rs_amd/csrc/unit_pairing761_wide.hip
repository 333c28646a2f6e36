// Translation unit: the single-product LATENCY path of the BW6-761 pairing check (Groth16 `verify`: one product of four pairs,
// crates/epoch-snark/src/api/verifier.rs:35 reached from crates/bls-snark-sys/src/snark/mod.rs:23-45).
//
// The throughput kernels give a product to one lane group, start to finish: for ONE product that is one wave walking 63 + 189
// Miller iterations per pair, then a 449-bit and a 575-bit exponentiation back to back - 65 ms of a 70 ms `verify` (round 2
// profile).  The pieces are independent, so here they run side by side in different WAVES (different loops must not share a
// wave: divergent lanes would run them one after the other):
//   k_wide_miller   block 0: the f_{x+1} loops of all pairs (one lane group per pair); block 1: the f_{x^3-x^2-x} loops, four waves:
//                   the point steps | three iteration ranges of the accumulator updates (see the kernel)
//   k_wide_easy     one group: f = prod f1_i * frob(prod f2_i), the easy part m = f^((q^3-1)(q+1)), and m^q
//   k_wide_pow      block 0: m^R0, block 1: (m^q)^R1, signed-digit (NAF) ladders, each split over two waves: squarings | multiplications
//   k_wide_final    one group: the product of the two powers, == 1, export
// Same field elements as the throughput path at every stage (tests: the reference's Groth16 vector, GT values vs the oracle).
#include "pairing_lanes_kernels.h"
#include <mutex>
#include <vector>

namespace celo {
namespace {
typedef LP761 LP;
typedef LP::Tow Tow;
typedef LP::Pair Pair;
typedef LP::QB QB;
typedef QB::V V;
typedef Tow::E12 E12;

// ---- three lane groups per value ("super-group" of nine lanes): the quadratic level's Karatsuba products side by side.
// An Fq6 product is three independent Fq3 products (a a', b b', (a + b)(a' + b')), a squaring two, the sparse line product
// three: every lane group of a super-group holds the WHOLE Fq6 element, group r computes product r (one instruction stream: the
// operands are selected per group, never the code), the results travel by ds_bpermute and every group finishes the same
// combination.  An Fq6 product then costs the latency of ONE Fq3 product (2 dependent multiplication rounds instead of 6), a
// squaring 2 instead of 4, a line product 2 instead of 5: this path exists for the latency of a lone product, lanes are free.
__device__ __forceinline__ int sub3() { return QB::group() % 3; }                 // which product this lane group computes
template <int R> __device__ __forceinline__ V from_sub(const V& x) {               // the value the same lane of sub-group R holds
  const int addr = ((int)__lane_id() + 3 * (R - sub3())) << 2;
  V r;
#pragma unroll
  for (int i = 0; i < QB::NWORDS; i++) r.l[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)x.l[i]);
  return r;
}
__device__ __forceinline__ V pick3(const V& a0, const V& a1, const V& a2) {        // by sub-group, mask arithmetic (no exec regions)
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
  r.b = QB::wred(QB::template sub<4>(QB::template sub<4>(t, v0), v1));
  r.a = QB::wred(QB::add(v0, Tow::mul_by_gen(v1)));
  return r;
}
__device__ __forceinline__ E12 sqr12_w3(const E12& x) {                            // Tow::sqr12 (sub-group 2 repeats product 0)
  const V s2 = QB::wred(QB::add(x.a, Tow::mul_by_gen(x.b)));
  const V p = Tow::mul6(pick3(x.a, QB::add(x.a, x.b), x.a), pick3(x.b, s2, x.b));
  const V ab = from_sub<0>(p), t = from_sub<1>(p);
  const V c0 = QB::template sub<64>(QB::template sub<4>(t, ab), Tow::mul_by_gen(ab));
  return {QB::wred(c0), QB::wred(QB::dbl(ab))};
}
// f *= line at P (Pair::ell = one scaling round + mul_by_014): v0 = f.a (s0 + s1 u), v1 = f.b (s4 u), t = (f.a + f.b)(s0 + (s1 + s4) u) as
// ONE mul6_by_01 with per-group operands (v1's d0 is zero: a wasted product, but no divergence)
__device__ __forceinline__ void ell_w3(E12& f, const Pair::Line& l, const V& px, const V& py) {
  const V sc = QB::mul(QB::template sel<0>(l.c1, l.c2), QB::pick(px, py, py));
  const V s0 = l.c0, s1 = QB::template bcast<0>(sc), s4 = QB::template bcast<1>(sc);
  const V p = Tow::mul6_by_01(pick3(f.a, f.b, QB::add(f.a, f.b)), pick3(s0, QB::zero(), s0), pick3(s1, s4, QB::add(s1, s4)));
  const V v0 = from_sub<0>(p), v1 = from_sub<1>(p), t = from_sub<2>(p);
  f.b = QB::wred(QB::template sub<4>(QB::template sub<4>(t, v0), v1));
  f.a = QB::wred(QB::add(v0, Tow::mul_by_gen(v1)));
}
// ---- one level further for a lone value (the ladders of the hard part): the two multiplication rounds of an Fq3 product are
// independent as well (the own products x_j y_j and the Karatsuba cross products (x_i + x_j)(y_i + y_j)), so a value takes
// EIGHTEEN lanes - half hh = group / 3 of a super-group computes round hh - and an Fq6 product costs the latency of ONE base
// field multiplication.  Lane = 9 hh + 3 r + j.
__device__ __forceinline__ int half2() { return QB::group() / 3; }                 // 0 or 1 inside the 18-lane super-group at lanes 0 .. 17
template <int H> __device__ __forceinline__ V from_half(const V& x) {
  const int addr = ((int)__lane_id() + 9 * (H - half2())) << 2;
  V r;
#pragma unroll
  for (int i = 0; i < QB::NWORDS; i++) r.l[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)x.l[i]);
  return r;
}
__device__ __forceinline__ V pickh(const V& a0, const V& a1) {
  const uint32_t m = QB::lane_mask(half2() == 0);
  V r;
#pragma unroll
  for (int i = 0; i < QB::NWORDS; i++) r.l[i] = (a0.l[i] & m) | (a1.l[i] & ~m);
  return r;
}
__device__ __forceinline__ V mul6_h(const V& x, const V& y) {                      // Tow::mul6 with its two rounds side by side
  const V xs = QB::add(QB::template perm<QP(1, 0, 0)>(x), QB::template perm<QP(2, 1, 2)>(x));
  const V ys = QB::add(QB::template perm<QP(1, 0, 0)>(y), QB::template perm<QP(2, 1, 2)>(y));
  const V p = QB::mul(pickh(x, xs), pickh(y, ys));
  const V v = from_half<0>(p), c = from_half<1>(p);
  const V t = QB::template sub<4>(QB::template sub<4>(c, QB::template perm<QP(1, 0, 0)>(v)), QB::template perm<QP(2, 1, 2)>(v));
  const V vr = QB::template perm<QP(0, 2, 1)>(v);
  const V xw = QB::mul_nr(QB::template sel<0>(t, vr));
  return QB::wred(QB::add(QB::template sel<0>(vr, t), QB::template sel<2>(vr, xw)));
}
__device__ __forceinline__ int sub3h() { return QB::group() % 3; }
__device__ __forceinline__ E12 mul12_h(const E12& x, const E12& y) {
  const V p = mul6_h(pick3(x.a, x.b, QB::add(x.a, x.b)), pick3(y.a, y.b, QB::add(y.a, y.b)));
  const V v0 = from_sub<0>(p), v1 = from_sub<1>(p), t = from_sub<2>(p);
  E12 r;
  r.b = QB::wred(QB::template sub<4>(QB::template sub<4>(t, v0), v1));
  r.a = QB::wred(QB::add(v0, Tow::mul_by_gen(v1)));
  return r;
}
__device__ __forceinline__ E12 sqr12_h(const E12& x) {
  const V s2 = QB::wred(QB::add(x.a, Tow::mul_by_gen(x.b)));
  const V p = mul6_h(pick3(x.a, QB::add(x.a, x.b), x.a), pick3(x.b, s2, x.b));
  const V ab = from_sub<0>(p), t = from_sub<1>(p);
  const V c0 = QB::template sub<64>(QB::template sub<4>(t, ab), Tow::mul_by_gen(ab));
  return {QB::wred(c0), QB::wred(QB::dbl(ab))};
}
__device__ __forceinline__ void store12_h(uint32_t* p, const E12& f) { if (QB::group() == 0) LP::store12(p, f); }
// value-per-super-group storage: sub-group 0 writes, every sub-group reads
__device__ __forceinline__ void store12_w3(uint32_t* p, const E12& f) { if (sub3() == 0) LP::store12(p, f); }
constexpr int WIDE_SUPER = 7;                    // super-groups per wave (63 lanes): pairs per product on this path

// The f_{x^3-x^2-x} loop (189 iterations, the longest chain of the whole check) is cut further.  The point arithmetic does not
// depend on f: wave 0 of its block walks R <- 2R (+- Q) for all pairs and writes every step's line coefficients to a buffer in
// global memory (233 steps of 1 KB per pair), bumping a counter in LDS.  The accumulator updates f <- f^2 * line are split by
// ITERATION RANGE over three more waves: with F_h the value after h iterations, F_n = F_h^(2^(n-h)) * G where G runs the same
// recurrence over iterations h .. n-1 starting from 1.  Wave c takes iterations [CUT[c], CUT[c+1]) from 1 and then squares
// n - CUT[c+1] times; the product of the three results is the loop's value (k_wide_easy multiplies everything anyway).  The cuts
// balance the three chains given where the loop's 30 additions sit (all but one in the first half) and what the point wave
// can deliver, with the f_{x+1} loop (block 0, one wave) beside them.  Exact arithmetic: the same field element.
constexpr int WIDE_LINE_WORDS = 3 * 3 * 28;      // per step and pair: c0, c1, c2, each as the group's three lanes hold it
constexpr int WIDE_CONSUMERS = 3;
__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void line_store(uint32_t* p, const Pair::Line& l) {
  const int j = LP::QB::lane();
  l.c0.store(p + j * 28); l.c1.store(p + (3 + j) * 28); l.c2.store(p + (6 + j) * 28);
}
__device__ __forceinline__ Pair::Line line_load(const uint32_t* p) {
  const int j = LP::QB::lane();
  return {Fw::load(p + j * 28), Fw::load(p + (3 + j) * 28), Fw::load(p + (6 + j) * 28)};
}
__device__ __forceinline__ int loop2_digit(int it) { return T761::LOOP2_NAF[T761::LOOP2_LEN - 2 - it]; }   // iteration it = 0 .. LOOP2_LEN - 2
// ---- the point wave's steps on TWO lane groups per pair (six lanes: half hp = group % 2 of them), products that do not depend
// on each other in the same round: a doubling is 2 dependent rounds instead of 3 (Y Z and X Y next to the three squares; the
// products of the new point next to (2e)^2), an addition 4 instead of 5 (lambda q.y next to theta^2, lambda^2, theta q.x; the
// two last rounds merged).  Same formulas and values as Pair::double_step / add_step (pairing_lanes.h).
__device__ __forceinline__ int halfp() { return QB::group() & 1; }
template <int H> __device__ __forceinline__ V from_halfp(const V& x) {
  const int addr = ((int)__lane_id() + 3 * (H - halfp())) << 2;
  V r;
#pragma unroll
  for (int i = 0; i < QB::NWORDS; i++) r.l[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)x.l[i]);
  return r;
}
__device__ __forceinline__ V pickp(const V& a0, const V& a1) {
  const uint32_t m = QB::lane_mask(halfp() == 0);
  V r;
#pragma unroll
  for (int i = 0; i < QB::NWORDS; i++) r.l[i] = (a0.l[i] & m) | (a1.l[i] & ~m);
  return r;
}
__device__ __forceinline__ void double_step_p(V& Rc, Pair::Line& l) {
  const V X = QB::template bcast<0>(Rc), Y = QB::template bcast<1>(Rc), Z = QB::template bcast<2>(Rc);
  // round 1: half 0: X^2, Y^2, Z^2; half 1: Y Z, X Y, (unused)
  const V p1 = QB::mul(pickp(Rc, QB::pick(Y, X, Z)), pickp(Rc, QB::pick(Z, Y, Z)));
  const V r1 = from_halfp<0>(p1), q1 = from_halfp<1>(p1);
  const V b = QB::template bcast<1>(r1), c = QB::template bcast<2>(r1);
  const V e = QB::wred(QB::dbl(QB::dbl(QB::tpl(c))));                // B' * 3c = 12 c
  const V e_2 = QB::dbl(e);
  const V h = QB::dbl(QB::template bcast<0>(q1));                    // 2YZ
  const V f3 = QB::tpl(e);
  const V g = QB::add(b, f3);
  const V a2 = QB::dbl(QB::template bcast<1>(q1));                   // 2XY
  const V i = QB::template sub<4>(e, b);
  // round 2: half 0: 2a (b - f3) = X', g^2, 4b h = Z'; half 1: (2e)^2 in every lane
  const V p2 = QB::mul(pickp(QB::pick(a2, g, QB::dbl(QB::dbl(b))), e_2), pickp(QB::pick(QB::template sub<16>(b, f3), g, h), e_2));
  const V r3 = from_halfp<0>(p2), e2s = from_halfp<1>(p2);
  const V y3 = QB::wred(QB::template sub<8>(r3, QB::tpl(e2s)));      // lane 1: g^2 - 3 (2e)^2
  Rc = QB::template sel<1>(y3, r3);
  l.c0 = QB::wred(i);
  l.c1 = QB::wred(QB::tpl(QB::template bcast<0>(r1)));
  l.c2 = QB::wred(QB::template neg<16>(h));
}
__device__ __forceinline__ void add_step_p(V& Rc, const V& Qc, Pair::Line& l) {   // Qc: lane 0 = Q.x, lane 1 = +-Q.y
  const V X = QB::template bcast<0>(Rc), Y = QB::template bcast<1>(Rc), Z = QB::template bcast<2>(Rc);
  const V qx = QB::template bcast<0>(Qc), qy = QB::template bcast<1>(Qc);
  const V r1 = QB::mul(QB::template sel<0>(qy, qx), Z);               // round 1 (both halves alike)
  const V theta = QB::template sub<4>(Y, QB::template bcast<0>(r1)), lambda = QB::template sub<4>(X, QB::template bcast<1>(r1));
  // round 2: half 0: theta^2, lambda^2, theta q.x; half 1: lambda q.y in every lane
  const V p2 = QB::mul(pickp(QB::pick(theta, lambda, theta), lambda), pickp(QB::pick(theta, lambda, qx), qy));
  const V r2 = from_halfp<0>(p2), lq = from_halfp<1>(p2);
  const V c = QB::template bcast<0>(r2), d = QB::template bcast<1>(r2);
  const V r3 = QB::mul(QB::pick(lambda, Z, X), QB::pick(d, c, d));    // round 3: e, f, g (both halves alike)
  const V e = QB::template bcast<0>(r3), g = QB::template bcast<2>(r3);
  const V h = QB::template sub<8>(QB::add(e, QB::template bcast<1>(r3)), QB::dbl(g));
  // round 4: half 0: (e Y), e Y, Z e = Z'; half 1: lambda h = X', theta (g - h)
  const V p4 = QB::mul(pickp(QB::pick(e, e, Z), QB::template sel<0>(lambda, theta)), pickp(QB::pick(Y, Y, e), QB::template sel<0>(h, QB::template sub<16>(g, h))));
  const V r4 = from_halfp<0>(p4), r5 = from_halfp<1>(p4);
  l.c0 = QB::wred(QB::template sub<4>(QB::template bcast<2>(r2), lq));
  const V y3 = QB::wred(QB::template sub<4>(r5, r4));                 // lane 1: theta (g - h) - e Y
  Rc = QB::pick(r5, y3, r4);
  l.c1 = QB::wred(QB::template neg<8>(theta));
  l.c2 = QB::wred(lambda);
}
// f_out: k values of the f_{x+1} loops, then WIDE_CONSUMERS * k partial values of the f_{x^3-x^2-x} loops.  k <= WIDE_SUPER pairs.
// Waves that update an accumulator give every pair a super-group (pair = group / 3); the point wave of block 1 one group (pair = group).
__global__ void __launch_bounds__(64 * (1 + WIDE_CONSUMERS)) LANES_OCC
k_wide_miller(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1, const uint64_t* __restrict__ g2, const uint8_t* __restrict__ inf2,
              uint32_t* __restrict__ f_out, uint32_t k, uint32_t* __restrict__ lines) {
  constexpr int N = T761::LOOP2_LEN - 1;                                      // iterations of the second loop
  constexpr int CUT[WIDE_CONSUMERS + 1] = {0, 73, 133, N};   // balanced for: squaring 2 rounds, line product 3, doubling 2, addition 4
  constexpr int W = lanes_gt_words<LP>();
  __shared__ uint32_t produced;                                              // line steps written so far
  if (threadIdx.x == 0) produced = 0;
  __syncthreads();
  const int wave = (int)(threadIdx.x >> 6), g = QB::group();
  if (g >= LP::GROUPS) return;
  if (blockIdx.x == 1 && wave == 0) {                                         // the points of the second loop: two groups per pair
    const uint32_t i = (uint32_t)(g / 2);
    if (i >= k) return;
    const bool writer = halfp() == 0;
    const V Qc = LP::load_q(g2 + (size_t)i * LP::G2W);
    const V Qn = QB::template sel<1>(QB::wred(QB::template neg<4>(Qc)), Qc);  // (Q.x, -Q.y)
    V Rc = QB::template sel<2>(QB::one(), Qc);
    uint32_t s = 0;
#pragma unroll 1
    for (int it = 0; it < N; it++) {
      Pair::Line l;
      double_step_p(Rc, l);
      if (writer) line_store(lines + ((size_t)s * k + i) * WIDE_LINE_WORDS, l);
      lds_st(&produced, ++s);
      const int d = loop2_digit(it);
      if (d != 0) {
        add_step_p(Rc, d > 0 ? Qc : Qn, l);
        if (writer) line_store(lines + ((size_t)s * k + i) * WIDE_LINE_WORDS, l);
        lds_st(&produced, ++s);
      }
    }
    return;
  }
  const uint32_t i = (uint32_t)(g / 3);                                       // this super-group's pair
  if (i >= k) return;
  const bool dead = (inf1 && inf1[i]) || (inf2 && inf2[i]);
  const V px = LP::load_p(g1 + (size_t)i * LP::G1W, 0), py = LP::load_p(g1 + (size_t)i * LP::G1W, 1);
  if (blockIdx.x == 0) {                                                      // the f_{x+1} loop: one wave, points and accumulator together
    if (wave != 0) return;
    const V Qc = LP::load_q(g2 + (size_t)i * LP::G2W);
    V Rc = QB::template sel<2>(QB::one(), Qc);
    E12 f = Tow::one12();
#pragma unroll 1
    for (int b = 62; b >= 0; b--) {
      f = sqr12_w3(f);
      Pair::Line l;
      Pair::double_step(Rc, l);
      ell_w3(f, l, px, py);
      if ((T761::LOOP1 >> b) & 1) {
        Pair::add_step(Rc, Qc, l);
        ell_w3(f, l, px, py);
      }
    }
    if (dead) f = Tow::one12();
    store12_w3(f_out + (size_t)i * W, f);
    return;
  }
  const int c = wave - 1, lo = CUT[c], hi = CUT[c + 1];
  uint32_t s = 0;                                                             // the step that holds iteration lo's doubling line
  for (int it = 0; it < lo; it++) s += 1 + (loop2_digit(it) != 0 ? 1 : 0);
  E12 f = Tow::one12();
#pragma unroll 1
  for (int it = lo; it < hi; it++) {
    if (it != lo) f = sqr12_w3(f);
    const int steps = 1 + (loop2_digit(it) != 0 ? 1 : 0);
    for (int q = 0; q < steps; q++) {
      while (lds_ld(&produced) <= s) __builtin_amdgcn_s_sleep(4);
      const Pair::Line l = line_load(lines + ((size_t)s * k + i) * WIDE_LINE_WORDS);
      ell_w3(f, l, px, py);
      s++;
    }
  }
#pragma unroll 1
  for (int q = hi; q < N; q++) f = sqr12_w3(f);
  if (dead) f = Tow::one12();
  store12_w3(f_out + (size_t)((1 + c) * k + i) * W, f);
}
// f_in: k f1 values then nb f2 values (partial products of the second loops).  out: [0] = m (or the Miller value f itself when !do_fe), [1] = m^q
__global__ void __launch_bounds__(64) LANES_OCC k_wide_easy(const uint32_t* __restrict__ f_in, uint32_t k, uint32_t nb, uint32_t* __restrict__ out, int do_fe) {
  if (QB::group() >= 3) return;                                              // one super-group
  constexpr int W = lanes_gt_words<LP>();
  E12 a = LP::load12(f_in), b = LP::load12(f_in + (size_t)k * W);
  for (uint32_t i = 1; i < k; i++) a = mul12_w3(a, LP::load12(f_in + (size_t)i * W));
  for (uint32_t i = 1; i < nb; i++) b = mul12_w3(b, LP::load12(f_in + (size_t)(k + i) * W));
  E12 f = mul12_w3(a, Pair::frob1(b));
  if (do_fe) {
    f = mul12_w3(Tow::conj12(f), Tow::inv12(f));                             // f^(q^3 - 1)   (Pair::easy_part)
    f = mul12_w3(Pair::frob1(f), f);                                         // ^(q + 1)
    store12_w3(out + W, Pair::frob1(f));
  }
  store12_w3(out, f);
}
// One ladder per block, TWO waves per ladder.  Right-to-left: wave 0 walks the squaring chain m^(2^i) (len - 1 dependent
// squarings: the floor of this stage) and hands the powers that carry a non-zero digit to wave 1 through a ring in LDS; wave 1
// multiplies them (or their conjugates: m is unitary) into the result.  A multiplication costs 1.5 squarings and every third
// digit is non-zero, so wave 1 keeps up and the ladder takes the time of its squarings: the same group element as the
// left-to-right ladder, a third less latency.
constexpr int POW_RING = 8;
__global__ void __launch_bounds__(128) LANES_OCC k_wide_pow(const uint32_t* __restrict__ m_in, const int8_t* __restrict__ naf0, int len0, int neg0,
                                                            const int8_t* __restrict__ naf1, int len1, int neg1, uint32_t* __restrict__ out) {
  constexpr int W = lanes_gt_words<LP>();
  __shared__ uint32_t ring[POW_RING * W];
  __shared__ uint32_t produced, consumed;                                    // counts of ring entries written / read so far
  const bool second = blockIdx.x != 0;                                       // wave-uniform
  const int8_t* d = second ? naf1 : naf0;
  const int len = second ? len1 : len0;
  if (threadIdx.x == 0) { produced = 0; consumed = 0; }
  __syncthreads();
  if (QB::group() >= 6) return;                                              // one 18-lane super-group per wave works
  if (threadIdx.x < 64) {
    Tow::E12 cur = LP::load12(m_in + (second ? W : 0));
    uint32_t np = 0;
#pragma unroll 1
    for (int i = 0; i < len; i++) {
      if (d[i] != 0) {
        while (np - lds_ld(&consumed) >= (uint32_t)POW_RING) __builtin_amdgcn_s_sleep(4);
        store12_h(ring + (np % POW_RING) * W, cur);
        lds_st(&produced, ++np);
      }
      if (i + 1 < len) cur = sqr12_h(cur);
    }
  } else {
    Tow::E12 acc = Tow::one12();
    bool first = true;
    uint32_t nc = 0;
#pragma unroll 1
    for (int i = 0; i < len; i++) {
      const int di = d[i];
      if (di == 0) continue;
      while (lds_ld(&produced) <= nc) __builtin_amdgcn_s_sleep(4);
      Tow::E12 e = LP::load12(ring + (nc % POW_RING) * W);
      lds_st(&consumed, ++nc);
      if (di < 0) e = Tow::conj12(e);
      acc = first ? e : mul12_h(acc, e);
      first = false;
    }
    const bool neg = second ? neg1 != 0 : neg0 != 0;
    store12_h(out + (second ? W : 0), neg ? Tow::conj12(acc) : acc);
  }
}
__global__ void __launch_bounds__(64) LANES_OCC k_wide_final(const uint32_t* __restrict__ p, int do_fe, uint8_t* __restrict__ is_one, uint64_t* __restrict__ gt_ark) {
  if (QB::group() >= 3) return;
  constexpr int W = lanes_gt_words<LP>();
  E12 r = LP::load12(p);
  if (do_fe) r = mul12_w3(r, LP::load12(p + W));
  const bool one = Tow::is_one12(r);
  if (sub3() != 0) return;
  if (is_one && LP::writer()) is_one[0] = one ? 1 : 0;
  if (gt_ark) LP::to_ark12(r, gt_ark);
}

// signed-digit recoding of a little-endian multi-limb exponent (host, once)
std::vector<int8_t> naf_of(const uint64_t* limbs, int nlimbs) {
  std::vector<uint64_t> e(limbs, limbs + nlimbs);
  e.push_back(0);
  std::vector<int8_t> d;
  auto is_zero = [&] { for (uint64_t w : e) if (w) return false; return true; };
  while (!is_zero()) {
    int8_t digit = 0;
    if (e[0] & 1) {
      digit = (int8_t)(2 - (int)(e[0] & 3));                                // +1 if e = 1 (mod 4), -1 if e = 3 (mod 4)
      if (digit > 0) e[0] -= 1;                                              // e odd: no borrow
      else { size_t i = 0; while (++e[i] == 0) i++; }                        // e += 1 with carry
    }
    d.push_back(digit);
    for (size_t i = 0; i + 1 < e.size(); i++) e[i] = (e[i] >> 1) | (e[i + 1] << 63);
    e.back() >>= 1;
  }
  return d;
}
struct NafTables { int8_t* d0 = nullptr; int8_t* d1 = nullptr; int len0 = 0, len1 = 0; };
int naf_tables(NafTables& out) {                                             // one device copy per device
  static std::mutex mu;
  static NafTables tabs[MAX_DEVICES];
  std::lock_guard<std::mutex> lk(mu);
  NafTables& t = tabs[api_device()];
  if (!t.d0) {
    const std::vector<int8_t> n0 = naf_of(T761::R0_MAG, (int)(sizeof(T761::R0_MAG) / 8)), n1 = naf_of(T761::R1_MAG, (int)(sizeof(T761::R1_MAG) / 8));
    if (hipMalloc(&t.d0, n0.size()) != hipSuccess || hipMalloc(&t.d1, n1.size()) != hipSuccess) { t.d0 = nullptr; return 1; }
    if (hipMemcpy(t.d0, n0.data(), n0.size(), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(t.d1, n1.data(), n1.size(), hipMemcpyHostToDevice) != hipSuccess) {
      t.d0 = nullptr; return 1;
    }
    t.len0 = (int)n0.size(); t.len1 = (int)n1.size();
  }
  out = t;
  return 0;
}
}  // namespace

// One product of k <= WIDE_SUPER (7) pairs, everything enqueued on `s`.  d_f: room for (1 + WIDE_CONSUMERS) k + 4 GT values; d_lines: wide_lines_words(k) words.
size_t wide_lines_words_761(uint32_t k) { return (size_t)256 * k * WIDE_LINE_WORDS; }    // 189 doublings + 30 additions of the second loop, rounded up
int wide_product_761(const uint64_t* d_g1, const uint8_t* d_i1, const uint64_t* d_g2, const uint8_t* d_i2, uint32_t k, uint32_t* d_f, uint32_t* d_lines, uint8_t* d_one,
                     uint64_t* d_gt, int do_fe, hipStream_t s) {
  NafTables t;
  if (naf_tables(t)) return 1;
  constexpr int W = lanes_gt_words<LP>();
  uint32_t* d_m = d_f + (size_t)(1 + WIDE_CONSUMERS) * k * W;      // m, m^q
  uint32_t* d_p = d_m + 2 * W;                                     // m^R0, (m^q)^R1
  hipLaunchKernelGGL(k_wide_miller, dim3(2), dim3(64 * (1 + WIDE_CONSUMERS)), 0, s, d_g1, d_i1, d_g2, d_i2, d_f, k, d_lines);
  hipLaunchKernelGGL(k_wide_easy, dim3(1), dim3(64), 0, s, d_f, k, (uint32_t)WIDE_CONSUMERS * k, d_m, do_fe);
  if (do_fe) hipLaunchKernelGGL(k_wide_pow, dim3(2), dim3(128), 0, s, d_m, t.d0, t.len0, T761::R0_NEG ? 1 : 0, t.d1, t.len1, T761::R1_NEG ? 1 : 0, d_p);
  hipLaunchKernelGGL(k_wide_final, dim3(1), dim3(64), 0, s, do_fe ? d_p : d_m, do_fe, d_one, d_gt);
  return 0;
}
}  // namespace celo
