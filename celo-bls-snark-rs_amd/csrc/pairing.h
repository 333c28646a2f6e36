// BLS12-377 optimal-ate pairing product check for gfx950.
//
// Replaces Bls12_377::product_of_pairings(..) == Fq12::one() at
//   crates/bls-crypto/src/bls/public.rs:102   (verify: 2 pairs)
//   crates/bls-crypto/src/bls/signature.rs:149 (batch_verify_hashes: n+1 pairs, ONE final exponentiation)
// and the per-batch checks of Batch::verify (crates/bls-crypto/src/bls/batch.rs:83) when many batches are verified
// together (BASELINE config 3: 4096 independent 2-pair products).
//
// Same formulas as ark-ec's bls12 engine so that Miller-loop outputs and GT values can be
// compared bit-for-bit with the oracle: homogeneous-projective doubling/addition steps with the D-twist line
// (-h, 3j, i) / (lambda, -theta, j), sparse mul_by_034, and arkworks' final-exponentiation chain (which yields the
// cube of the reduced pairing).  What differs is the shape: arkworks precomputes 69 line triples per G2 point
// (G2Prepared) and walks all pairs of a product inside one serial loop with a shared squaring; here the Miller loop is
// fused (line coefficients are consumed as they are produced, nothing is materialised in HBM) and runs either per pair
// (the product of the per-pair Miller values equals the shared-squaring multi-Miller value) or per product with a shared
// accumulator, then one final exponentiation per product writes the accept bit.
// This header holds the ONE-LANE formulation (a whole pairing's state in one lane: the host-checked and self-tested twin)
// and the engine; the kernels that run are the lane-parallel ones of pairing_lanes.h (three lanes per pairing).
// Pairs with a point at infinity contribute 1 (ark-ec bls12 miller_loop skips them).
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#include "runtime.h"
#endif
#include <cstdio>
#include <cstring>
#include "tower.h"

namespace celo {

struct G2Proj { Fq2 x, y, z; };   // homogeneous projective, clean values
struct Ell { Fq2 c0, c1, c2; };

HD Fq2 twist_b() { return {Fq::zero(), Fq::from_limbs(T377::TWIST_B_C1)}; }

// ark-ec bls12/g2.rs doubling_step
TW_FN void pairing_double_step(G2Proj& r, Ell& l) {
  const Fq two_inv = Fq::from_limbs(T377::TWO_INV);
  Fq2 a, b, c, e, g, h, j, e2, t;
  f2_mul(t, r.x, r.y);
  f2_mul_fp(a, t, two_inv);
  f2_sqr(b, r.y);
  f2_sqr(c, r.z);
  f2_mul(e, twist_b(), f2_tpl(c));
  Fq2 f = f2_tpl(e);                                  // vb 9
  f2_mul_fp(g, f2_add(b, f), two_inv);                // (b + f)/2
  f2_sqr(t, f2_add(r.y, r.z));
  h = f2_sub<8>(t, f2_add(b, c));                     // vb 11
  Fq2 i = f2_sub<4>(e, b);
  f2_sqr(j, r.x);
  f2_sqr(e2, e);
  f2_mul(t, a, f2_sub<16>(b, f));
  r.x = t;
  f2_sqr(t, g);
  r.y = f2_wred(f2_sub<16>(t, f2_tpl(e2)));
  f2_mul(t, b, h);
  r.z = t;
  l.c0 = f2_wred(f2_neg<16>(h));
  l.c1 = f2_wred(f2_tpl(j));
  l.c2 = f2_wred(i);
}
// ark-ec bls12/g2.rs addition_step
TW_FN void pairing_add_step(G2Proj& r, const Fq2& qx, const Fq2& qy, Ell& l) {
  Fq2 t, c, d, e, f, g;
  f2_mul(t, qy, r.z);
  Fq2 theta = f2_sub<4>(r.y, t);
  f2_mul(t, qx, r.z);
  Fq2 lambda = f2_sub<4>(r.x, t);
  f2_sqr(c, theta);
  f2_sqr(d, lambda);
  f2_mul(e, lambda, d);
  f2_mul(f, r.z, c);
  f2_mul(g, r.x, d);
  Fq2 h = f2_sub<8>(f2_add(e, f), f2_dbl(g));          // vb 14
  f2_mul(t, lambda, h);
  Fq2 nx = t;
  Fq2 u, v;
  f2_mul(u, theta, f2_sub<16>(g, h));
  f2_mul(v, e, r.y);
  r.y = f2_wred(f2_sub<4>(u, v));
  r.x = nx;
  f2_mul(t, r.z, e);
  r.z = t;
  f2_mul(u, theta, qx);
  f2_mul(v, lambda, qy);
  l.c0 = f2_wred(lambda);
  l.c1 = f2_wred(f2_neg<8>(theta));
  l.c2 = f2_wred(f2_sub<4>(u, v));
}
// ell: f *= line evaluated at P (D-twist: c0 *= P.y, c1 *= P.x)
TW_FN void pairing_ell(Fq12& f, const Ell& l, const Fq& px, const Fq& py) {
  Fq2 s0, s3;
  f2_mul_fp(s0, l.c0, py);
  f2_mul_fp(s3, l.c1, px);
  f12_mul_by_034(f, s0, s3, l.c2);
}
// f_{x,Q}(P) with its own accumulator
TW_FN void miller_loop_single(Fq12& f, const Fq& px, const Fq& py, const Fq2& qx, const Fq2& qy) {
  G2Proj r = {qx, qy, Fq2::one()};
  f = f12_one();
  Ell l;
  for (int i = 62; i >= 0; i--) {
    Fq12 t;
    f12_sqr(t, f);
    f = t;
    pairing_double_step(r, l);
    pairing_ell(f, l, px, py);
    if ((T377::X >> i) & 1) {
      pairing_add_step(r, qx, qy, l);
      pairing_ell(f, l, px, py);
    }
  }
}
TW_FN void exp_by_x(Fq12& r, const Fq12& f) {
  Fq12 acc = f;  // top bit of X
  for (int i = 62; i >= 0; i--) {
    Fq12 t;
    f12_cyclotomic_sqr(t, acc);
    acc = t;
    if ((T377::X >> i) & 1) { f12_mul(t, acc, f); acc = t; }
  }
  r = acc;
}
// ark-ec bls12 final_exponentiation (easy part, then the x-chain; result = reduced pairing cubed)
TW_FN void final_exponentiation(Fq12& out, const Fq12& f) {
  Fq12 f1 = f12_conj(f), f2, r, t;
  f12_inv(f2, f);
  f12_mul(r, f1, f2);
  f2 = r;
  f12_frob<2>(t, r);
  f12_mul(r, t, f2);
  Fq12 y0, y1, y2, y3, y4, y5;
  f12_cyclotomic_sqr(t, r); y0 = f12_conj(t);
  exp_by_x(y5, r);
  f12_cyclotomic_sqr(y1, y5);
  f12_mul(y3, y0, y5);
  exp_by_x(y0, y3);
  exp_by_x(y2, y0);
  exp_by_x(y4, y2);
  f12_mul(t, y4, y1); y4 = t;
  exp_by_x(y1, y4);
  y3 = f12_conj(y3);
  f12_mul(t, y1, y3); y1 = t;
  f12_mul(t, y1, r); y1 = t;
  y3 = f12_conj(r);
  f12_mul(t, y0, r); y0 = t;
  f12_frob<3>(t, y0); y0 = t;
  f12_mul(t, y4, y3); y4 = t;
  f12_frob<1>(t, y4); y4 = t;
  f12_mul(t, y5, y2); y5 = t;
  f12_frob<2>(t, y5); y5 = t;
  f12_mul(t, y5, y0); y5 = t;
  f12_mul(t, y5, y4); y5 = t;
  f12_mul(out, y5, y1);
}


// ================================================================== BW6-761 (Groth16 verify, crates/epoch-snark/src/api/verifier.rs:35)
// ark-ec models/bw6: two Miller loops (x+1, and the signed-digit expansion of x^3-x^2-x), f1 * frob(f2), then
// (q^3-1)(q+1) and the hard part m^R0(x) * (m^q)^R1(x) (El Housni-Guillevic, eprint 2020/351 Alg. 6).  G2 coordinates
// are in Fq (M-type sextic twist y^2 = x^3 + 4).
struct G2ProjW { Fw x, y, z; };
struct EllW { Fw c0, c1, c2; };
typedef Base761 BW;

TW_FN void bw6_double_step(G2ProjW& r, EllW& l) {
  Fw a, b, c, j, t, g2, e2s;
  BW::mul(a, r.x, r.y);
  BW::sqr(b, r.y);
  BW::sqr(c, r.z);
  Fw c3 = BW::tpl(c);
  Fw e = BW::wred(BW::dbl(BW::dbl(c3)));                 // B' * 3c = 12c
  Fw f = BW::tpl(e);                                     // vb 9
  Fw g = BW::add(b, f);
  BW::sqr(t, BW::add(r.y, r.z));
  Fw h = BW::sub<8>(t, BW::add(b, c));
  Fw i = BW::sub<4>(e, b);
  BW::sqr(j, r.x);
  BW::sqr(e2s, BW::dbl(e));
  BW::mul(t, BW::dbl(a), BW::sub<16>(b, f));
  r.x = t;
  BW::sqr(g2, g);
  r.y = BW::wred(BW::sub<8>(g2, BW::tpl(e2s)));
  BW::mul(t, BW::dbl(BW::dbl(b)), h);
  r.z = t;
  l.c0 = BW::wred(i);
  l.c1 = BW::wred(BW::tpl(j));
  l.c2 = BW::wred(BW::neg<16>(h));
}
TW_FN void bw6_add_step(G2ProjW& r, const Fw& qx, const Fw& qy, EllW& l) {
  Fw t, c, d, e, f, g, u, v;
  BW::mul(t, qy, r.z);
  Fw theta = BW::sub<4>(r.y, t);
  BW::mul(t, qx, r.z);
  Fw lambda = BW::sub<4>(r.x, t);
  BW::sqr(c, theta);
  BW::sqr(d, lambda);
  BW::mul(e, lambda, d);
  BW::mul(f, r.z, c);
  BW::mul(g, r.x, d);
  Fw h = BW::sub<8>(BW::add(e, f), BW::dbl(g));
  BW::mul(t, lambda, h);
  Fw nx = t;
  BW::mul(u, theta, BW::sub<16>(g, h));
  BW::mul(v, e, r.y);
  r.y = BW::wred(BW::sub<4>(u, v));
  r.x = nx;
  BW::mul(t, r.z, e);
  r.z = t;
  BW::mul(u, theta, qx);
  BW::mul(v, lambda, qy);
  l.c0 = BW::wred(BW::sub<4>(u, v));
  l.c1 = BW::wred(BW::neg<8>(theta));
  l.c2 = BW::wred(lambda);
}
TW_FN void bw6_ell(Fw6& f, const EllW& l, const Fw& px, const Fw& py) {
  Fw s1, s4;
  BW::mul(s1, l.c1, px);
  BW::mul(s4, l.c2, py);
  fw6_mul_by_014(f, l.c0, s1, s4);
}
TW_FN void bw6_miller_loop_single(Fw6& out, const Fw& px, const Fw& py, const Fw& qx, const Fw& qy) {
  EllW l;
  // f_{x+1,Q}(P)
  G2ProjW r = {qx, qy, Fw::one()};
  Fw6 f1 = quad_one<Base761>(), t;
  for (int i = 62; i >= 0; i--) {
    quad_sqr(t, f1); f1 = t;
    bw6_double_step(r, l);
    bw6_ell(f1, l, px, py);
    if ((T761::LOOP1 >> i) & 1) {
      bw6_add_step(r, qx, qy, l);
      bw6_ell(f1, l, px, py);
    }
  }
  // f_{x^3-x^2-x,Q}(P), signed digits
  r = {qx, qy, Fw::one()};
  const Fw nqy = BW::wred(BW::neg<4>(qy));
  Fw6 f2 = quad_one<Base761>();
  for (int i = T761::LOOP2_LEN - 1; i >= 1; i--) {
    if (i != T761::LOOP2_LEN - 1) { quad_sqr(t, f2); f2 = t; }
    bw6_double_step(r, l);
    bw6_ell(f2, l, px, py);
    int d = T761::LOOP2_NAF[i - 1];
    if (d != 0) {
      bw6_add_step(r, qx, d > 0 ? qy : nqy, l);
      bw6_ell(f2, l, px, py);
    }
  }
  fw6_frob1(t, f2);
  quad_mul(out, f1, t);
}
template <int NLIMBS> TW_FN void bw6_pow(Fw6& r, const Fw6& f, const uint64_t* e, int bits, bool neg) {
  Fw6 acc = f, t;  // top bit
  for (int i = bits - 2; i >= 0; i--) {
    quad_sqr(t, acc); acc = t;
    if ((e[i >> 6] >> (i & 63)) & 1) { quad_mul(t, acc, f); acc = t; }
  }
  r = neg ? quad_conj(acc) : acc;  // inverse == conjugate in the cyclotomic subgroup
}
TW_FN void bw6_final_exponentiation(Fw6& out, const Fw6& f) {
  Fw6 inv, a, m, t, p0, p1, mq;
  quad_inv(inv, f);
  quad_mul(a, quad_conj(f), inv);       // f^(q^3 - 1)
  fw6_frob1(t, a);
  quad_mul(m, t, a);                    // ^(q + 1)
  uint64_t e0[sizeof(T761::R0_MAG) / 8], e1[sizeof(T761::R1_MAG) / 8];
  for (unsigned i = 0; i < sizeof(T761::R0_MAG) / 8; i++) e0[i] = T761::R0_MAG[i];
  for (unsigned i = 0; i < sizeof(T761::R1_MAG) / 8; i++) e1[i] = T761::R1_MAG[i];
  bw6_pow<8>(p0, m, e0, T761::R0_BITS, T761::R0_NEG);
  fw6_frob1(mq, m);
  bw6_pow<9>(p1, mq, e1, T761::R1_BITS, T761::R1_NEG);
  quad_mul(out, p0, p1);
}

}  // namespace celo
#include "pairing_lanes.h"
namespace celo {

// ================================================================== pairing policies (one per curve) for the kernels below
struct PP377 {
  typedef Base377 BP;
  static constexpr int G1_ARK64 = 12, G2_ARK64 = 24;
  typedef struct LaneLaunch377 LL;     // kernels: pairing_lanes_kernels.h (the one-lane functions above stay as the host- and self-tested twin)
};
struct PP761 {
  typedef Base761 BP;
  static constexpr int G1_ARK64 = 24, G2_ARK64 = 24;
  typedef struct LaneLaunch761 LL;
};

#if defined(__HIPCC__)
// ---------------------------------------------------------------- lane-parallel kernels: 3 lanes per pairing / product.
// The kernels (pairing_lanes_kernels.h) are instantiated in their own translation units, which also define these launchers.
#define CELO_DECLARE_LANE_LAUNCH(NAME)                                                                                              \
  struct NAME {                                                                                                                     \
    static void miller(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, uint32_t* f, uint32_t n, hipStream_t s); \
    static void miller_product(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off,  \
                               uint32_t* prod, uint32_t m, hipStream_t s);                                                         \
    static void miller_product2(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off, \
                                uint32_t* prod, uint32_t m, hipStream_t s);   /* every product has <= 2 pairs */                   \
    static bool has_prepared();     /* products of <= 2 pairs whose first pair shares one G2 point: prepared lines */              \
    static void first_q_same(const uint64_t* g2, const uint32_t* off, uint32_t m, uint32_t* flag, hipStream_t s);                  \
    static void prepare_lines(const uint64_t* q0, uint32_t* lines, hipStream_t s);                                                 \
    static void miller_prepared(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off, \
                                const uint32_t* lines, uint32_t* prod, uint32_t m, hipStream_t s);                                 \
    static bool has_split();        /* a few thousand verify-shaped products: each cut in two by iteration range */                \
    static void miller_prepared_split(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2,                 \
                                      const uint32_t* lines, uint32_t* f, uint32_t m, hipStream_t s);                               \
    static void gt_product(const uint32_t* f, const uint32_t* off, uint32_t* prod, uint32_t m, hipStream_t s);                     \
    static void gt_tree(const uint32_t* in, uint32_t* out, uint32_t n_in, hipStream_t s);                                          \
    static void final_exp(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, int do_fe, hipStream_t s);              \
  };
CELO_DECLARE_LANE_LAUNCH(LaneLaunch377)
CELO_DECLARE_LANE_LAUNCH(LaneLaunch761)

#define PAIR_HIP_OK(x)                                                                                          \
  do {                                                                                                          \
    hipError_t e_ = (x);                                                                                        \
    if (e_ != hipSuccess) {                                                                                     \
      fprintf(stderr, "[celo-amd] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);         \
      return 1;                                                                                                 \
    }                                                                                                           \
  } while (0)

struct PairingTimings { float miller = 0, product = 0, final_exp = 0, total = 0; };
// single-product latency path (unit_pairing761_wide.hip): the loops and exponentiations of ONE product side by side in several waves
int wide_product_761(const uint64_t* d_g1, const uint8_t* d_i1, const uint64_t* d_g2, const uint8_t* d_i2, uint32_t k, uint32_t* d_f, uint32_t* d_lines, uint8_t* d_one,
                     uint64_t* d_gt, int do_fe, hipStream_t s);
size_t wide_lines_words_761(uint32_t k);
int wide_products_377(const uint64_t* d_g1, const uint8_t* d_i1, const uint64_t* d_g2, const uint8_t* d_i2, const uint32_t* d_off, uint32_t m, uint32_t kt, uint32_t* d_f,
                      uint32_t* d_lines, uint8_t* d_one, uint64_t* d_gt, int do_fe, hipStream_t s);
size_t wide_lines_words_377(uint32_t k);
template <class PP> struct WideProduct {   // curves without such a path
  static constexpr bool available = false;
  static constexpr uint32_t MAX_PAIRS = 0, MAX_PRODUCTS = 0;
  static size_t lines_words(uint32_t) { return 0; }
  static int run(const uint64_t*, const uint8_t*, const uint64_t*, const uint8_t*, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint8_t*, uint64_t*, int, hipStream_t) { return 1; }
};
template <> struct WideProduct<PP377> {
  static constexpr bool available = true;
  static constexpr uint32_t MAX_PAIRS = 3;    // per product: one super-group of 18 lanes per pair (unit_pairing377_wide.hip)
  static constexpr uint32_t MAX_PRODUCTS = 768;   // a block of four waves each (measured: 512 products 2.9 ms, 768: 3.5, 1024: 5.1 against 4.7 ms on the throughput kernels)
  static size_t lines_words(uint32_t k) { return wide_lines_words_377(k); }
  static int run(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t* off, uint32_t m, uint32_t k, uint32_t* f, uint32_t* lines,
                 uint8_t* one, uint64_t* gt, int do_fe, hipStream_t s) { return wide_products_377(g1, i1, g2, i2, off, m, k, f, lines, one, gt, do_fe, s); }
};
template <> struct WideProduct<PP761> {
  static constexpr bool available = true;
  static constexpr uint32_t MAX_PAIRS = 7;    // one super-group of nine lanes per pair (unit_pairing761_wide.hip)
  static constexpr uint32_t MAX_PRODUCTS = 1;
  static size_t lines_words(uint32_t k) { return wide_lines_words_761(k); }
  static int run(const uint64_t* g1, const uint8_t* i1, const uint64_t* g2, const uint8_t* i2, const uint32_t*, uint32_t, uint32_t k, uint32_t* f, uint32_t* lines,
                 uint8_t* one, uint64_t* gt, int do_fe, hipStream_t s) { return wide_product_761(g1, i1, g2, i2, k, f, lines, one, gt, do_fe, s); }
};

template <class PP> class PairingEngine {
 public:
  typedef QuadIO<typename PP::BP> IO;
  ~PairingEngine() { release(); }
  void release() {
    if (arena) { (void)hipFree(arena); arena = nullptr; arena_bytes = 0; }
    for (int i = 0; i < 4; i++) if (ev[i]) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
  }
  PairingTimings tm;
  hipStream_t own_stream() { return stream_.get(); }
  // Host pointers.  m products; product p covers pairs [offsets[p], offsets[p+1]) (offsets[m] = total pairs k).
  // mode: 0 = full check (Miller + final exp), 1 = Miller loop product only (no final exp; test hook)
  // out_is_one[m] (may be null), out_gt[m*72] in arkworks Montgomery form (may be null).
  int run(const uint64_t* g1, const uint8_t* inf1, const uint64_t* g2, const uint8_t* inf2, const uint32_t* offsets, size_t m,
          uint8_t* out_is_one, uint64_t* out_gt, int mode, hipStream_t stream) {
    if (m == 0) return 0;
    const uint32_t k = offsets[m];
    Staged st;
    if (stage(k, m, &st)) return 1;
    if (k) {
      PAIR_HIP_OK(hipMemcpyAsync(st.d_g1, g1, (size_t)k * PP::G1_ARK64 * 8, hipMemcpyHostToDevice, stream));
      PAIR_HIP_OK(hipMemcpyAsync(st.d_g2, g2, (size_t)k * PP::G2_ARK64 * 8, hipMemcpyHostToDevice, stream));
      if (inf1) PAIR_HIP_OK(hipMemcpyAsync(st.d_i1, inf1, k, hipMemcpyHostToDevice, stream));
      if (inf2) PAIR_HIP_OK(hipMemcpyAsync(st.d_i2, inf2, k, hipMemcpyHostToDevice, stream));
    }
    return run_staged(offsets, m, inf1 != nullptr, inf2 != nullptr, out_is_one, out_gt, mode, stream);
  }
  // The two halves of run(), for callers whose pairs are produced ON the device (batch verification: the normalised MSM
  // results are written straight into the input slots): stage() lays out the workspace for k pairs in m products and hands
  // out the device input slots; run_staged() runs the check on whatever those slots hold by then (stream order).
  struct Staged { uint64_t* d_g1; uint64_t* d_g2; uint8_t* d_i1; uint8_t* d_i2; };
  int stage(uint32_t k, size_t m, Staged* st) {
    const size_t W = IO::WORDS;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += (b + 255) & ~size_t(255); return o; };
    if (lay.k != k || lay.m != m) lines_valid = false;      // another layout: whatever sat at the cached lines' place may be overwritten
    lay.k = k; lay.m = m;
    lay.o_g1 = take((size_t)k * PP::G1_ARK64 * 8 + 8); lay.o_g2 = take((size_t)k * PP::G2_ARK64 * 8 + 8); lay.o_i1 = take(k + 8); lay.o_i2 = take(k + 8);
    lay.o_off = take((m + 1) * 4); lay.o_f = take((4 * (size_t)k + 8) * W * 4); lay.o_f2 = take(((size_t)k / 2 + 2) * W * 4);
    lay.o_prod = take((size_t)m * W * 4); lay.o_one = take(m + 8); lay.o_gt = take((size_t)m * 72 * 8);
    lay.o_lines = take((size_t)69 * 96 * 4 + 256); lay.o_flag = take(256);
    lay.o_wide = take(WideProduct<PP>::available && m <= WideProduct<PP>::MAX_PRODUCTS && k <= m * WideProduct<PP>::MAX_PAIRS ? WideProduct<PP>::lines_words(k) * 4 : 0);
    if (ensure(off)) return 1;
    char* A = arena;
    *st = {(uint64_t*)(A + lay.o_g1), (uint64_t*)(A + lay.o_g2), (uint8_t*)(A + lay.o_i1), (uint8_t*)(A + lay.o_i2)};
    return 0;
  }
  int run_staged(const uint32_t* offsets, size_t m, bool has_inf1, bool has_inf2, uint8_t* out_is_one, uint64_t* out_gt, int mode, hipStream_t stream) {
    if (m == 0) return 0;
    if (m != lay.m || offsets[m] != lay.k) return 2;
    const uint32_t k = lay.k;
    const size_t W = IO::WORDS;
    char* A = arena;
    uint64_t* d_g1 = (uint64_t*)(A + lay.o_g1); uint64_t* d_g2 = (uint64_t*)(A + lay.o_g2);
    uint8_t* d_i1 = (uint8_t*)(A + lay.o_i1); uint8_t* d_i2 = (uint8_t*)(A + lay.o_i2);
    uint32_t* d_off = (uint32_t*)(A + lay.o_off); uint32_t* d_f = (uint32_t*)(A + lay.o_f); uint32_t* d_f2 = (uint32_t*)(A + lay.o_f2);
    uint32_t* d_prod = (uint32_t*)(A + lay.o_prod); uint8_t* d_one = (uint8_t*)(A + lay.o_one); uint64_t* d_gt = (uint64_t*)(A + lay.o_gt);
    PAIR_HIP_OK(hipMemcpyAsync(d_off, offsets, (m + 1) * 4, hipMemcpyHostToDevice, stream));
    PAIR_HIP_OK(hipEventRecord(ev[0], stream));
    bool wide = WideProduct<PP>::available && m <= WideProduct<PP>::MAX_PRODUCTS && k >= 1;
    for (size_t p = 0; p < m && wide; p++) wide = offsets[p + 1] - offsets[p] <= WideProduct<PP>::MAX_PAIRS;
    if (wide) {   // few products of few pairs: the latency path (the pieces of a product side by side in several lane groups and waves)
      if (WideProduct<PP>::run(d_g1, has_inf1 ? d_i1 : nullptr, d_g2, has_inf2 ? d_i2 : nullptr, d_off, (uint32_t)m, k, d_f, (uint32_t*)(A + lay.o_wide),
                               out_is_one ? d_one : nullptr, out_gt ? d_gt : nullptr, mode == 0 ? 1 : 0, stream)) return 1;
      PAIR_HIP_OK(hipEventRecord(ev[3], stream));
      if (out_is_one) PAIR_HIP_OK(hipMemcpyAsync(out_is_one, d_one, m, hipMemcpyDeviceToHost, stream));
      if (out_gt) PAIR_HIP_OK(hipMemcpyAsync(out_gt, d_gt, (size_t)m * 72 * 8, hipMemcpyDeviceToHost, stream));
      PAIR_HIP_OK(hipStreamSynchronize(stream));
      PAIR_HIP_OK(hipGetLastError());
      tm = PairingTimings();
      (void)hipEventElapsedTime(&tm.total, ev[0], ev[3]);
      return 0;
    }
    // shared-accumulator mode: one lane group per product (<= 4 pairs each) when the products alone fill the chip
    bool shared = m >= SHARED_MIN_PRODUCTS;
    uint32_t most = 0;
    for (size_t p = 0; p < m && shared; p++) { const uint32_t c = offsets[p + 1] - offsets[p]; shared = c <= 4; most = c > most ? c : most; }
    typedef typename PP::LL LL;
    bool prepared = false, split_done = false;
    // a few thousand products of exactly two pairs (Batch::verify's verdicts): too few to fill the chip with one group per product, and one
    // group per PAIR spends ~900 product rounds per pair; if the first pairs share their G2 point, every product is cut in two by iteration
    // range instead (k_miller_prepared_split_slots: two groups per product, ~710 rounds each)
    bool split_ok = !shared && LL::has_split() && m > WideProduct<PP>::MAX_PRODUCTS && m <= SPLIT_MAX_PRODUCTS && k == 2 * m && miller_split_enabled();
    for (size_t p = 0; p < m && split_ok; p++) split_ok = offsets[p + 1] - offsets[p] == 2;
    if (((shared && most <= 2) || split_ok) && LL::has_prepared() && k) {
      // verify shapes: every product's first pair on the same G2 point (-g2)?  Then its line coefficients are computed once.
      uint32_t* d_flag = (uint32_t*)(A + lay.o_flag);
      uint32_t* d_lines = (uint32_t*)(A + lay.o_lines);
      uint32_t h_flag = 1;
      PAIR_HIP_OK(hipMemcpyAsync(d_flag, &h_flag, 4, hipMemcpyHostToDevice, stream));
      LL::first_q_same(d_g2, d_off, (uint32_t)m, d_flag, stream);
      PAIR_HIP_OK(hipMemcpyAsync(&h_flag, d_flag, 4, hipMemcpyDeviceToHost, stream));
      uint64_t h_q0[PP::G2_ARK64];
      PAIR_HIP_OK(hipMemcpyAsync(h_q0, d_g2 + (size_t)offsets[0] * PP::G2_ARK64, sizeof h_q0, hipMemcpyDeviceToHost, stream));
      PAIR_HIP_OK(hipStreamSynchronize(stream));
      if (h_flag) {
        // the shared point is -g2 in every verify / Batch::verify of the reference: its 69 line triples are kept between calls (what
        // ark-ec's G2Prepared is for a caller that holds one) and k_prepare_lines - 0.7 ms on one wave - runs when the point changes
        const bool hit = lines_valid && lines_at == (const void*)d_lines && memcmp(lines_q0, h_q0, sizeof h_q0) == 0;
        if (!hit) {
          LL::prepare_lines(d_g2 + (size_t)offsets[0] * PP::G2_ARK64, d_lines, stream);
          memcpy(lines_q0, h_q0, sizeof h_q0);
          lines_at = (const void*)d_lines;
          lines_valid = true;
        }
        if (shared) {
          LL::miller_prepared(d_g1, has_inf1 ? d_i1 : nullptr, d_g2, has_inf2 ? d_i2 : nullptr, d_off, d_lines, d_prod, (uint32_t)m, stream);
          prepared = true;
        } else {
          LL::miller_prepared_split(d_g1, has_inf1 ? d_i1 : nullptr, d_g2, has_inf2 ? d_i2 : nullptr, d_lines, d_f, (uint32_t)m, stream);
          split_done = true;      // two values per product at d_f[2 p], d_f[2 p + 1]: the product kernel below multiplies them (offsets are 0, 2, 4, ...)
        }
      }
    }
    if (prepared || split_done) {
    } else if (shared && most <= 2) LL::miller_product2(d_g1, has_inf1 ? d_i1 : nullptr, d_g2, has_inf2 ? d_i2 : nullptr, d_off, d_prod, (uint32_t)m, stream);
    else if (shared) LL::miller_product(d_g1, has_inf1 ? d_i1 : nullptr, d_g2, has_inf2 ? d_i2 : nullptr, d_off, d_prod, (uint32_t)m, stream);
    else if (k) LL::miller(d_g1, has_inf1 ? d_i1 : nullptr, d_g2, has_inf2 ? d_i2 : nullptr, d_f, k, stream);
    PAIR_HIP_OK(hipEventRecord(ev[1], stream));
    if (shared) {
      // products already formed by the Miller kernel
    } else if (m == 1 && k > 8) {  // one large product: pairwise tree, log2(k) levels
      uint32_t n_in = k;
      uint32_t* src = d_f; uint32_t* dst = d_f2;
      while (n_in > 1) {
        LL::gt_tree(src, dst, n_in, stream);
        n_in = (n_in + 1) / 2;
        uint32_t* t = src; src = dst; dst = t;
      }
      PAIR_HIP_OK(hipMemcpyAsync(d_prod, src, W * 4, hipMemcpyDeviceToDevice, stream));
    } else {
      LL::gt_product(d_f, d_off, d_prod, (uint32_t)m, stream);
    }
    PAIR_HIP_OK(hipEventRecord(ev[2], stream));
    LL::final_exp(d_prod, out_is_one ? d_one : nullptr, out_gt ? d_gt : nullptr, (uint32_t)m, mode == 0 ? 1 : 0, stream);
    PAIR_HIP_OK(hipEventRecord(ev[3], stream));
    if (out_is_one) PAIR_HIP_OK(hipMemcpyAsync(out_is_one, d_one, m, hipMemcpyDeviceToHost, stream));
    if (out_gt) PAIR_HIP_OK(hipMemcpyAsync(out_gt, d_gt, (size_t)m * 72 * 8, hipMemcpyDeviceToHost, stream));
    PAIR_HIP_OK(hipStreamSynchronize(stream));
    PAIR_HIP_OK(hipGetLastError());
    (void)hipEventElapsedTime(&tm.miller, ev[0], ev[1]);
    (void)hipEventElapsedTime(&tm.product, ev[1], ev[2]);
    (void)hipEventElapsedTime(&tm.final_exp, ev[2], ev[3]);
    (void)hipEventElapsedTime(&tm.total, ev[0], ev[3]);
    return 0;
  }
  static constexpr size_t SHARED_MIN_PRODUCTS = 16384;
  static constexpr size_t SPLIT_MAX_PRODUCTS = 5120;       // two groups per product, ten per wave: at most one wave per SIMD
  static bool miller_split_enabled() { return true; }      // (the round-4 A/B switch CELO_NO_MILLER_SPLIT is gone: measured, kept on)

 private:
  uint64_t lines_q0[PP::G2_ARK64] = {};     // the G2 point whose prepared lines the arena holds at lines_at (run_staged)
  const void* lines_at = nullptr;
  bool lines_valid = false;
  struct Layout { uint32_t k = 0; size_t m = 0, o_g1 = 0, o_g2 = 0, o_i1 = 0, o_i2 = 0, o_off = 0, o_f = 0, o_f2 = 0, o_prod = 0, o_one = 0, o_gt = 0, o_lines = 0, o_flag = 0, o_wide = 0; } lay;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  OwnedStream stream_;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int ensure(size_t bytes) {
    if (!ev[0])
      for (int i = 0; i < 4; i++) PAIR_HIP_OK(hipEventCreate(&ev[i]));
    if (bytes > arena_bytes) {
      if (arena) (void)hipFree(arena);
      arena = nullptr; arena_bytes = 0;
      lines_valid = false;
      PAIR_HIP_OK(hipMalloc(&arena, bytes));
      arena_bytes = bytes;
    }
    return 0;
  }
};
#endif  // __HIPCC__

}  // namespace celo
