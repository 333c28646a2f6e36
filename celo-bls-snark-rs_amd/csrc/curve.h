// Short-Weierstrass (a = 0) point arithmetic in extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), generic over the coordinate field F
// (Fp<P377> for BLS12-377 G1, Fp2 for G2, Fp<P761> for both BW6-761 groups).
//
// Replaces ark-ec's GroupProjective::{add_assign_mixed, add_assign, double_in_place}
//  underneath VariableBaseMSM (crates/bls-crypto/src/bls/signature.rs:85,
// public.rs:61).  The reference uses Jacobian madd-2007-bl (7M+4S); on the GPU the bucket
// accumulators use XYZZ (madd-2008-s: 8M+2S, no field additions on the critical multiplication
// inputs and cheaper general adds for the bucket reduction).  Group elements are unique, so the
// affine-normalised result is bit-identical to the reference's.
//
// Every formula is annotated with the [lb, vb] bounds of fp.h's contract.  Operand contract of the products: stored coordinates,
// loaded points and product outputs are normalised (lb 1); the few loosely reduced values a formula multiplies (the differences
// Pd, R, the doubled Y, 3 X^2 and t) go through F::prep() once - a carry pass for the fields whose products need normalised
// operands (Fp2, the 28-limb field), nothing for the 14-limb field - and the products are the _nn forms that do not carry again.
#pragma once
#include "fp.h"

namespace celo {

template <class F> struct Affine {
  F x, y;  // normalised limbs (lb<=1), vb<=2
};

template <class F> struct Xyzz {
  F X, Y, ZZ, ZZZ;  // stored normalised: lb 1; vb(X) <= 19, vb(Y) <= 7, vb(ZZ), vb(ZZZ) <= 3 (Fp2 products are < 3p)
  HD static Xyzz identity() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  HD bool is_identity() const { return ZZ.limbs_all_zero(); }
  HD static Xyzz from_affine(const Affine<F>& p) { return {p.x, p.y, F::one(), F::one()}; }
};

// 2*(x,y) for an affine point (mdbl-2008-s-1, a = 0).  y == 0 cannot occur on prime-order-cofactor
// curves' r-torsion; handled anyway (result identity).
template <class F> HD Xyzz<F> xyzz_dbl_affine(const Affine<F>& p) {
  if (p.y.is_zero_mod_p()) return Xyzz<F>::identity();
  F U = F::prep(F::dbl(p.y));                         // [2, 4]   (prep / _nn: operands carried once, see Fp::prep)
  F V = F::sqr_nn(U);                                 // [1, 2]
  F W = F::mul_nn(U, V);                              // [1, 2]
  F S = F::mul_nn(p.x, V);                            // [1, 2]
  F xx = F::sqr_nn(p.x);
  F M = F::prep(F::add(F::add(xx, xx), xx));          // [3, 6]
  F M2 = F::sqr_nn(M);                                // 9 ok
  F X3 = F::norm(F::template sub<16, 3>(M2, F::dbl(S)));   // [1, 10]
  F t = F::prep(F::template sub<32, 1>(S, X3));       // [3, 18]
  F Y3 = F::template mul_sub_nn_at<0>(M, t, W, p.y);  // [1, <=7]   (the _at<site> tag only matters to the signed-pass reproducer builds: fp2.h)
  return {X3, Y3, V, W};
}

template <class F> HD Xyzz<F> xyzz_dbl(const Xyzz<F>& a) {
  if (a.is_identity()) return a;
  if (a.Y.is_zero_mod_p()) return Xyzz<F>::identity();
  F U = F::prep(F::dbl(a.Y));                         // [2, 12]
  F V = F::sqr_nn(U);
  F W = F::mul_nn(U, V);
  F S = F::mul_nn(a.X, V);
  F xx = F::sqr_nn(a.X);
  F M = F::prep(F::add(F::add(xx, xx), xx));          // [3, 6]
  F M2 = F::sqr_nn(M);
  F X3 = F::norm(F::template sub<16, 3>(M2, F::dbl(S)));
  F t = F::prep(F::template sub<32, 1>(S, X3));
  F Y3 = F::template mul_sub_nn_at<1>(M, t, W, a.Y);
  return {X3, Y3, F::mul_nn(V, a.ZZ), F::mul_nn(W, a.ZZZ)};
}

// acc += (x2, y2) affine (madd-2008-s).  acc may be the identity.
template <class F> HD void xyzz_madd(Xyzz<F>& a, const Affine<F>& p) {
  if (a.is_identity()) { a = Xyzz<F>::from_affine(p); return; }
  F U2 = F::mul_nn(p.x, a.ZZ);                        // [1, 2]
#ifdef CELO_MADD_R_LATE
  // reproducer builds only (tools/repro_acc/REPORT.md): S2 and R computed AFTER the exact-zero test of Pd instead of before it - which values are
  // live across that test's cold canonical reduction is the variable of the experiment
  F Pd = F::prep(F::template sub<32, 1>(U2, a.X));    // [3, 18]
  if (Pd.is_zero_mod_p()) {
    F R0 = F::prep(F::template sub<16, 1>(F::mul_nn(p.y, a.ZZZ), a.Y));
    if (R0.is_zero_mod_p()) a = xyzz_dbl_affine(p);
    else a = Xyzz<F>::identity();
    return;
  }
  F S2 = F::mul_nn(p.y, a.ZZZ);
  F R = F::prep(F::template sub<16, 1>(S2, a.Y));     // [3, 18]
#else
  F S2 = F::mul_nn(p.y, a.ZZZ);
  F Pd = F::prep(F::template sub<32, 1>(U2, a.X));    // [3, 18]
  F R = F::prep(F::template sub<16, 1>(S2, a.Y));     // [3, 18]
  if (Pd.is_zero_mod_p()) {
    if (R.is_zero_mod_p()) a = xyzz_dbl_affine(p);
    else a = Xyzz<F>::identity();
    return;
  }
#endif
  F PP = F::sqr_nn(Pd);                               // 9 ok -> [1, 2]
  F PPP = F::mul_nn(Pd, PP);
  F Q = F::mul_nn(a.X, PP);
  F R2 = F::sqr_nn(R);
  F s = F::add(F::add(PPP, Q), Q);                    // [3, 6]
  F X3 = F::norm(F::template sub<16, 3>(R2, s));       // [1, 10]
  F t = F::prep(F::template sub<32, 1>(Q, X3));       // [3, 18]
  F Y3 = F::template mul_sub_nn_at<2>(R, t, a.Y, PPP); // R*t - Y1*PPP, one reduction pass: [1, <=7]
  a.ZZ = F::mul_nn(a.ZZ, PP);
  a.ZZZ = F::mul_nn(a.ZZZ, PPP);
  a.X = X3;
  a.Y = Y3;
}

// p + q for two AFFINE points, result XYZZ (mmadd-2007-bl's shape: U2 = x2, S2 = y2, ZZ3 = PP, ZZZ3 = PPP): 4 products + 2 squares
// where xyzz_madd on from_affine(p) spends 8 + 2.  The first addition of every bucket run (k_accumulate).  x < 2 p (stored
// bases); y < 2 p, or in (2 p, 4 p) after affine_neg - which is why the y difference takes 8 p of slack: a subtrahend of up to K p
// exactly can carry the top limb of K p itself, one more than the redundant form's (the low limbs borrowed from it), and
// sub<4, 1> then wraps the top limb of the difference (a negated y below 0.57 * 2^364 against a y below 2^364;
// tests/test_host_field.py::test_add_affine_negated_y_top_limb).  sub<K, M> needs its subtrahend strictly below K p - p.
template <class F> HD Xyzz<F> xyzz_add_affine(const Affine<F>& p, const Affine<F>& q) {
  F Pd = F::prep(F::template sub<4, 1>(q.x, p.x));    // [3, <= 8]
  F R = F::prep(F::template sub<8, 1>(q.y, p.y));     // [3, <= 12]
  if (Pd.is_zero_mod_p()) {
    if (R.is_zero_mod_p()) return xyzz_dbl_affine(p);
    return Xyzz<F>::identity();
  }
  F PP = F::sqr_nn(Pd);
  F PPP = F::mul_nn(Pd, PP);
  F Q = F::mul_nn(p.x, PP);
  F R2 = F::sqr_nn(R);
  F s = F::add(F::add(PPP, Q), Q);
  F X3 = F::norm(F::template sub<16, 3>(R2, s));
  F t = F::prep(F::template sub<32, 1>(Q, X3));
  F Y3 = F::template mul_sub_nn_at<3>(R, t, p.y, PPP);
  return {X3, Y3, PP, PPP};
}

// a += b (add-2008-s), both XYZZ, either may be the identity
template <class F> HD void xyzz_add(Xyzz<F>& a, const Xyzz<F>& b) {
  if (b.is_identity()) return;
  if (a.is_identity()) { a = b; return; }
  F U1 = F::mul_nn(a.X, b.ZZ);
  F U2 = F::mul_nn(b.X, a.ZZ);
  F S1 = F::mul_nn(a.Y, b.ZZZ);
  F S2 = F::mul_nn(b.Y, a.ZZZ);
  F Pd = F::prep(F::template sub<4, 1>(U2, U1));      // [3, 6]
  F R = F::prep(F::template sub<4, 1>(S2, S1));
  if (Pd.is_zero_mod_p()) {
    if (R.is_zero_mod_p()) a = xyzz_dbl<F>(a);
    else a = Xyzz<F>::identity();
    return;
  }
  F PP = F::sqr_nn(Pd);
  F PPP = F::mul_nn(Pd, PP);
  F Q = F::mul_nn(U1, PP);
  F R2 = F::sqr_nn(R);
  F s = F::add(F::add(PPP, Q), Q);
  F X3 = F::norm(F::template sub<16, 3>(R2, s));
  F t = F::prep(F::template sub<32, 1>(Q, X3));
  F Y3 = F::template mul_sub_nn_at<4>(R, t, S1, PPP);
  a.ZZ = F::mul_nn(F::mul_nn(a.ZZ, b.ZZ), PP);
  a.ZZZ = F::mul_nn(F::mul_nn(a.ZZZ, b.ZZZ), PPP);
  a.X = X3;
  a.Y = Y3;
}

// a += the XYZZ point stored at src (X | Y | ZZ | ZZZ, F::WORDS words apart; the identity stored as exact zeros), its coordinates
// loaded WHERE THEY ARE USED instead of up front (round 4).  A full addition of Fq2 points holds the accumulator (112 registers), the
// second point (112) and the addition's temporaries; with the point loaded first the one-lane kernels of the batched path spilled to
// scratch (k_batch_reduce<G2_377>: 1264 B/lane, k_batch_bitsums: 180 B and 233 scratch reloads per addition, each waited for by a lone
// wave: 230 us per addition against 24 us per mixed addition in k_accumulate).  Memory is the spill space: ZZ2 and ZZZ2 are read twice.
template <class F> HD void xyzz_add_mem(Xyzz<F>& a, const uint32_t* src) {
  constexpr int FW = F::WORDS;
#define XAM_FENCE() asm volatile("" ::: "memory")
  {
    const F bzz = F::load(src + 2 * FW);
    if (bzz.limbs_all_zero()) return;
    if (a.is_identity()) { a = {F::load(src), F::load(src + FW), bzz, F::load(src + 3 * FW)}; return; }
  }
  XAM_FENCE();
  F Pd, R, U1, S1;
  {
    U1 = F::mul_nn(a.X, F::load(src + 2 * FW));
    const F U2 = F::mul_nn(F::load(src), a.ZZ);
    Pd = F::prep(F::template sub<4, 1>(U2, U1));      // [3, 6]
  }
  XAM_FENCE();
  {
    S1 = F::mul_nn(a.Y, F::load(src + 3 * FW));
    const F S2 = F::mul_nn(F::load(src + FW), a.ZZZ);
    R = F::prep(F::template sub<4, 1>(S2, S1));
  }
  XAM_FENCE();
  if (Pd.is_zero_mod_p()) {
    if (R.is_zero_mod_p()) a = xyzz_dbl<F>(a);
    else a = Xyzz<F>::identity();
    return;
  }
  F PP = F::sqr_nn(Pd);
  F PPP = F::mul_nn(Pd, PP);
  F Q = F::mul_nn(U1, PP);
  F R2 = F::sqr_nn(R);
  F s = F::add(F::add(PPP, Q), Q);
  F X3 = F::norm(F::template sub<16, 3>(R2, s));
  F t = F::prep(F::template sub<32, 1>(Q, X3));
  F Y3 = F::template mul_sub_nn_at<4>(R, t, S1, PPP);
  XAM_FENCE();
  a.ZZ = F::mul_nn(F::mul_nn(a.ZZ, F::load(src + 2 * FW)), PP);
  XAM_FENCE();
  a.ZZZ = F::mul_nn(F::mul_nn(a.ZZZ, F::load(src + 3 * FW)), PPP);
  a.X = X3;
  a.Y = Y3;
#undef XAM_FENCE
}

// Out-of-line variants for the latency-bound reduction kernels (one code copy per translation unit instead of one per
// call site: the inlined bodies are ~8-17k instructions each and dominated the build time).
#if defined(__HIPCC__)
#define CURVE_FN __host__ __device__ inline __attribute__((noinline))
#else
#define CURVE_FN inline
#endif
template <class F> CURVE_FN void xyzz_add_outline(Xyzz<F>& a, const Xyzz<F>& b) { xyzz_add<F>(a, b); }
template <class F> CURVE_FN void xyzz_dbl_outline(Xyzz<F>& a) { a = xyzz_dbl<F>(a); }
// 14-limb fields (BLS12-377 G1, the headline path) keep the inlined bodies: out-of-line calls pass the points through
// private memory and doubled the per-addition latency of the (latency-bound) reduction kernels.
template <class F> HD void xyzz_add_fn(Xyzz<F>& a, const Xyzz<F>& b) {
  if constexpr (sizeof(F) <= 64) xyzz_add(a, b);
  else xyzz_add_outline(a, b);
}
template <class F> HD void xyzz_dbl_fn(Xyzz<F>& a) {
  if constexpr (sizeof(F) <= 64) a = xyzz_dbl(a);
  else xyzz_dbl_outline(a);
}

template <class F> HD Affine<F> affine_neg(const Affine<F>& p) {
  return {p.x, F::norm(F::template neg<4, 1>(p.y))};
}

// k * a for a small non-negative k (bucket-reduction fix-ups), MSB-first double-and-add
template <class F> HD Xyzz<F> xyzz_mul_small(const Xyzz<F>& a, uint32_t k) {
  Xyzz<F> r = Xyzz<F>::identity();
  if (k == 0) return r;
  int top = 31;
  while (!((k >> top) & 1)) top--;
  r = a;
  for (int i = top - 1; i >= 0; i--) {
    xyzz_dbl_fn(r);
    if ((k >> i) & 1) xyzz_add_fn(r, a);
  }
  return r;
}

}  // namespace celo
