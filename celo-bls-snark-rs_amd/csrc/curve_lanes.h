// XYZZ point doubling and addition spread over a group of three lanes (backends of lanes.h): the same formulas as curve.h
// (dbl-2008-s-1, add-2008-s), their independent field products issued side by side - a doubling is 4 dependent product
// rounds instead of 9 products, an addition 5 instead of 14.  For chains that are pure latency: the per-instance Horner of
// the batched MSM (msm.h k_batch_horner_lanes: 136 dependent doublings per Batch::verify instance, crates/bls-crypto/src/
// bls/batch.rs:69,76), where one lane per instance leaves the GPU idle and each doubling costs its full serial latency.
// All lanes of a group hold the whole point (X, Y, ZZ, ZZZ); each computes the product assigned to it, the results are
// broadcast inside the group.  Same bounds contract as curve.h: stored coordinates normalised, vb(X) <= 19, vb(Y) <= 7,
// vb(ZZ), vb(ZZZ) <= 3.
#pragma once
#include "curve.h"
#include "lanes.h"

namespace celo {

template <class QB> struct LanePoint {
  typedef typename QB::V V;
  struct P { V X, Y, ZZ, ZZZ; };   // group-uniform

  // 2a (a finite with Y != 0 is the caller's responsibility: see dbl() below for the checked entry)
  QFN static P dbl_nz(const P& a) {
    V U = QB::dbl(a.Y);                                                           // vb <= 14
    V r1 = QB::mul(QB::pick(U, a.X, U), QB::pick(U, a.X, U));                     // U^2, X^2
    V Vv = QB::template bcast<0>(r1);
    V M = QB::tpl(QB::template bcast<1>(r1));                                     // 3 X^2, vb <= 9
    V r2 = QB::mul(QB::pick(U, a.X, Vv), QB::pick(Vv, Vv, a.ZZ));                 // W = U V, S = X V, ZZ3 = V ZZ
    V W = QB::template bcast<0>(r2), S = QB::template bcast<1>(r2);
    V r3 = QB::mul(QB::pick(M, W, W), QB::pick(M, a.Y, a.ZZZ));                   // M^2, W Y, ZZZ3 = W ZZZ
    V X3 = QB::template sub<16>(QB::template bcast<0>(r3), QB::dbl(S));           // vb <= 19
    V t = QB::template sub<32>(S, X3);                                            // vb <= 35
    V Y3 = QB::template sub<4>(QB::mul(M, t), QB::template bcast<1>(r3));         // every lane: M (S - X3) - W Y
    return {X3, Y3, QB::template bcast<2>(r2), QB::template bcast<2>(r3)};
  }
  // a + b for finite a, b with a != +-b (the caller checks Pd, R below through add())
  struct AddMid { V U1, S1, Pd, R, ZZ12, ZZZ12; };
  QFN static AddMid add_mid(const P& a, const P& b) {
    V r1 = QB::mul(QB::pick(a.X, b.X, a.Y), QB::pick(b.ZZ, a.ZZ, b.ZZZ));         // U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2
    V r2 = QB::mul(QB::pick(b.Y, a.ZZ, a.ZZZ), QB::pick(a.ZZZ, b.ZZ, b.ZZZ));     // S2 = Y2 ZZZ1, ZZ1 ZZ2, ZZZ1 ZZZ2
    AddMid m;
    m.U1 = QB::template bcast<0>(r1);
    m.S1 = QB::template bcast<2>(r1);
    m.Pd = QB::template sub<4>(QB::template bcast<1>(r1), m.U1);                  // vb <= 7
    m.R = QB::template sub<4>(QB::template bcast<0>(r2), m.S1);
    m.ZZ12 = QB::template bcast<1>(r2);
    m.ZZZ12 = QB::template bcast<2>(r2);
    return m;
  }
  QFN static P add_finish(const AddMid& m) {
    V r3 = QB::mul(QB::pick(m.Pd, m.R, m.Pd), QB::pick(m.Pd, m.R, m.Pd));        // PP = P^2, R^2
    V PP = QB::template bcast<0>(r3), R2 = QB::template bcast<1>(r3);
    V r4 = QB::mul(QB::pick(m.Pd, m.U1, m.ZZ12), PP);                             // PPP = P PP, Q = U1 PP, ZZ3 = ZZ1 ZZ2 PP
    V PPP = QB::template bcast<0>(r4), Q = QB::template bcast<1>(r4);
    V X3 = QB::template sub<16>(R2, QB::add(PPP, QB::dbl(Q)));                    // PPP + 2Q: vb <= 9; X3 vb <= 19
    V t = QB::template sub<32>(Q, X3);
    V r5 = QB::mul(QB::pick(m.R, m.S1, m.ZZZ12), QB::pick(t, PPP, PPP));          // R (Q - X3), S1 PPP, ZZZ3 = ZZZ1 ZZZ2 PPP
    V Y3 = QB::template sub<4>(QB::template bcast<0>(r5), QB::template bcast<1>(r5));
    return {X3, Y3, QB::template bcast<2>(r4), QB::template bcast<2>(r5)};
  }
  // the checked entry points (identity, P + P, P - P), decided per group: the point is group-uniform
  struct Pt { P p; bool inf; };
  QFN static void dbl(Pt& a) {
    if (a.inf) return;
    if (QB::is_zero_u(a.p.Y)) { a.inf = true; return; }
    a.p = dbl_nz(a.p);
  }
  QFN static void add(Pt& a, const P& b, bool b_inf) {
    if (b_inf) return;
    if (a.inf) { a.p = b; a.inf = false; return; }
    const AddMid m = add_mid(a.p, b);
    if (QB::is_zero_u(m.Pd)) {
      if (QB::is_zero_u(m.R)) dbl(a);
      else a.inf = true;
      return;
    }
    a.p = add_finish(m);
  }
};

}  // namespace celo
