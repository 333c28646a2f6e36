// Host-only build of the product's field/curve templates with run-time bounds tracking
// (-DCELO_FP_TRACK): every mul/sub asserts the limb/value bounds of fp.h's contract.
// Driven by tests/test_host_field.py through ctypes; never shipped.
#include <vector>
#include "curve.h"
#include "gls.h"
#include "fp2.h"
#include "pairing.h"
#include "curve_lanes.h"
#include "wire.h"
#include "hash_direct.h"
#include "host64.h"
#include <cstring>
using namespace celo;

template <class F> static void xyzz_to_jac(const Xyzz<F>& p, uint64_t* out) {
  constexpr int A = F::ARK64;
  if (p.is_identity()) { F::zero().to_ark(out); F::one().to_ark(out + A); F::zero().to_ark(out + 2 * A); return; }
  F::mul(p.X, p.ZZ).to_ark(out);
  F::mul(p.Y, p.ZZZ).to_ark(out + A);
  p.ZZ.to_ark(out + 2 * A);
}
template <class F> static Affine<F> load_aff(const uint64_t* xy) { return {F::from_ark(xy), F::from_ark(xy + F::ARK64)}; }

// op: 0 mul, 1 sqr, 2 add(norm), 3 sub, 4 inv, 5 roundtrip ark->dev->ark, 6 canonical roundtrip
template <class F> static void field_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  F x = F::from_ark(a), y = F::from_ark(b), r;
  switch (op) {
    case 0: r = F::mul(x, y); break;
    case 1: r = F::sqr(x); break;
    case 2: r = F::norm(F::add(x, y)); break;
    case 3: r = F::norm(F::template sub<4, 1>(x, y)); break;
    case 4: r = F::inv(x); break;
    default: r = x; break;
  }
  r.to_ark(out);
}
// chained stress: exercises lazy bounds across many dependent ops like the curve formulas do
template <class F> static void point_op(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) {
  Affine<F> a = load_aff<F>(p1), b = load_aff<F>(p2);
  Xyzz<F> acc = Xyzz<F>::from_affine(a);
  switch (op) {
    case 0: xyzz_madd(acc, b); break;                       // a + b
    case 1: acc = xyzz_dbl(acc); break;                     // 2a
    case 2: { Xyzz<F> t = Xyzz<F>::from_affine(b); t = xyzz_dbl(t); xyzz_madd(t, a); xyzz_add(acc, t); } break;  // a + (2b + a)
    case 3: acc = xyzz_mul_small(acc, k); break;            // k*a
    case 4: xyzz_madd(acc, affine_neg(a)); break;           // a - a = 0
    case 5: xyzz_madd(acc, a); break;                       // a + a via madd (doubling branch)
    case 6: { for (uint32_t i = 0; i < k; i++) xyzz_madd(acc, b); } break;  // a + k*b by repeated madd
    case 7: { Xyzz<F> t = acc; xyzz_add(acc, t); } break;   // add-with-self (doubling branch of add)
    case 8: { for (uint32_t i = 0; i < k; i++) acc = xyzz_dbl(acc); xyzz_madd(acc, b); } break;  // 2^k a + b (a Horner step)
    case 9: acc = xyzz_add_affine(a, b); break;             // a + b, both affine (the first addition of a bucket run)
    case 10: acc = xyzz_add_affine(a, a); break;            // its doubling branch
    case 11: acc = xyzz_add_affine(a, affine_neg(a)); break;  // its cancellation branch
    case 12: { acc = xyzz_add_affine(affine_neg(a), affine_neg(b)); for (uint32_t i = 0; i < k; i++) xyzz_madd(acc, b); } break;  // -a - b + k b
    // 13 .. 17: xyzz_add_mem (the second point read from its stored form where its coordinates are used): t = 2b + a stored, then
    // a + t; the identity stored as zeros; into an identity accumulator; the doubling and the cancellation branch
    case 13: case 14: case 15: case 16: case 17: {
      Xyzz<F> t = Xyzz<F>::from_affine(b); t = xyzz_dbl(t); xyzz_madd(t, a);
      if (op == 14) t = Xyzz<F>::identity();
      if (op == 16) t = acc;                                           // a + a
      if (op == 17) t = Xyzz<F>::from_affine(affine_neg(a));           // a - a
      if (op == 15) acc = Xyzz<F>::identity();                         // 0 + t
      alignas(16) uint32_t buf[4 * F::WORDS];
      t.X.store(buf); t.Y.store(buf + F::WORDS); t.ZZ.store(buf + 2 * F::WORDS); t.ZZZ.store(buf + 3 * F::WORDS);
      xyzz_add_mem(acc, buf);
    } break;
    default: break;
  }
  xyzz_to_jac(acc, out);
}

// lane-parallel XYZZ arithmetic (curve_lanes.h) on the three-explicit-lanes host backend: the same ops as point_op
template <class F> static void lane_point_op(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) {
  typedef QHostT<FieldBase<F>> QB;
  typedef LanePoint<QB> LP;
  Affine<F> a = load_aff<F>(p1), b = load_aff<F>(p2);
  auto lift = [](const Affine<F>& p) { typename LP::P r = {QB::uni(p.x), QB::uni(p.y), QB::uni(F::one()), QB::uni(F::one())}; return r; };
  typename LP::Pt acc = {lift(a), false};
  switch (op) {
    case 0: LP::add(acc, lift(b), false); break;                                            // a + b
    case 1: LP::dbl(acc); break;                                                            // 2a
    case 2: { typename LP::Pt t = {lift(b), false}; LP::dbl(t); LP::add(t, lift(a), false); LP::add(acc, t.p, t.inf); } break;  // a + (2b + a)
    case 4: LP::add(acc, lift(affine_neg(a)), false); break;                                // a - a = 0
    case 6: { for (uint32_t i = 0; i < k; i++) LP::add(acc, lift(b), false); } break;        // a + k*b
    case 7: { typename LP::P t = acc.p; LP::add(acc, t, false); } break;                    // a + a through add (doubling branch)
    case 8: { for (uint32_t i = 0; i < k; i++) LP::dbl(acc); LP::add(acc, lift(b), false); } break;  // 2^k a + b (a Horner step)
    default: break;
  }
  Xyzz<F> r = acc.inf ? Xyzz<F>::identity() : Xyzz<F>{acc.p.X.v[0], acc.p.Y.v[0], acc.p.ZZ.v[0], acc.p.ZZZ.v[0]};
  for (int l = 1; l < 3 && !acc.inf; l++) {   // the lanes must agree
    if (!F::norm(F::template sub<64, 1>(F::norm(acc.p.X.v[l]), F::norm(r.X))).is_zero_mod_p() ||
        !F::norm(F::template sub<64, 1>(F::norm(acc.p.ZZZ.v[l]), F::norm(r.ZZZ))).is_zero_mod_p()) r = Xyzz<F>::identity();
  }
  xyzz_to_jac(r, out);
}

// the same ops for G2 of BLS12-377 on the SIX-lane host backend (halves of an Fq2 in adjacent lanes: msm.h k_batch_horner_hex)
static void lane_point_op_hex(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) {
  typedef Fp2<P377> F;
  typedef QHostHex377 QB;
  typedef LanePoint<QB> LP;
  Affine<F> a = load_aff<F>(p1), b = load_aff<F>(p2);
  auto lift = [](const Affine<F>& p) { LP::P r = {QB::uni2(p.x.c0, p.x.c1), QB::uni2(p.y.c0, p.y.c1), QB::uni2(Fq::one(), Fq::zero()), QB::uni2(Fq::one(), Fq::zero())}; return r; };
  LP::Pt acc = {lift(a), false};
  switch (op) {
    case 0: LP::add(acc, lift(b), false); break;
    case 1: LP::dbl(acc); break;
    case 2: { LP::Pt t = {lift(b), false}; LP::dbl(t); LP::add(t, lift(a), false); LP::add(acc, t.p, t.inf); } break;
    case 4: LP::add(acc, lift(affine_neg(a)), false); break;
    case 6: { for (uint32_t i = 0; i < k; i++) LP::add(acc, lift(b), false); } break;
    case 7: { LP::P t = acc.p; LP::add(acc, t, false); } break;
    case 8: { for (uint32_t i = 0; i < k; i++) LP::dbl(acc); LP::add(acc, lift(b), false); } break;
    default: break;
  }
  auto f2 = [](const QB::V& v, int j) { return F{v.v[2 * j], v.v[2 * j + 1]}; };
  Xyzz<F> r = acc.inf ? Xyzz<F>::identity() : Xyzz<F>{f2(acc.p.X, 0), f2(acc.p.Y, 0), f2(acc.p.ZZ, 0), f2(acc.p.ZZZ, 0)};
  for (int l = 1; l < 3 && !acc.inf; l++) {   // the tower lanes must agree
    if (!F::norm(F::template sub<64, 1>(F::norm(f2(acc.p.X, l)), F::norm(r.X))).is_zero_mod_p() ||
        !F::norm(F::template sub<64, 1>(F::norm(f2(acc.p.ZZZ, l)), F::norm(r.ZZZ))).is_zero_mod_p()) r = Xyzz<F>::identity();
  }
  xyzz_to_jac(r, out);
}

// pairing tower on the host with bounds tracking.  mode 0: product of pairings (GT), 1: Miller-loop product only,
// 2: final exponentiation of the given GT (in72), 3: Fq12 mul (in72 * in72b), 4: Fq12 inverse, 5: cyclotomic square,
// 6: frobenius 1, 7: frobenius 2, 8: frobenius 3, 9: Fq12 square
static void pairing_op(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, const uint64_t* in72, const uint64_t* in72b,
                       uint64_t* out72, int* is_one) {
  Fq12 r;
  if (mode <= 1) {
    Fq12 acc = f12_one();
    for (size_t i = 0; i < k; i++) {
      Fq px = Fq::from_ark(g1 + i * 12), py = Fq::from_ark(g1 + i * 12 + 6);
      Fq2 qx = Fq2::from_ark(g2 + i * 24), qy = Fq2::from_ark(g2 + i * 24 + 12);
      Fq12 f, t;
      miller_loop_single(f, px, py, qx, qy);
      f12_mul(t, acc, f);
      acc = t;
    }
    if (mode == 0) final_exponentiation(r, acc);
    else r = acc;
  } else {
    Fq12 a = f12_from_ark(in72);
    switch (mode) {
      case 2: final_exponentiation(r, a); break;
      case 3: { Fq12 b = f12_from_ark(in72b); f12_mul(r, a, b); } break;
      case 4: f12_inv(r, a); break;
      case 5: f12_cyclotomic_sqr(r, a); break;
      case 6: f12_frob<1>(r, a); break;
      case 7: f12_frob<2>(r, a); break;
      case 8: f12_frob<3>(r, a); break;
      default: f12_sqr(r, a); break;
    }
  }
  f12_to_ark(r, out72);
  if (is_one) *is_one = f12_is_one(r) ? 1 : 0;
}

// the lane-parallel pairing (pairing_lanes.h) on its explicit-lanes host backends - three lanes per pairing (an Fq2 per lane)
// and six (half an Fq2 per lane); same modes as pairing_op
struct HostTri {
  typedef QHost377 QB;
  static QB::V v2(const Fq2& lane0, const Fq2& lane1, const Fq2& lane2) { QB::V r; r.v[0] = lane0; r.v[1] = lane1; r.v[2] = lane2; return r; }
  static QB::F f1(const Fq& x) { QB::F r; for (int l = 0; l < 3; l++) r.v[l] = x; return r; }
  static Fq2 get(const QB::V& v, int j) { return v.v[j]; }
};
struct HostHex {
  typedef QHostHex377 QB;
  static QB::V v2(const Fq2& lane0, const Fq2& lane1, const Fq2& lane2) {
    QB::V r;
    const Fq2* s[3] = {&lane0, &lane1, &lane2};
    for (int j = 0; j < 3; j++) { r.v[2 * j] = s[j]->c0; r.v[2 * j + 1] = s[j]->c1; }
    return r;
  }
  static QB::F f1(const Fq& x) { QB::F r; for (int l = 0; l < 6; l++) r.v[l] = x; return r; }
  static Fq2 get(const QB::V& v, int j) { return {v.v[2 * j], v.v[2 * j + 1]}; }
};
template <class H> static void pairing_op_lanes_t(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, const uint64_t* in72, const uint64_t* in72b,
                                                  uint64_t* out72, int* is_one) {
  typedef typename H::QB QB;
  typedef QTower<QB> HQT;
  typedef QPairing377<QB> HQP;
  auto from_f12 = [](const Fq12& x) {
    typename HQT::E12 e;
    e.a = H::v2(x.c0.c0, x.c0.c1, x.c0.c2);
    e.b = H::v2(x.c1.c0, x.c1.c1, x.c1.c2);
    return e;
  };
  auto to_f12 = [](const typename HQT::E12& e) {
    Fq12 x;
    x.c0.c0 = H::get(e.a, 0); x.c0.c1 = H::get(e.a, 1); x.c0.c2 = H::get(e.a, 2);
    x.c1.c0 = H::get(e.b, 0); x.c1.c1 = H::get(e.b, 1); x.c1.c2 = H::get(e.b, 2);
    return x;
  };
  auto load_pair = [&](size_t i, typename QB::F& px, typename QB::F& py, typename QB::V& Qc) {
    px = H::f1(Fq::from_ark(g1 + i * 12)); py = H::f1(Fq::from_ark(g1 + i * 12 + 6));
    const Fq2 qx = Fq2::from_ark(g2 + i * 24), qy = Fq2::from_ark(g2 + i * 24 + 12);
    Qc = H::v2(qx, qy, qx);                                      // lanes 0, 2: Q.x; lane 1: Q.y
  };
  typename HQT::E12 r;
  if (mode == 10 || mode == 11) {  // whole product in one group with a shared accumulator (k <= 4); 10: GT value, 11: Miller value
    typename QB::F px[4], py[4];
    typename QB::V Qc[4];
    for (size_t i = 0; i < k && i < 4; i++) load_pair(i, px[i], py[i], Qc[i]);
    typename HQT::E12 acc = HQP::template miller_multi<4>((int)k, px, py, Qc);
    r = mode == 10 ? HQP::final_exponentiation(acc) : acc;
  } else if (mode == 13 || mode == 14) {  // exactly two pairs, the lines of every step multiplied first (miller_pair2); 13: GT value, 14: Miller value
    typename QB::F px[2], py[2];
    typename QB::V Qc[2];
    for (size_t i = 0; i < 2; i++) load_pair(i, px[i], py[i], Qc[i]);
    typename HQT::E12 acc = HQP::miller_pair2(px, py, Qc);
    r = mode == 13 ? HQP::final_exponentiation(acc) : acc;
  } else if (mode == 12) {         // the product tree of the GPU engine: Miller values multiplied with each other pairwise (bounds: value x value)
    std::vector<typename HQT::E12> m(k);
    for (size_t i = 0; i < k; i++) {
      typename QB::F px, py;
      typename QB::V Qc;
      load_pair(i, px, py, Qc);
      m[i] = HQP::miller(px, py, Qc);
    }
    for (size_t n = k; n > 1; n = (n + 1) / 2)
      for (size_t t = 0; 2 * t < n; t++) m[t] = 2 * t + 1 < n ? HQT::mul12(m[2 * t], m[2 * t + 1]) : m[2 * t];
    r = HQP::final_exponentiation(m[0]);
  } else if (mode <= 1) {
    typename HQT::E12 acc = HQT::one12();
    for (size_t i = 0; i < k; i++) {
      typename QB::F px, py;
      typename QB::V Qc;
      load_pair(i, px, py, Qc);
      acc = HQT::mul12(acc, HQP::miller(px, py, Qc));
    }
    r = mode == 0 ? HQP::final_exponentiation(acc) : acc;
  } else {
    typename HQT::E12 a = from_f12(f12_from_ark(in72));
    switch (mode) {
      case 2: r = HQP::final_exponentiation(a); break;
      case 3: r = HQT::mul12(a, from_f12(f12_from_ark(in72b))); break;
      case 4: r = HQT::inv12(a); break;
      case 5: r = HQT::cyclotomic_sqr(a); break;
      case 6: r = HQP::template frob12<1>(a); break;
      case 7: r = HQP::template frob12<2>(a); break;
      case 8: r = HQP::template frob12<3>(a); break;
      default: r = HQT::sqr12(a); break;
    }
  }
  f12_to_ark(to_f12(r), out72);
  if (is_one) *is_one = HQT::is_one12(r) ? 1 : 0;
}
static void pairing_op_lanes(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, const uint64_t* in72, const uint64_t* in72b,
                             uint64_t* out72, int* is_one) { pairing_op_lanes_t<HostTri>(mode, g1, g2, k, in72, in72b, out72, is_one); }
static void pairing_op_hex(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, const uint64_t* in72, const uint64_t* in72b,
                           uint64_t* out72, int* is_one) { pairing_op_lanes_t<HostHex>(mode, g1, g2, k, in72, in72b, out72, is_one); }
static void pairing_op_761_lanes(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out72, int* is_one) {
  typedef QTower<QHost761> T6;
  typedef QPairing761<QHost761> P6;
  T6::E12 acc = T6::one12();
  for (size_t i = 0; i < k; i++) {
    QHost761::V px, py, Qc;
    for (int l = 0; l < 3; l++) {
      px.v[l] = Fw::from_ark(g1 + i * 24); py.v[l] = Fw::from_ark(g1 + i * 24 + 12);
      Qc.v[l] = Fw::from_ark(g2 + i * 24 + (l & 1) * 12);
    }
    acc = T6::mul12(acc, P6::miller(px, py, Qc));
  }
  T6::E12 r = mode == 0 ? P6::final_exponentiation(acc) : acc;
  Fw6 x;
  Fw* c[6] = {&x.c0.c0, &x.c0.c1, &x.c0.c2, &x.c1.c0, &x.c1.c1, &x.c1.c2};
  for (int j = 0; j < 3; j++) { *c[j] = r.a.v[j]; *c[3 + j] = r.b.v[j]; }
  QuadIO<Base761>::to_ark(x, out72);
  if (is_one) *is_one = T6::is_one12(r) ? 1 : 0;
}

static void pairing_op_761(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out72, int* is_one) {
  Fw6 acc = quad_one<Base761>();
  for (size_t i = 0; i < k; i++) {
    Fw6 f, t;
    { Fw px = Fw::from_ark(g1 + i * 24), py = Fw::from_ark(g1 + i * 24 + 12), qx = Fw::from_ark(g2 + i * 24), qy = Fw::from_ark(g2 + i * 24 + 12);
      bw6_miller_loop_single(f, px, py, qx, qy); }
    quad_mul(t, acc, f);
    acc = t;
  }
  Fw6 r;
  if (mode == 0) bw6_final_exponentiation(r, acc);
  else r = acc;
  QuadIO<Base761>::to_ark(r, out72);
  if (is_one) *is_one = quad_is_one(r) ? 1 : 0;
}

extern "C" int celo_ifma_available();
extern "C" int celo_ifma_horner_377(const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out, int* inf);
extern "C" int celo_ifma_horner_761(const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out, int* inf);
extern "C" {
// wire.h under bounds tracking: n compressed points -> affine ark limbs + status (the GPU kernels run the same functions)
void ht_wire_decode(int g2, const uint8_t* in, size_t n, int check, uint64_t* out, uint8_t* status) {
  const WireConsts& k = wire_consts();
  for (size_t i = 0; i < n; i++) {
    uint64_t* o = out + i * (g2 ? 24 : 12);
    memset(o, 0, (g2 ? 24 : 12) * 8);
    if (g2) {
      Affine<Fq2> p = {Fq2::zero(), Fq2::zero()};
      status[i] = wire_decode_g2(in + i * 96, k, check != 0, p);
      if (status[i] == WIRE_OK) { p.x.c0.to_ark(o); p.x.c1.to_ark(o + 6); p.y.c0.to_ark(o + 12); p.y.c1.to_ark(o + 18); }
    } else {
      Affine<Fq> p = {Fq::zero(), Fq::zero()};
      status[i] = wire_decode_g1(in + i * 48, k, check != 0, p);
      if (status[i] == WIRE_OK) { p.x.to_ark(o); p.y.to_ark(o + 6); }
    }
  }
}
// the two forms of the subgroup test on one affine on-curve point (ark limbs): bit 0 = endomorphism form (what the decoders run),
// bit 1 = the r * P ladder (the reference's definition)
int ht_wire_subgroup_both(int g2, const uint64_t* xy) {
  const WireConsts& k = wire_consts();
  if (g2) {
    const Affine<Fq2> p = {{Fq::from_ark(xy), Fq::from_ark(xy + 6)}, {Fq::from_ark(xy + 12), Fq::from_ark(xy + 18)}};
    const Affine<Fq2> q = {Fq2::norm(p.x), Fq2::norm(p.y)};
    return (wire_in_subgroup(q, k) ? 1 : 0) | (wire_in_subgroup_ladder(q, k) ? 2 : 0);
  }
  const Affine<Fq> q = {Fq::norm(Fq::from_ark(xy)), Fq::norm(Fq::from_ark(xy + 6))};
  return (wire_in_subgroup(q, k) ? 1 : 0) | (wire_in_subgroup_ladder(q, k) ? 2 : 0);
}
// hash_direct.h under bounds tracking: one try-and-increment hash; returns the attempt counter, -1 when none succeeds
int ht_hash_to_g1_direct(const uint8_t* dom, const uint8_t* msg, size_t mlen, const uint8_t* extra, size_t elen, uint64_t* out_xy) {
  Affine<Fq> p = {Fq::zero(), Fq::zero()};
  int c = -1;
  if (!hash_to_g1_direct_tai(dom, msg, mlen, extra, elen, wire_consts(), p, c)) return -1;
  p.x.to_ark(out_xy);
  p.y.to_ark(out_xy + 6);
  return c;
}
// table-driven root vs the Tonelli-Shanks loop on one Fq element (ark limbs): returns 1 + 2*(roots agree up to sign) when a root
// exists for both, 0 when both say non-residue, -1 on disagreement
int ht_wire_fq_sqrt_both(const uint64_t* a, uint64_t* out) {
  const Fq x = Fq::from_ark(a);
  Fq r1, r2;
  const bool o1 = wire_fq_sqrt(x, wire_consts(), r1), o2 = wire_fq_sqrt_ts(x, wire_consts(), r2);
  if (o1 != o2) return -1;
  if (!o1) return 0;
  r1.to_ark(out);
  const bool same = wire_eq(r1, r2) || wire_eq(r1, wire_neg(r2));
  return wire_eq(Fq::sqr(r1), x) && same ? 3 : -1;
}
// square root in Fq2 (ark limbs in and out); returns 1 when a root exists
int ht_wire_fq2_sqrt(const uint64_t* a, uint64_t* out) {
  Fq2 r;
  const Fq2 x = {Fq::from_ark(a), Fq::from_ark(a + 6)};
  if (!wire_fq2_sqrt(x, wire_consts(), r)) return 0;
  r.c0.to_ark(out);
  r.c1.to_ark(out + 6);
  return 1;
}
void ht_pairing_761_lanes(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out72, int* is_one) { pairing_op_761_lanes(mode, g1, g2, k, out72, is_one); }
void ht_pairing_761(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out72, int* is_one) { pairing_op_761(mode, g1, g2, k, out72, is_one); }
void ht_pairing_377(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, const uint64_t* in72, const uint64_t* in72b,
                    uint64_t* out72, int* is_one) { pairing_op(mode, g1, g2, k, in72, in72b, out72, is_one); }
void ht_pairing_377_lanes(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, const uint64_t* in72, const uint64_t* in72b,
                         uint64_t* out72, int* is_one) { pairing_op_lanes(mode, g1, g2, k, in72, in72b, out72, is_one); }
// modular inversion: the safegcd routine on plain integers (field 0: BLS12-377 Fq, 1: BW6-761 Fq), and Fp::inv against Fermat
void ht_modinv(int field, const uint64_t* x, uint64_t* out) {
  if (field == 0) SafeGcd<P377>::inv(x, out); else SafeGcd<P761>::inv(x, out);
}
int ht_inv_matches_fermat(int field, const uint64_t* ark) {
  if (field == 0) {
    const Fq a = Fq::from_ark(ark);
    return Fq::eq_mod_p(Fq::norm(Fq::inv(a)), Fq::norm(Fq::inv_fermat(a))) ? 1 : 0;
  }
  const Fw a = Fw::from_ark(ark);
  return Fw::eq_mod_p(Fw::norm(Fw::inv(a)), Fw::norm(Fw::inv_fermat(a))) ? 1 : 0;
}
void ht_pairing_377_hex(int mode, const uint64_t* g1, const uint64_t* g2, size_t k, const uint64_t* in72, const uint64_t* in72b,
                       uint64_t* out72, int* is_one) { pairing_op_hex(mode, g1, g2, k, in72, in72b, out72, is_one); }
void ht_fq377(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) { field_op<Fp<P377>>(op, a, b, out); }
void ht_fq761(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) { field_op<Fp<P761>>(op, a, b, out); }
// Fr(BLS12-377), the 10-limb field of the hash-helper proof's NTT (ntt.h); op 7 = weak reduction of a lazily grown sum (the butterfly's
// x + y path), op 8 = canonical-integer round trip
void ht_fr377(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  typedef Fp<P253> F;
  if (op == 7) { F x = F::from_ark(a), y = F::from_ark(b); F t = F::add(F::add(x, y), F::add(x, y)); t = F::add(t, t); F::wred(F::add(t, x)).to_ark(out); return; }   // 5x + 4y, value bound ~ 27p
  if (op == 8) { F x = F::from_canonical(a); x.to_canonical(out); return; }
  field_op<F>(op, a, b, out);
}
void ht_fq2_377(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) { field_op<Fp2<P377>>(op, a, b, out); }
void ht_g1_377(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { point_op<Fp<P377>>(op, p1, p2, k, out); }
void ht_g2_377(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { point_op<Fp2<P377>>(op, p1, p2, k, out); }
void ht_lane_g1_377(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { lane_point_op<Fp<P377>>(op, p1, p2, k, out); }
void ht_lane_g2_377(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { lane_point_op<Fp2<P377>>(op, p1, p2, k, out); }
void ht_lane_g2_377_hex(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { lane_point_op_hex(op, p1, p2, k, out); }
void ht_lane_g_761(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { lane_point_op<Fp<P761>>(op, p1, p2, k, out); }
void ht_g_761(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { point_op<Fp<P761>>(op, p1, p2, k, out); }
// Horner step lists (host64.h) on the 64-bit-limb path and on the IFMA path (host_ifma.cpp), field 0: BLS12-377 Fq, 1: BW6-761 Fq.
// out: X, Y, ZZ, ZZZ (arkworks form; all zero = identity).  ht_horner_ifma returns -1 without AVX-512 IFMA, 1 on a special case.
void ht_horner64(int field, const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out) {
  if (field == 0) { auto r = host64_horner<HFp<P377>>(pts, stride, 6, order, steps); r.X.store(out); r.Y.store(out + 6); r.ZZ.store(out + 12); r.ZZZ.store(out + 18); }
  else { auto r = host64_horner<HFp<P761>>(pts, stride, 12, order, steps); r.X.store(out); r.Y.store(out + 12); r.ZZ.store(out + 24); r.ZZZ.store(out + 36); }
}
int ht_horner_ifma(int field, const uint64_t* pts, size_t stride, const int32_t* order, int steps, uint64_t* out) {
  if (!celo_ifma_available()) return -1;
  int inf = 0;
  return field == 0 ? celo_ifma_horner_377(pts, stride, order, steps, out, &inf) : celo_ifma_horner_761(pts, stride, order, steps, out, &inf);
}
// base-x digits (gls.h) of a scalar of nw significant words into nd digits: d[4][2] words; the (nw, nd) pairs the batched MSM uses
int ht_gls_digits(int nw, int nd, const uint32_t* k8, uint32_t* d8) {
  uint32_t d[4][2];
  switch (nw * 10 + nd) {
    case 32: gls_digits_base_x<3, 2>(k8, d); break;
    case 42: gls_digits_base_x<4, 2>(k8, d); break;
    case 43: gls_digits_base_x<4, 3>(k8, d); break;
    case 53: gls_digits_base_x<5, 3>(k8, d); break;
    case 63: gls_digits_base_x<6, 3>(k8, d); break;
    case 64: gls_digits_base_x<6, 4>(k8, d); break;
    case 74: gls_digits_base_x<7, 4>(k8, d); break;
    case 84: gls_digits_base_x<8, 4>(k8, d); break;
    default: return 1;
  }
  for (int j = 0; j < 4; j++) { d8[2 * j] = d[j][0]; d8[2 * j + 1] = d[j][1]; }
  return 0;
}
void ht_glv_split(const uint32_t* k8, uint32_t* k0, uint32_t* k1) { glv_split_x2<8>(k8, k0, k1); }
void ht_fq377_canon(const uint64_t* canon, uint64_t* out_ark, uint64_t* out_canon) {
  Fp<P377> x = Fp<P377>::from_canonical(canon);
  x.to_ark(out_ark);
  x.to_canonical(out_canon);
}
// xyzz_add_affine on RAW device limbs (14 x 28 bits per coordinate, not necessarily on the curve: the formulas are rational functions
// of the coordinates): -(x1, y1) + (x2, y2) through the affine + affine start and through from_affine + xyzz_madd, both exported as
// Jacobian ark limbs.  The top-limb edge of ADVICE r3 (a negated y whose top limb is 4 p's own against a y whose top limb is zero)
// cannot be reached through from_ark, whose products pick the representative.
void ht_add_affine_raw_377(const uint32_t* x1, const uint32_t* y1, const uint32_t* x2, const uint32_t* y2, uint64_t* out_start, uint64_t* out_madd,
                           uint32_t* neg_y_top) {
  typedef Fp<P377> F;
  Affine<F> p = affine_neg(Affine<F>{F::from_limbs(x1), F::from_limbs(y1)}), q = {F::from_limbs(x2), F::from_limbs(y2)};
  *neg_y_top = p.y.l[F::L - 1];
  Xyzz<F> a = xyzz_add_affine(p, q);
  Xyzz<F> b = Xyzz<F>::from_affine(p);
  xyzz_madd(b, q);
  xyzz_to_jac(a, out_start);
  xyzz_to_jac(b, out_madd);
}
}
