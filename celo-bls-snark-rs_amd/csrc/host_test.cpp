// Host-only build of the product's field/curve templates with run-time bounds tracking
// (-DCELO_FP_TRACK): every mul/sub asserts the limb/value bounds of fp.h's contract.
// Driven by tests/test_host_field.py through ctypes; never shipped.
#include "curve.h"
#include "fp2.h"
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
    default: break;
  }
  xyzz_to_jac(acc, out);
}

extern "C" {
void ht_fq377(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) { field_op<Fp<P377>>(op, a, b, out); }
void ht_fq761(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) { field_op<Fp<P761>>(op, a, b, out); }
void ht_fq2_377(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) { field_op<Fp2<P377>>(op, a, b, out); }
void ht_g1_377(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { point_op<Fp<P377>>(op, p1, p2, k, out); }
void ht_g2_377(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { point_op<Fp2<P377>>(op, p1, p2, k, out); }
void ht_g_761(int op, const uint64_t* p1, const uint64_t* p2, uint32_t k, uint64_t* out) { point_op<Fp<P761>>(op, p1, p2, k, out); }
void ht_fq377_canon(const uint64_t* canon, uint64_t* out_ark, uint64_t* out_canon) {
  Fp<P377> x = Fp<P377>::from_canonical(canon);
  x.to_ark(out_ark);
  x.to_canonical(out_canon);
}
}
