// Lane-group backends shared by the lane-parallel pairings (pairing_lanes.h) and the lane-parallel point arithmetic of the
// batched MSM (curve_lanes.h): a value type V = "one base-field element per lane of a group of three adjacent lanes" with
// elementwise arithmetic, permutations inside the group and per-lane selection.  QTriT is the device backend (ds_bpermute_b32,
// 21 groups per wave64), QHostT the host one (three explicit lanes) so that the same algorithm templates run under the
// bounds-tracking host build and against the oracle.
#pragma once
#include "fp2.h"

namespace celo {

typedef Fp<P377> Fq;    // BLS12-377 base field (= the scalar field of BW6-761)
typedef Fp2<P377> Fq2;
typedef Fp<P761> Fw;    // BW6-761 base field

// lane j of a group reads lane QP(..)[j] of the same group
#define QP(a, b, c) ((a) | ((b) << 2) | ((c) << 4))

// a base policy over any field F of fp.h / fp2.h: what the backends need from it (the pairing towers bring richer policies)
template <class F> struct FieldBase {
  typedef F T;
  HD static T zero() { return F::zero(); }
  HD static T one() { return F::one(); }
  HD static T mul_inl(const T& a, const T& b) { return F::mul(a, b); }
  HD static T add(const T& a, const T& b) { return F::norm(F::add(a, b)); }
  HD static T dbl(const T& a) { return F::norm(F::add(a, a)); }
  HD static T tpl(const T& a) { return F::norm(F::add(F::add(a, a), a)); }
  template <int K> HD static T sub(const T& a, const T& b) { return F::norm(F::template sub<K, 1>(a, b)); }
  template <int K> HD static T neg(const T& a) { return F::norm(F::template neg<K, 1>(a)); }
  HD static bool is_zero(const T& a) { return a.is_zero_mod_p(); }
};

// ================================================================== host backend: three explicit lanes, any base policy
template <class BP> struct QHostT {
  typedef typename BP::T T;
  static constexpr int NL = 3;
  struct V { T v[NL]; };
  template <class Fn> static V map2(const V& a, const V& b, Fn fn) { V r; for (int i = 0; i < NL; i++) r.v[i] = fn(a.v[i], b.v[i]); return r; }
  template <class Fn> static V map1(const V& a, Fn fn) { V r; for (int i = 0; i < NL; i++) r.v[i] = fn(a.v[i]); return r; }
  static V uni(const T& x) { V r; for (int i = 0; i < NL; i++) r.v[i] = x; return r; }
  static V mul(const V& a, const V& b) { return map2(a, b, [](const T& x, const T& y) { return BP::mul_inl(x, y); }); }
  static V add(const V& a, const V& b) { return map2(a, b, [](const T& x, const T& y) { return BP::add(x, y); }); }
  static V dbl(const V& a) { return map1(a, [](const T& x) { return BP::dbl(x); }); }
  static V tpl(const V& a) { return map1(a, [](const T& x) { return BP::tpl(x); }); }
  template <int K> static V sub(const V& a, const V& b) { return map2(a, b, [](const T& x, const T& y) { return BP::template sub<K>(x, y); }); }
  template <int K> static V neg(const V& a) { return map1(a, [](const T& x) { return BP::template neg<K>(x); }); }
  static V wred(const V& a) { return map1(a, [](const T& x) { return BP::wred(x); }); }
  static V lred(const V& a) { return wred(a); }
  static V mul_nr(const V& a) { return map1(a, [](const T& x) { return BP::mul_nr(x); }); }
  // the "lazy" forms (result only ever handed to sub / wred, which carry-propagate themselves; mul_nr_k: operand vb <= K):
  // these backends keep every value normalised, the six-lane ones (pairing_lanes.h) skip the carry passes
  static constexpr bool LAZY = false;
  static V add_l(const V& a, const V& b) { return add(a, b); }
  static V dbl_l(const V& a) { return dbl(a); }
  template <int K> static V sub_l(const V& a, const V& b) { return sub<K>(a, b); }
  template <int K> static V mul_nr_k(const V& a) { return mul_nr(a); }
  template <int K> static V mul_nr_k_l(const V& a) { return mul_nr(a); }
  static V half(const V& a) { return map1(a, [](const T& x) { return BP::half(x); }); }
  static V inv(const V& a) { return map1(a, [](const T& x) { return BP::inv_inl(x); }); }
  template <int CTRL> static V perm(const V& x) { V r; for (int i = 0; i < NL; i++) r.v[i] = x.v[(CTRL >> (2 * i)) & 3]; return r; }
  template <int K> static V bcast(const V& x) { return perm<QP(K, K, K)>(x); }
  template <int K> static V sel(const V& onk, const V& other) { V r; for (int i = 0; i < NL; i++) r.v[i] = (i == K) ? onk.v[i] : other.v[i]; return r; }
  static V pick(const V& a0, const V& a1, const V& a2) { V r; r.v[0] = a0.v[0]; r.v[1] = a1.v[1]; r.v[2] = a2.v[2]; return r; }
  static V zero() { return uni(BP::zero()); }
  static V one() { return uni(BP::one()); }
  static bool is_zero_u(const V& a) { return BP::is_zero(a.v[0]); }      // for group-uniform values
  // all three lanes must hold: a == (lane 0 ? 1 : 0) and b == 0
  static bool is_one3(const V& a, const V& b) {
    bool ok = BP::is_one(a.v[0]) && BP::is_zero(b.v[0]);
    for (int i = 1; i < 3; i++) ok = ok && BP::is_zero(a.v[i]) && BP::is_zero(b.v[i]);
    return ok;
  }
};
#if defined(__HIPCC__)
// ================================================================== device backend: groups of three lanes, ds_bpermute
#define QDEV __device__ __forceinline__
template <class BP> struct QTriT {
  typedef typename BP::T T;
  typedef T V;
  static constexpr int NL = 3, GROUPS_PER_WAVE = 21;
  static constexpr int NWORDS = (int)(sizeof(T) / 4);   // 28 limbs for Fq2 of BLS12-377 and Fq of BW6-761, 14 for Fq of BLS12-377
  QDEV static int wave_lane() { return (int)__lane_id(); }
  QDEV static int group() { return (wave_lane() * 86) >> 8; }            // lane / 3 for lane < 64
  QDEV static int lane() { return wave_lane() - 3 * group(); }          // lane % 3
  QDEV static V mul(const V& a, const V& b) { return BP::mul_inl(a, b); }
  QDEV static V add(const V& a, const V& b) { return BP::add(a, b); }
  QDEV static V dbl(const V& a) { return BP::dbl(a); }
  QDEV static V tpl(const V& a) { return BP::tpl(a); }
  template <int K> QDEV static V sub(const V& a, const V& b) { return BP::template sub<K>(a, b); }
  template <int K> QDEV static V neg(const V& a) { return BP::template neg<K>(a); }
  QDEV static V wred(const V& a) { return BP::wred(a); }
  QDEV static V lred(const V& a) { return BP::wred(a); }
  QDEV static V mul_nr(const V& a) { return BP::mul_nr(a); }
  static constexpr bool LAZY = false;
  QDEV static V add_l(const V& a, const V& b) { return BP::add(a, b); }      // lazy forms: see QHostT
  QDEV static V dbl_l(const V& a) { return BP::dbl(a); }
  template <int K> QDEV static V sub_l(const V& a, const V& b) { return BP::template sub<K>(a, b); }
  template <int K> QDEV static V mul_nr_k(const V& a) { return BP::mul_nr(a); }
  template <int K> QDEV static V mul_nr_k_l(const V& a) { return BP::mul_nr(a); }
  QDEV static V half(const V& a) { return BP::half(a); }
  QDEV static V inv(const V& a) { return BP::inv_inl(a); }
  template <int CTRL> QDEV static int src_addr() {
    const int j = lane();
    return (wave_lane() - j + ((CTRL >> (2 * j)) & 3)) << 2;
  }
  // word i of a base element: Fq2 = (c0.l[0..13], c1.l[0..13]), Fq761 = l[0..27] (member access keeps the values in VGPRs)
  QDEV static uint32_t& word(Fq2& x, int i) { return i < 14 ? x.c0.l[i] : x.c1.l[i - 14]; }
  QDEV static const uint32_t& word(const Fq2& x, int i) { return i < 14 ? x.c0.l[i] : x.c1.l[i - 14]; }
  QDEV static uint32_t& word(Fw& x, int i) { return x.l[i]; }
  QDEV static uint32_t& word(Fq& x, int i) { return x.l[i]; }
  QDEV static const uint32_t& word(const Fq& x, int i) { return x.l[i]; }
  QDEV static const uint32_t& word(const Fw& x, int i) { return x.l[i]; }
  template <int CTRL> QDEV static V perm(const V& x) {
    const int addr = src_addr<CTRL>();
    V r;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) word(r, i) = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)word(x, i));
    return r;
  }
  template <int K> QDEV static V bcast(const V& x) { return perm<QP(K, K, K)>(x); }
  // selections by mask arithmetic: as `c ? a : b` the compiler sinks the operands' computations into exec-masked regions (hundreds
  // per Miller step: scalar bookkeeping for no saved work on a SIMD, and divergent regions cost registers) - see QHex377::choose
  QDEV static uint32_t lane_mask(bool c) { return 0u - (uint32_t)c; }
  template <int K> QDEV static V sel(const V& onk, const V& other) {
    const uint32_t m = lane_mask(lane() == K);
    V r;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) word(r, i) = (word(onk, i) & m) | (word(other, i) & ~m);
    return r;
  }
  QDEV static V pick(const V& a0, const V& a1, const V& a2) {
    const int q = lane();
    const uint32_t m0 = lane_mask(q == 0), m1 = lane_mask(q == 1);
    V r;
#pragma unroll
    for (int i = 0; i < NWORDS; i++) {
      const uint32_t t = (word(a1, i) & m1) | (word(a2, i) & ~m1);
      word(r, i) = (word(a0, i) & m0) | (t & ~m0);
    }
    return r;
  }
  QDEV static V zero() { return BP::zero(); }
  QDEV static V one() { return BP::one(); }
  QDEV static bool is_zero_u(const V& a) { return BP::is_zero(a); }       // for group-uniform values
  QDEV static bool is_one3(const V& a, const V& b) {
    const int q = lane();
    const int ok = ((q == 0 ? BP::is_one(a) : BP::is_zero(a)) && BP::is_zero(b)) ? 1 : 0;
    const int base = (wave_lane() - q) << 2;
    return (__builtin_amdgcn_ds_bpermute(base, ok) & __builtin_amdgcn_ds_bpermute(base + 4, ok) & __builtin_amdgcn_ds_bpermute(base + 8, ok)) != 0;
  }
};
#define QFN __host__ __device__ __forceinline__
#define QNI __host__ __device__ __attribute__((noinline))   // out of line: one copy of the 6-product Fq12 routines per kernel
#else
#define QFN inline
#define QNI inline
#endif


}  // namespace celo
