// Reduced-radix Montgomery prime-field arithmetic for gfx950 (and the host epilogues).
//
// Replaces ark-ff's Fp384 / Fp768 Montgomery arithmetic that the reference reaches through
// VariableBaseMSM::multi_scalar_mul and PairingEngine::product_of_pairings
// (crates/bls-crypto/src/bls/public.rs:61,102; signature.rs:85,149).
//
// MI355X design notes (measured with tools/ubench_valu.hip):
//  * v_mad_u64_u32 issues at ~the cost of any other 3-operand VALU op, and the carry chain of a
//    full-radix (2^32) multiply needs ~2 extra VALU ops per product.  So limbs are W = 28 bits
//    in 32-bit registers: a whole product column (<= 2L terms of < 2^58) is accumulated into
//    ONE 64-bit register pair by back-to-back v_mad_u64_u32 with NO carry handling; one shift
//    per column propagates.  (FIPS: finely integrated product scanning.)
//  * Values are kept loosely reduced (value < ~64p, limbs < ~3*2^W): additions are plain limb
//    adds, subtractions add a redundant multiple of p whose limbs dominate the subtrahend's,
//    and carries are only propagated ("norm") where a bound below requires it.
//  * Montgomery radix is R_d = 2^(W*L) (2^392 / 2^784), not arkworks' 2^384 / 2^768; one
//    multiplication by a constant converts on the way in and out (from_ark / to_ark).
//
// Bounds contract (checked at run time in host builds with -DCELO_FP_TRACK):
//   lb = max limb / 2^W   (limb bound),  vb = value / p  (value bound)
//   mul(a,b), sqr(a):  L*(lb_a*lb_b + 1) <= 255  and  vb_a*vb_b <= 2^15  ->  lb = 1, vb < 2
//   add(a,b):          lb = lb_a + lb_b,  vb = vb_a + vb_b
//   sub<K,M>(a,b):     needs lb_b <= M, vb_b <= K                -> lb = lb_a + M + 1, vb = vb_a + K
//                      and vb_b < K strictly unless a carry pass follows before any product (the top limb of K p's redundant form
//                      is one below K p's own; the tracker's `topwrap` flag follows such a difference until norm() clears it)
//   norm(a):           lb = 1 (top limb keeps the excess), vb unchanged
#pragma once
#include <cstdint>
#include "fp_consts.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
// Device pass: everything is force-inlined (register-resident operands).  Host pass: plain `inline` — the host only runs
// epilogues and plumbing, and force-inlining the ~600-instruction multiply bodies everywhere made the host pass the long
// pole of the build (158 s of a 180 s unit).
#if defined(__HIP_DEVICE_COMPILE__)
#define HD __host__ __device__ __forceinline__
#else
#define HD __host__ __device__ inline
#endif
#else
#define HD inline
struct uint4 { uint32_t x, y, z, w; };  // host-only builds (bounds-tracking unit tests)
#endif

// CELO_LONG_BRANCH note.  Out-of-line (noinline) DEVICE functions of the wide fields are longer than the reach of s_cbranch (2^17 bytes), so
// branch relaxation expands their far branches into s_getpc_b64 / s_add_u32 / s_addc_u32 / s_setpc_b64 on an SGPR pair.  With this compiler
// (ROCm 7.2, clang 22) the pair is the "long-branch reserved register": the highest free pair before register allocation, moved to the lowest
// pair the function leaves unused after it.  In a leaf function that needs few SGPRs the lowest unused pair is s[30:31] - the function's own
// return address: the first far branch taken overwrites it and the function returns into its own body.  That is round 5's "k_combine_big<G_761>
// never returns with immediate K p tables": without the tables' ~100 SGPRs xyzz_add_outline<Fp<P761>> became such a function (with them the pair
// was s[98:99], which callers know about through the callee's clobber mask).  Every unit is therefore compiled with
// `-mllvm -amdgpu-long-branch-factor=0` (csrc/Makefile): no reservation, the pair is scavenged from registers proved dead (s[2:3], s[4:5] ...).
// tools/scan_long_branch.py checks the built library for the shape (tests/test_abi_symbols.py); tools/repro_combine/ is the reproducer.

#ifdef CELO_FP_TRACK
#include <cassert>
#include <cstdio>
#define TRK(x) x
#else
#define TRK(x)
#endif

namespace celo {

template <class P> struct Fp {
  static constexpr int L = P::L;
  static constexpr int W = P::W;
  static constexpr uint32_t MASK = P::MASK;
  uint32_t l[L];
  TRK(double lb = 1; double vb = 2; bool topwrap = false;)

  HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = 0;
    TRK(r.lb = 0; r.vb = 0;)
    return r;
  }
  HD static Fp from_limbs(const uint32_t* c) {
    Fp r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = c[i];
    TRK(r.lb = 1; r.vb = 2;)
    return r;
  }
  HD static Fp one() { return from_limbs(P::ONE); }

  // The Montgomery quotient digit -p^-1 * lo mod 2^W.  p = 1 mod 2^W (BLS12-377's Fq and Fr): a negation.  Otherwise (BW6-761) a
  // multiplication whose low word is all that is used - which the compiler narrows to v_mul_lo_u32, a QUARTER-rate instruction on
  // gfx950 (one per column: 28 per product, 6 % of the issue slots of the BW6-761 kernels).  Keeping the 64-bit product opaque makes
  // it a v_mad_u64_u32, which issues at the rate of every other multiply-add of the pass.
  HD static uint32_t mont_digit(uint32_t lo) {
    if constexpr (P::INV == MASK) return (0u - lo) & MASK;
    else {
#if defined(__HIP_DEVICE_COMPILE__)
      uint64_t q = (uint64_t)lo * P::INV;
      asm("" : "+v"(q));                       // no instruction: only hides from the optimiser that the high word is unused
      return (uint32_t)q & MASK;
#else
      return (lo * P::INV) & MASK;
#endif
    }
  }

  // ---- one column of a Montgomery pass: the quotient digit m of the column sum, then (acc + m p_0) >> W.
  // For p = 1 mod 2^W (both BLS12-377 fields: p_0 = 1, m = -acc mod 2^W) the sum acc + m is acc rounded UP to a multiple of 2^W, so
  // (acc + m) >> W = (acc + 2^W - 1) >> W and m = ~(acc + 2^W - 1) mod 2^W: one 64-bit add of a constant (which the compiler folds
  // into the column's multiply-add chain), a not-and and the shift - instead of a negation, a mask, a zero-extension move, a 64-bit
  // add of m and the shift.  Two instructions less per column, 28 per product of the 14-limb field: DESIGN.md section 3.
  template <bool SIGNED> HD static uint32_t mont_step(uint64_t& acc) {
    uint32_t m;
    if constexpr (P::INV == MASK) {
      static_assert(P::INV != MASK || P::P[0] == 1, "p = 1 mod 2^W");
      acc += MASK;
      m = ~(uint32_t)acc & MASK;
    } else {
      m = mont_digit((uint32_t)acc);
      acc += (uint64_t)m * P::P[0];
    }
    if constexpr (SIGNED) acc = (uint64_t)((int64_t)acc >> W);
    else acc >>= W;
    return m;
  }

  // ---- Montgomery multiplication, product scanning with interleaved reduction.
  // For L > 25 (BW6-761: 28 limbs) a column of 2L products only fits 64 bits when lb_a*lb_b <= 8, so the
  // un-normalised subtraction results (lb 3) the curve formulas feed in are normalised on entry there.
  static constexpr bool NORM_IN = (L * 10 > 255);
  // NIN = false: the caller hands over a first operand whose limbs already meet the column bound (the _nn forms below)
  template <bool NIN = NORM_IN> HD static Fp mul(const Fp& a_, const Fp& b) {
    const Fp a = NIN ? norm(a_) : a_;
    TRK(assert(L * (a.lb * b.lb + 1) <= 255.5); assert(a.vb * b.vb <= 32768.0); assert(!a.topwrap && !b.topwrap);)
    Fp r;
    uint32_t m[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
      m[k] = mont_step<false>(acc);
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::P[k - i];
      r.l[k - L] = (uint32_t)acc & MASK;
      acc >>= W;
    }
    r.l[L - 1] = (uint32_t)acc;
    TRK(r.lb = 1; r.vb = 2;)
    return r;
  }
  template <bool NIN = NORM_IN> HD static Fp sqr(const Fp& a_) {
    const Fp a = NIN ? norm(a_) : a_;
    TRK(assert(L * (a.lb * a.lb + 1) <= 255.5); assert(a.vb * a.vb <= 32768.0); assert(!a.topwrap);)
    Fp r;
    uint32_t m[L], a2[L];
#pragma unroll
    for (int i = 0; i < L; i++) a2[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
      for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.l[k - i];
      if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
      m[k] = mont_step<false>(acc);
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = k - L + 1; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.l[k - i];
      if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::P[k - i];
      r.l[k - L] = (uint32_t)acc & MASK;
      acc >>= W;
    }
    r.l[L - 1] = (uint32_t)acc;
    TRK(r.lb = 1; r.vb = 2;)
    return r;
  }

  HD static Fp add(const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = a.l[i] + b.l[i];
    TRK(r.lb = a.lb + b.lb; r.vb = a.vb + b.vb; assert(r.lb <= 15); r.topwrap = a.topwrap || b.topwrap;)
    return r;
  }
  HD static Fp dbl(const Fp& a) { return add(a, a); }
  // a - b + K*p  (K*p in redundant form dominating limbs of b up to M*(2^W-1))
  template <int K, int M> HD static const uint32_t* kp() {
    if constexpr (K == 4 && M == 1) return P::KP4_M1;
    else if constexpr (K == 8 && M == 1) return P::KP8_M1;
    else if constexpr (K == 16 && M == 1) return P::KP16_M1;
    else if constexpr (K == 32 && M == 1) return P::KP32_M1;
    else if constexpr (K == 64 && M == 1) return P::KP64_M1;
    else if constexpr (K == 8 && M == 3) return P::KP8_M3;
    else if constexpr (K == 16 && M == 3) return P::KP16_M3;
    else if constexpr (K == 32 && M == 3) return P::KP32_M3;
    else return nullptr;
  }
  template <int K, int M = 1> HD static Fp sub(const Fp& a, const Fp& b) {
    TRK(assert(b.lb <= M + 1e-9); assert(b.vb <= K);)
    Fp r;
    if constexpr (K == 4 && M == 1) {
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = a.l[i] + P::KP4_M1[i] - b.l[i];
    } else if constexpr (K == 8 && M == 1) {
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = a.l[i] + P::KP8_M1[i] - b.l[i];
    } else if constexpr (K == 16 && M == 1) {
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = a.l[i] + P::KP16_M1[i] - b.l[i];
    } else if constexpr (K == 32 && M == 1) {
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = a.l[i] + P::KP32_M1[i] - b.l[i];
    } else if constexpr (K == 64 && M == 1) {
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = a.l[i] + P::KP64_M1[i] - b.l[i];
    } else if constexpr (K == 8 && M == 3) {
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = a.l[i] + P::KP8_M3[i] - b.l[i];
    } else if constexpr (K == 16 && M == 3) {
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = a.l[i] + P::KP16_M3[i] - b.l[i];
    } else {
      static_assert(K == 32 && M == 3, "unsupported sub<K,M>");
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = a.l[i] + P::KP32_M3[i] - b.l[i];
    }
    // a subtrahend that may reach K p itself can carry K p's own top limb, one more than the redundant form's: the difference's top
    // limb then wraps modulo 2^32 when the minuend's is zero.  The value is still right modulo 2^32 per limb, so a carry pass (norm)
    // repairs it; a product must not see it (topwrap: asserted by every product and by wred's uncarried quotient estimate).
    TRK(r.lb = a.lb + M + 1; r.vb = a.vb + K; assert(r.lb <= 15); r.topwrap = a.topwrap || b.topwrap || !(b.vb < K);)
    return r;
  }
  template <int K, int M = 1> HD static Fp neg(const Fp& b) { return sub<K, M>(zero(), b); }
  // carry propagation: limbs 0..L-2 < 2^W, the top limb keeps whatever is left (value < 2^(W*L) assumed)
  HD static Fp norm(const Fp& a) {
    Fp r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      uint32_t t = a.l[i] + c;
      r.l[i] = t & MASK;
      c = t >> W;
    }
    r.l[L - 1] = a.l[L - 1] + c;
    TRK(r.lb = 1; r.vb = a.vb;)
    return r;
  }

  // ---- exact halving mod p: (a + (a odd ? p : 0)) >> 1.  lb = 1, vb -> (vb + 1) / 2.  Used by the lane-parallel pairing
  // (pairing_lanes.h) in place of the multiplications by 1/2 of ark-ec's doubling step: same field element, ~50 VALU ops.
  HD static Fp half(const Fp& a_) {
    const Fp a = norm(a_);
    const uint32_t odd = 0u - (a.l[0] & 1u);
    Fp t;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      uint32_t s = a.l[i] + (P::P[i] & odd) + c;
      t.l[i] = s & MASK;
      c = s >> W;
    }
    t.l[L - 1] = a.l[L - 1] + (P::P[L - 1] & odd) + c;
    Fp r;
#pragma unroll
    for (int i = 0; i < L - 1; i++) r.l[i] = (t.l[i] >> 1) | ((t.l[i + 1] & 1u) << (W - 1));
    r.l[L - 1] = t.l[L - 1] >> 1;
    TRK(r.lb = 1; r.vb = (a.vb + 1) / 2;)
    return r;
  }

  // ---- weak reduction: value < ~300p (normalised limbs)  ->  value < ~2.1p.  Quotient estimate from the top limb(s),
  // one multiply-subtract sweep, ~60-110 VALU ops: used by the pairing towers
  // to stop the value growth of lazy additions without paying a full Montgomery multiplication.
  HD static Fp wred(const Fp& a_) {
    TRK(assert(a_.vb <= 300);)
    uint32_t q;
    if constexpr (P::P[L - 1] >= 4096) {
      // BLS12-377: 13-bit top limb, integer reciprocal - and NO carry pass first.  The quotient is estimated from the top limb AS IT
      // IS: with lazy limbs (lb <= 15) it is at most 16 below the carried one, so q stays <= floor(a / p) and a - q p < 2.1 p still
      // holds (16 / 6884 more); the sweep below carries while it subtracts and takes limbs of up to 32 bits.
      TRK(assert(a_.lb <= 15); assert(!a_.topwrap);)
      constexpr uint64_t D = (uint64_t)P::P[L - 1] + 1;
      constexpr uint64_t M = ((1ull << 34) + D - 1) / D;      // ceil(2^34 / D); exact floor(t/D) for t < 2^34 / D (a < 362 p)
      q = (uint32_t)(((uint64_t)a_.l[L - 1] * M) >> 34);
      Fp r;
      int64_t carry = 0;
#pragma unroll
      for (int i = 0; i < L - 1; i++) {
        int64_t t = (int64_t)a_.l[i] - (int64_t)((uint64_t)q * P::P[i]) + carry;
        r.l[i] = (uint32_t)t & MASK;
        carry = t >> W;
      }
      r.l[L - 1] = (uint32_t)((int64_t)a_.l[L - 1] - (int64_t)((uint64_t)q * P::P[L - 1]) + carry);
      TRK(r.lb = 1; r.vb = 3;)
      return r;
    }
    const Fp a = norm(a_);
    {                                                         // BW6-761: 5-bit top limb -> estimate from the top two limbs
      constexpr double DINV = 1.0 / (double)((((uint64_t)P::P[L - 1]) << W) + P::P[L - 2] + 1);
      const uint64_t t2 = (((uint64_t)a.l[L - 1]) << W) + a.l[L - 2];
      const uint32_t e = (uint32_t)((double)t2 * DINV);
      q = e ? e - 1 : 0;                                      // never above floor(a/p)
    }
    Fp r;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      int64_t t = (int64_t)a.l[i] - (int64_t)((uint64_t)q * P::P[i]) + carry;
      r.l[i] = (uint32_t)t & MASK;
      carry = t >> W;
    }
    r.l[L - 1] = (uint32_t)((int64_t)a.l[L - 1] - (int64_t)((uint64_t)q * P::P[L - 1]) + carry);
    TRK(r.lb = 1; r.vb = 3;)
    return r;
  }

  // ---- canonical reduction to [0, p): value must be < 128p.  Slow path only (equality tests, export).
  // (round 5: the K p tables enter as VALUES of a function-local constexpr array, not through a pointer to P::NPk.  A pointer into a
  // host-device constexpr member is an odr-use: the device build then holds the table as a global the host could overwrite, its limbs cannot
  // become immediates, and the compiler hoisted the 7 x L scalar loads of this COLD path into the prologue of every kernel that tests a value
  // for zero - ~100 SGPRs parked in VGPR lanes for the kernel's lifetime: "SGPRs Spill: 107-131" of k_accumulate, the register shape VERDICT r4
  // item 5 asked to leave.)
  template <int K> static constexpr uint32_t np_limb(int i) {
    return K == 64 ? P::NP64[i] : K == 32 ? P::NP32[i] : K == 16 ? P::NP16[i] : K == 8 ? P::NP8[i] : K == 4 ? P::NP4[i] : K == 2 ? P::NP2[i] : P::NP1[i];
  }
  struct KpTable { uint32_t v[L]; };
  template <int K> static constexpr KpTable kp_table() {
    KpTable t{};
    for (int i = 0; i < L; i++) t.v[i] = np_limb<K>(i);
    return t;
  }
  HD static Fp reduce(const Fp& a) {
    Fp r = norm(a);
    cond_sub_k<64>(r);
    cond_sub_k<32>(r);
    cond_sub_k<16>(r);
    cond_sub_k<8>(r);
    cond_sub_k<4>(r);
    cond_sub_k<2>(r);
    cond_sub_k<1>(r);
    TRK(r.lb = 1; r.vb = 1;)
    return r;
  }
  template <int K> HD static void cond_sub_k(Fp& r) {
#if defined(CELO_KP_PTR_TABLES)
    // reproducer builds only (tools/repro_combine): rounds 1-5's pointer form of the tables for the 28-limb field.  Round 5 kept it there because
    // the immediates form made `k_combine_big<G_761>` never return; round 6 found why (CELO_LONG_BRANCH note above): the
    // out-of-line addition it calls lost its return address to a far branch.  Nothing to do with the tables.
    constexpr bool PTR_TABLES = L > 14;
#else
    constexpr bool PTR_TABLES = false;
#endif
    if constexpr (PTR_TABLES) {
      cond_sub(r, K == 64 ? P::NP64 : K == 32 ? P::NP32 : K == 16 ? P::NP16 : K == 8 ? P::NP8 : K == 4 ? P::NP4 : K == 2 ? P::NP2 : P::NP1);
#if defined(__HIP_DEVICE_COMPILE__)
    } else if constexpr (L > 14) {
      // The 28-limb field on the device (end of round 6): the table behind a pointer the optimiser cannot see through, so that its scalar
      // loads STAY in this cold block - not hoisted into every kernel's prologue and parked in VGPR lanes (the pointer form of rounds 1-5:
      // 217 spilled SGPRs in k_accumulate<G_761>), and not 28 literals per cond_sub either (the immediates form: same-box A/B on config 4,
      // profiles/r6_ab_kp_tables_761.txt: accumulate 30.5 ms with immediates, 29.8 with the old pointers, 28.9 with this form - 4 spilled
      // SGPRs, 800 fewer instructions per mixed addition).  The 14-limb fields are level either way (hipcc -S) and keep the immediates.
      const uint32_t* kp = K == 64 ? P::NP64 : K == 32 ? P::NP32 : K == 16 ? P::NP16 : K == 8 ? P::NP8 : K == 4 ? P::NP4 : K == 2 ? P::NP2 : P::NP1;
      asm volatile("" : "+s"(kp));
      cond_sub(r, kp);
#endif
    } else {
      constexpr KpTable T = kp_table<K>();
      cond_sub(r, T.v);
    }
  }
  HD static void cond_sub(Fp& r, const uint32_t* kp) {
    // r (normalised) >= kp ? r - kp : r
    uint32_t t[L];
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      int32_t d = (int32_t)r.l[i] - (int32_t)kp[i] + borrow;
      t[i] = (uint32_t)d & MASK;
      borrow = d >> W;  // arithmetic shift: 0 or -1
    }
    int64_t dt = (int64_t)r.l[L - 1] - (int64_t)kp[L - 1] + borrow;
    bool ge = dt >= 0;
    t[L - 1] = (uint32_t)dt;
    if (ge) {
#pragma unroll
      for (int i = 0; i < L; i++) r.l[i] = t[i];
    }
  }
  // exact zero test (mod p).  Fast filter on the low limb: x == j*p for small j requires
  // (x0 * p0^-1) mod 2^W == j; only then pay for the canonical reduction.
  HD bool is_zero_mod_p() const {
    uint32_t lo = l[0] & MASK;
    uint32_t t = (P::INV == MASK) ? ((0u - lo) & MASK) : ((lo * P::INV) & MASK);  // = -j mod 2^W
    uint32_t j = (0u - t) & MASK;
#if defined(CELO_ZERO_ALWAYS) && defined(__HIP_DEVICE_COMPILE__)
    // reproducer builds only (tools/repro_acc/REPORT.md): the canonical reduction on EVERY test, no low-limb filter - if the cold path is what
    // goes wrong in the signed instantiation, this makes it go wrong everywhere
    (void)j;
    {
      Fp r = reduce(*this);
      uint32_t o = 0;
#pragma unroll
      for (int i = 0; i < L; i++) o |= r.l[i];
      return o == 0;
    }
#elif defined(CELO_ZERO_UNIFORM) && defined(__HIP_DEVICE_COMPILE__)
    // reproducer builds only (tools/repro_acc, DESIGN.md section 3 "the signed pass"): the slow path entered by the whole wave on a vote.
    // Together with the signed form of xyzz_madd's Fq2 pass this instantiation of k_accumulate<G2_377> gives wrong sums in ~0.7 % of
    // its waves, deterministically - the compact form of the round-3 finding.  Not built into the library.
    const bool maybe = j <= 130;
    if (!__any(maybe ? 1 : 0)) return false;
    Fp r = reduce(*this);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < L; i++) o |= r.l[i];
    return maybe && o == 0;
#else
    if (j > 130) return false;
    Fp r = reduce(*this);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < L; i++) o |= r.l[i];
    return o == 0;
#endif
  }
  HD bool limbs_all_zero() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < L; i++) o |= l[i];
    return o == 0;
  }
  HD static bool eq_mod_p(const Fp& a, const Fp& b) {  // a, b normalised (lb<=1), vb <= 64
    return sub<64, 1>(a, b).is_zero_mod_p();
  }


  // ---- device memory layout: L limbs padded to a multiple of 4 words (16-byte vector accesses)
  static constexpr int WORDS = (L + 3) / 4 * 4;
  static constexpr int ARK64 = P::N64;
  HD static Fp load(const uint32_t* p) {
    Fp r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int k = 0; k < WORDS / 4; k++) {
      uint4 v = q[k];
      if (4 * k < L) r.l[4 * k] = v.x;
      if (4 * k + 1 < L) r.l[4 * k + 1] = v.y;
      if (4 * k + 2 < L) r.l[4 * k + 2] = v.z;
      if (4 * k + 3 < L) r.l[4 * k + 3] = v.w;
    }
    TRK(r.lb = 1; r.vb = 64;)
    return r;
  }
  HD void store(uint32_t* p) const {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int k = 0; k < WORDS / 4; k++) {
      uint4 v;
      v.x = 4 * k < L ? l[4 * k] : 0u;
      v.y = 4 * k + 1 < L ? l[4 * k + 1] : 0u;
      v.z = 4 * k + 2 < L ? l[4 * k + 2] : 0u;
      v.w = 4 * k + 3 < L ? l[4 * k + 3] : 0u;
      q[k] = v;
    }
  }

  // -(x*y) as one SIGNED multiply-add operand (v_mad_i64_i32) instead of an unsigned product chain merged by a 64-bit subtraction per
  // column (v_sub_co + v_subb: 54 instructions per pass).  x, y < 2^31 as values.
  HD static uint64_t nprod(uint32_t x, uint32_t y) { return (uint64_t)((int64_t)(0 - (int32_t)x) * (int64_t)(int32_t)y); }
  // ---- sum of two products in one reduction pass:  (a*b + KC*c*d)/R  (+ p when KC < 0, which keeps the result positive)
  // with KC in {+1, -1, -5}.  One Montgomery reduction for two limb-product sweeps.  Users: Fp2 (u^2 = -5: c0 = a0 b0 - 5 a1 b1,
  // c1 = a0 b1 + a1 b0) and the curve formulas (Y3 = R*t - Y1*PPP).  Column bound: L*(lb_a*lb_b + |KC|*lb_c*lb_d + 1) <= 255.
  template <int KC> HD static Fp mul2k(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
    static_assert(KC == 1 || KC == -1 || KC == -5, "unsupported multiplier");
    constexpr uint32_t AK = KC < 0 ? (uint32_t)(-KC) : (uint32_t)KC;
    constexpr bool NEG = KC < 0;
    TRK(assert(L * (a.lb * b.lb + AK * c.lb * d.lb + 1) <= 255.5); assert(a.vb * b.vb + AK * c.vb * d.vb <= 32768.0);
        assert(!a.topwrap && !b.topwrap && !c.topwrap && !d.topwrap);)
    Fp r;
    uint32_t m[L], cc[L];
#pragma unroll
    for (int i = 0; i < L; i++) cc[i] = AK == 1 ? c.l[i] : c.l[i] * AK;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc += (uint64_t)a.l[i] * b.l[k - i];
        if (KC == -1) acc += nprod(cc[i], d.l[k - i]);     // the curve formulas' R t - Y1 PPP (base field): signed multiply-adds
        else if (NEG) acc -= (uint64_t)cc[i] * d.l[k - i];
        else acc += (uint64_t)cc[i] * d.l[k - i];
      }
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
      m[k] = mont_step<true>(acc);
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = k - L + 1; i < L; i++) {
        acc += (uint64_t)a.l[i] * b.l[k - i];
        if (KC == -1) acc += nprod(cc[i], d.l[k - i]);     // the curve formulas' R t - Y1 PPP (base field): signed multiply-adds
        else if (NEG) acc -= (uint64_t)cc[i] * d.l[k - i];
        else acc += (uint64_t)cc[i] * d.l[k - i];
      }
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::P[k - i];
      if (NEG) acc += P::P[k - L];  // + p after the division by R keeps the result positive
      r.l[k - L] = (uint32_t)acc & MASK;
      acc = (uint64_t)((int64_t)acc >> W);
    }
    if (NEG) acc += P::P[L - 1];
    r.l[L - 1] = (uint32_t)acc;
    TRK(r.lb = 1; r.vb = NEG ? 3 : 2;)
    return r;
  }
  // (a*b + c*d)/R + p with c handed over as SIGNED limbs cs[i] (|cs[i]| <= 5 * 2^W): one instruction stream for both halves of an
  // Fq2 product when the two halves sit in two different lanes (lanes.h, QHex: c0 = a0 b0 - 5 a1 b1 takes cs = -5 a1, c1 =
  // a0 b1 + a1 b0 takes cs = a1; v_mad_i64_i32 costs what v_mad_u64_u32 costs).  a, b, d normalised.  Column bound as mul2k<-5>.
  HD static Fp mul2s(const Fp& a, const Fp& b, const int32_t* cs, const Fp& d) {
    TRK(assert(a.lb <= 1 && b.lb <= 1 && d.lb <= 1); assert(a.vb * b.vb + 5 * 64 * d.vb <= 32768.0); assert(!a.topwrap && !b.topwrap && !d.topwrap);)
    Fp r;
    uint32_t m[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc += (uint64_t)a.l[i] * b.l[k - i];
        acc += (uint64_t)((int64_t)cs[i] * (int64_t)(int32_t)d.l[k - i]);
      }
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
      m[k] = mont_step<true>(acc);
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = k - L + 1; i < L; i++) {
        acc += (uint64_t)a.l[i] * b.l[k - i];
        acc += (uint64_t)((int64_t)cs[i] * (int64_t)(int32_t)d.l[k - i]);
      }
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::P[k - i];
      acc += P::P[k - L];  // + p after the division by R keeps the result positive
      r.l[k - L] = (uint32_t)acc & MASK;
      acc = (uint64_t)((int64_t)acc >> W);
    }
    acc += P::P[L - 1];
    r.l[L - 1] = (uint32_t)acc;
    TRK(r.lb = 1; r.vb = 3;)
    return r;
  }
  template <bool SUB5> HD static Fp mul2(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {  // inputs normalised (Fp2)
    TRK(assert(a.lb <= 1 && b.lb <= 1 && c.lb <= 1 && d.lb <= 1);)
    return mul2k<SUB5 ? -5 : 1>(a, b, c, d);
  }
  // ---- a^2 - 5 c^2 in one reduction pass with the symmetric limb products taken once (Fp2 squaring's real part, u^2 = -5): 2 x 105
  // limb products instead of the 2 x 196 of mul2k<-5>(a, a, c, c).  Inputs normalised; column bound as mul2k<-5>.
  HD static Fp sqr2m5(const Fp& a, const Fp& c) {
    TRK(assert(a.lb <= 1 && c.lb <= 1); assert(L * 7 <= 255); assert(a.vb * a.vb + 5 * c.vb * c.vb <= 32768.0); assert(!a.topwrap && !c.topwrap);)
    Fp r;
    uint32_t m[L], a2[L], c5[L], c10[L];
#pragma unroll
    for (int i = 0; i < L; i++) { a2[i] = a.l[i] << 1; c5[i] = c.l[i] * 5u; c10[i] = c.l[i] * 10u; }
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
      for (int i = 0; 2 * i < k; i++) { acc += (uint64_t)a2[i] * a.l[k - i]; acc -= (uint64_t)c10[i] * c.l[k - i]; }
      if ((k & 1) == 0) { acc += (uint64_t)a.l[k / 2] * a.l[k / 2]; acc -= (uint64_t)c5[k / 2] * c.l[k / 2]; }
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
      m[k] = mont_step<true>(acc);
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = k - L + 1; 2 * i < k; i++) { acc += (uint64_t)a2[i] * a.l[k - i]; acc -= (uint64_t)c10[i] * c.l[k - i]; }
      if ((k & 1) == 0) { acc += (uint64_t)a.l[k / 2] * a.l[k / 2]; acc -= (uint64_t)c5[k / 2] * c.l[k / 2]; }
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::P[k - i];
      acc += P::P[k - L];  // + p after the division by R keeps the result positive
      r.l[k - L] = (uint32_t)acc & MASK;
      acc = (uint64_t)((int64_t)acc >> W);
    }
    acc += P::P[L - 1];
    r.l[L - 1] = (uint32_t)acc;
    TRK(r.lb = 1; r.vb = 3;)
    return r;
  }
  // ---- four products in one reduction pass: (a b + KC c d - e f - KC g h)/R + p, KC in {-5, +1}: the two halves of an Fp2
  // difference of products a b - c d (Y3 of the curve formulas over Fp2: 8 limb-product sweeps and 2 reductions instead of 4).
  // Inputs normalised; columns: L (1 + |KC| + 1 + |KC| + 1) <= 255 needs L <= 19 (the 14-limb field).
  // SGN (round-3 open finding, kept for the reproducer tools/repro_mul4k.hip): the negative products as signed multiply-adds (nprod)
  // instead of unsigned products subtracted per column.  The two forms are the same function modulo 2^64 per column; the shipped
  // kernels use SGN = false.
  template <int KC, bool SGN = false> HD static Fp mul4k(const Fp& a, const Fp& b, const Fp& c, const Fp& d, const Fp& e, const Fp& f, const Fp& g, const Fp& h) {
    static_assert(KC == 1 || KC == -5, "unsupported multiplier");
    constexpr uint32_t AK = KC < 0 ? (uint32_t)(-KC) : (uint32_t)KC;
    static_assert(L * (2 * AK + 3) <= 255, "column bound");
    TRK(assert(a.lb <= 1 && b.lb <= 1 && c.lb <= 1 && d.lb <= 1 && e.lb <= 1 && f.lb <= 1 && g.lb <= 1 && h.lb <= 1);
        assert(a.vb * b.vb + AK * c.vb * d.vb + e.vb * f.vb + AK * g.vb * h.vb <= 32768.0);
        assert(!a.topwrap && !b.topwrap && !c.topwrap && !d.topwrap && !e.topwrap && !f.topwrap && !g.topwrap && !h.topwrap);)
    Fp r;
    uint32_t m[L], cc[L], gg[L];
#pragma unroll
    for (int i = 0; i < L; i++) { cc[i] = AK == 1 ? c.l[i] : c.l[i] * AK; gg[i] = AK == 1 ? g.l[i] : g.l[i] * AK; }
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        acc += (uint64_t)a.l[i] * b.l[k - i];
        if (SGN) acc += nprod(e.l[i], f.l[k - i]); else acc -= (uint64_t)e.l[i] * f.l[k - i];
        if (KC < 0) { if (SGN) acc += nprod(cc[i], d.l[k - i]); else acc -= (uint64_t)cc[i] * d.l[k - i]; acc += (uint64_t)gg[i] * h.l[k - i]; }
        else { acc += (uint64_t)cc[i] * d.l[k - i]; if (SGN) acc += nprod(gg[i], h.l[k - i]); else acc -= (uint64_t)gg[i] * h.l[k - i]; }
      }
#pragma unroll
      for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
      m[k] = mont_step<true>(acc);
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = k - L + 1; i < L; i++) {
        acc += (uint64_t)a.l[i] * b.l[k - i];
        if (SGN) acc += nprod(e.l[i], f.l[k - i]); else acc -= (uint64_t)e.l[i] * f.l[k - i];
        if (KC < 0) { if (SGN) acc += nprod(cc[i], d.l[k - i]); else acc -= (uint64_t)cc[i] * d.l[k - i]; acc += (uint64_t)gg[i] * h.l[k - i]; }
        else { acc += (uint64_t)cc[i] * d.l[k - i]; if (SGN) acc += nprod(gg[i], h.l[k - i]); else acc -= (uint64_t)gg[i] * h.l[k - i]; }
      }
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::P[k - i];
      acc += P::P[k - L];  // + p after the division by R keeps the result positive
      r.l[k - L] = (uint32_t)acc & MASK;
      acc = (uint64_t)((int64_t)acc >> W);
    }
    acc += P::P[L - 1];
    r.l[L - 1] = (uint32_t)acc;
    TRK(r.lb = 1; r.vb = 3;)
    return r;
  }
  // ---- the curve formulas' interface for operands they have prepared once (curve.h): prep() carries the limbs where this field's
  // products need it (28 limbs) and is the identity where loosely reduced operands fit the column bound anyway (14 limbs); the _nn
  // products then skip the per-call carry pass of mul / sqr / mul_sub - a mixed addition of the 28-limb field ran 13 of them for
  // the 3 operands (the two differences and t) that need one.
  HD static Fp prep(const Fp& a) { return NORM_IN ? norm(a) : a; }
  HD static Fp mul_nn(const Fp& a, const Fp& b) { return mul<false>(a, b); }
  HD static Fp sqr_nn(const Fp& a) { return sqr<false>(a); }
  HD static Fp mul_sub_nn(const Fp& a, const Fp& b, const Fp& c, const Fp& d) { return mul2k<-1>(a, b, c, d); }
  template <int SITE> HD static Fp mul_sub_nn_at(const Fp& a, const Fp& b, const Fp& c, const Fp& d) { return mul_sub_nn(a, b, c, d); }
  // a*b - c*d in ONE reduction pass.  The column bound L (lb_a lb_b + lb_c lb_d + 1) <= 255 holds for the 14-limb field with the
  // loosely reduced operands the curve formulas hand over (lb_a lb_b <= 9, lb_c lb_d <= 1); the 28-limb field first carries its
  // operands (two ~84-instruction passes for the 784 multiply-adds of the reduction saved: 5 % of a mixed addition)
  HD static Fp mul_sub(const Fp& a, const Fp& b, const Fp& c, const Fp& d) {
    if constexpr (L * 12 <= 255) return mul2k<-1>(a, b, c, d);
    else {
      static_assert(L * 3 <= 255, "column bound with carried operands");
      return mul2k<-1>(norm(a), norm(b), norm(c), norm(d));
    }
  }

  // ---- arkworks Montgomery (R = 2^(64*N64), 64-bit limbs) <-> device form
  HD static Fp repack_from64(const uint64_t* s) {
    Fp r;
#pragma unroll
    for (int i = 0; i < L; i++) {
      int bit = i * W;
      int w = bit >> 6, off = bit & 63;
      uint64_t v = (w < P::N64) ? (s[w] >> off) : 0;
      if (off + W > 64 && w + 1 < P::N64) v |= s[w + 1] << (64 - off);
      r.l[i] = (uint32_t)v & MASK;
    }
    TRK(r.lb = 1; r.vb = 2;)
    return r;
  }
  HD void repack_to64(uint64_t* d) const {  // *this canonical & normalised
#pragma unroll
    for (int w = 0; w < P::N64; w++) d[w] = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      int bit = i * W;
      int w = bit >> 6, off = bit & 63;
      if (w < P::N64) {
        d[w] |= (uint64_t)l[i] << off;
        if (off + W > 64 && w + 1 < P::N64) d[w + 1] |= (uint64_t)l[i] >> (64 - off);
      }
    }
  }
  HD static Fp from_ark(const uint64_t* s) { return mul(repack_from64(s), from_limbs(P::C_IN)); }
  HD void to_ark(uint64_t* d) const { reduce(mul(*this, from_limbs(P::C_OUT))).repack_to64(d); }
  // canonical integer (64-bit limbs) <-> device form
  HD static Fp from_canonical(const uint64_t* s) { return mul(repack_from64(s), from_limbs(P::R2)); }
  HD void to_canonical(uint64_t* d) const { reduce(mul(*this, from_limbs(P::RAW_ONE))).repack_to64(d); }

  // ---- exponentiation / inversion (slow paths: host epilogues, final exponentiation easy part)
  HD static Fp pow64(const Fp& a, const uint64_t* e, int nlimbs) {
    Fp r = one();
    bool started = false;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
      if (started) r = sqr(r);
      if ((e[i >> 6] >> (i & 63)) & 1) {
        r = started ? mul(r, a) : a;
        started = true;
      }
    }
    return r;
  }
  // a^-1 (0 for 0): safegcd division steps on the canonical integer (modinv.h) - uniform control flow, ~6x (377 bits) to ~16x
  // (761 bits) fewer instructions than Fermat's a^(p-2), which stays as the cross-check
  HD static Fp inv(const Fp& a);
  HD static Fp inv_fermat(const Fp& a) {
    uint64_t e[P::N64];
#pragma unroll
    for (int i = 0; i < P::N64; i++) e[i] = P::P64[i];
    e[0] -= 2;
    return pow64(norm(a), e, P::N64);
  }
};

typedef Fp<P377> Fq377d;
typedef Fp<P761> Fq761d;

}  // namespace celo
#include "modinv.h"
namespace celo {
template <class P> HD Fp<P> Fp<P>::inv(const Fp& a) {
  uint64_t x[P::N64], y[P::N64];
  norm(a).to_canonical(x);
  SafeGcd<P>::inv(x, y);
  return from_canonical(y);
}
}  // namespace celo
