// Fixed-base MSM: per-key tables T[j][i] = 2^(cf j) P_i for the Groth16 prover's queries, virtual windows, compacted digits.
// (part of the MSM pipeline: csrc/msm.h includes the pieces in order and carries the overview)
#pragma once

namespace celo {

// =====================================================================================================================
// FIXED-BASE MSM: the Groth16 prover hands the SAME Parameters to every proof (crates/epoch-snark/src/api/prover.rs:78,112;
// they are created once, crates/epoch-snark/src/api/setup.rs:63-105), so its queries can carry per-key tables T[j][i] = 2^(cf j) P_i.  Then
//   sum_i k_i P_i = sum_i sum_j d_ij T[j][i]      (d_ij the signed cf-bit digits of k_i)
// and ALL n W digit entries fall into ONE set of 2^(cf - 1) buckets: one bucket reduction instead of W, no Horner chain over the windows,
// and cf is free to grow beyond the 16 bits of the variable-base windows (fewer digits per scalar = fewer additions: 19 instead of 24 for
// the 377-bit scalars at cf = 20).  The pipeline below the digits is the variable-base one, unchanged: the table is handed to it as E = n W
// bases, and the 2^(cf - 1) buckets as NV VIRTUAL windows of M buckets each (bucket b = v M + low) - entry p = j n + i carries its digit in
// the virtual window it belongs to and a zero digit in the others.  M = 2^15 - 1 for cf > 16: the pipeline's 16-bit digit is 15 bits of
// bucket + the sign + the value 0xFFFF for "no digit", and (low = 0x7FFF, negative) IS 0xFFFF - with cf = 16 a negative digit never
// reaches that bucket, with virtual windows it does (caught by the scalar r - 1 at cf = 19), so the last bucket of every virtual window
// stays empty and NV = floor((2^(cf-1) - 1) / M) + 1.  What changes is the end:
//   total = sum_v [ S_v + v M T_v ],   S_v = the window's weighted sum (node + sum_l 2^(15 - l) O_l),   T_v = its plain sum (node),
// one Horner pass of ~15 + log2(NV) doublings on the host (run_device_windows, fx branch): the bits of v M from the top, the plain sums
// of the windows that have the bit added at each step, the levels' O_l joining in from bit 14 down.
struct FixedTable {
  uint32_t* table = nullptr;      // E affine points, device form, entry j n + i = 2^(cf j) P_i
  uint8_t* tinf = nullptr;        // E flags: the entry is the identity (a flagged base; a base whose 2^(cf j) multiple is the identity)
  uint32_t n = 0, W = 0, NV = 0, M = 0;      // M: buckets used per virtual window (the divisor of the bucket index)
  int cf = 0, device = 0;
  size_t bytes = 0;
  float build_ms = 0;
  uint32_t E() const { return n * W; }
};
// T[j] from T[j - 1]: cf doublings and one inversion per point (the inversion is ~20 % of the lane's work: no batching needed for a
// table that is built once per key)
template <class G>
__global__ void __launch_bounds__(128) k_fixed_next(const uint32_t* __restrict__ prev, const uint8_t* __restrict__ pinf, uint32_t* __restrict__ next,
                                                    uint8_t* __restrict__ ninf, uint32_t n, int cf) {
  typedef typename G::F F;
  typedef PointIO<F> IO;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> r = {F::zero(), F::zero()};
  uint8_t fl = 1;
  if (!pinf[i]) {
    Xyzz<F> a = Xyzz<F>::from_affine(IO::load_affine(prev + (size_t)i * IO::AFF_WORDS));
    for (int k = 0; k < cf; k++) xyzz_dbl_fn(a);
    if (!a.is_identity() && !a.ZZ.is_zero_mod_p()) {
      const F t = F::inv(F::mul(a.ZZ, a.ZZZ));          // x = X / ZZ, y = Y / ZZZ with one inversion
      r = {F::mul(a.X, F::mul(t, a.ZZZ)), F::mul(a.Y, F::mul(t, a.ZZ))};
      fl = 0;
    }
  }
  IO::store_affine(next + (size_t)i * IO::AFF_WORDS, r);
  ninf[i] = fl;
}
template <class G>
__global__ void __launch_bounds__(256) k_fixed_first_flags(const uint8_t* __restrict__ inf, uint8_t* __restrict__ tinf, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tinf[i] = inf ? (inf[i] ? 1 : 0) : 0;
}
// signed cf-bit digits of the first n_sc scalars (a shorter scalar list leaves the remaining bases out: VariableBaseMSM zips), written
// per virtual window: digits[v E + j n + i] = the low 15 bits of (|d| - 1) | sign << 15 if the entry belongs to v, else the zero digit.
template <int SW, int BITS>
__global__ void __launch_bounds__(256) k_fixed_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ tinf, uint16_t* __restrict__ digits,
                                                      uint32_t n, uint32_t n_sc, int cf, uint32_t W, uint32_t NV, uint32_t M) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t E = (size_t)n * W;
  uint32_t s[SW + 1];
  if (i < n_sc) {
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * SW);
#pragma unroll
    for (int k = 0; k < SW / 4; k++) { const uint4 v = sp[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
  } else {
#pragma unroll
    for (int k = 0; k < SW; k++) s[k] = 0;
  }
  s[SW] = 0;
  if constexpr (BITS < 32 * SW) {          // bits from the scalar length up are not part of the scalar (k_digits, ark-ec)
    s[BITS / 32] &= (1u << (BITS % 32)) - 1u;
#pragma unroll
    for (int k = BITS / 32 + 1; k < SW; k++) s[k] = 0;
  }
  const uint32_t half = 1u << (cf - 1);
  uint32_t carry = 0;
  for (uint32_t j = 0; j < W; j++) {
    const uint32_t bit = j * (uint32_t)cf, wi = bit >> 5, off = bit & 31;
    uint32_t raw = 0;
    if (wi < (uint32_t)SW) {
      uint32_t w0 = 0, w1 = 0;
#pragma unroll
      for (int k = 0; k <= SW; k++) { if ((uint32_t)k == wi) w0 = s[k]; if ((uint32_t)k == wi + 1) w1 = s[k]; }
      raw = (uint32_t)((((uint64_t)w1 << 32) | w0) >> off) & ((1u << cf) - 1u);
    }
    const uint32_t d = raw + carry;
    const uint32_t neg = d > half ? 1u : 0u;
    const uint32_t mag = neg ? (1u << cf) - d : d;
    carry = neg;
    const size_t p = (size_t)j * n + i;
    const bool live = mag != 0 && !tinf[p];
    const uint32_t full = mag - 1u, v = full / M;
    const uint16_t dg = (uint16_t)((full - v * M) | (neg << 15));
    for (uint32_t vv = 0; vv < NV; vv++) digits[(size_t)vv * E + p] = (live && vv == v) ? dg : (uint16_t)0xFFFF;
  }
}

// ---- the same digits COMPACTED BY VIRTUAL WINDOW (round 4).  k_fixed_digits hands the pipeline NV windows of E slots each, all but one of
// an entry's slots holding "no digit": at cf = 20 the two partition passes and the digit kernel move 17 x 4 10^7 x 2 bytes three times and
// the sort costs 1.7 ms where the variable-base sort of as many real entries costs 0.45.  Here every entry gets ONE record (window id,
// digit), the records are placed window by window - window v's entries in slots [0, count_v) of a row of Ep >= max_v count_v slots, the
// rest padded with "no digit" - and the partition pass carries the table index of a slot along (k_part_scatter's remap): the pipeline
// then sees NV windows of Ep ~ E / (NV - 1) slots.  The order of a window's slots is whatever the atomics give; sums do not care.
template <int SW, int BITS>
__global__ void __launch_bounds__(256) k_fixed_digits_c(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ tinf, uint8_t* __restrict__ v8,
                                                        uint16_t* __restrict__ dg16, uint32_t* __restrict__ counts, uint32_t n, uint32_t n_sc, int cf, uint32_t W,
                                                        uint32_t NV, uint32_t M) {
  __shared__ uint32_t lc[128];
  if (threadIdx.x < 128) lc[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint32_t s[SW + 1];
    if (i < n_sc) {
      const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * SW);
#pragma unroll
      for (int k = 0; k < SW / 4; k++) { const uint4 v = sp[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
    } else {
#pragma unroll
      for (int k = 0; k < SW; k++) s[k] = 0;
    }
    s[SW] = 0;
    if constexpr (BITS < 32 * SW) {
      s[BITS / 32] &= (1u << (BITS % 32)) - 1u;
#pragma unroll
      for (int k = BITS / 32 + 1; k < SW; k++) s[k] = 0;
    }
    const uint32_t half = 1u << (cf - 1);
    uint32_t carry = 0;
    for (uint32_t j = 0; j < W; j++) {
      const uint32_t bit = j * (uint32_t)cf, wi = bit >> 5, off = bit & 31;
      uint32_t raw = 0;
      if (wi < (uint32_t)SW) {
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int k = 0; k <= SW; k++) { if ((uint32_t)k == wi) w0 = s[k]; if ((uint32_t)k == wi + 1) w1 = s[k]; }
        raw = (uint32_t)((((uint64_t)w1 << 32) | w0) >> off) & ((1u << cf) - 1u);
      }
      const uint32_t d = raw + carry;
      const uint32_t neg = d > half ? 1u : 0u;
      const uint32_t mag = neg ? (1u << cf) - d : d;
      carry = neg;
      const size_t p = (size_t)j * n + i;
      const bool live = mag != 0 && !tinf[p];
      const uint32_t full = mag - 1u, v = full / M;
      v8[p] = live ? (uint8_t)v : (uint8_t)0xFF;
      dg16[p] = (uint16_t)((full - v * M) | (neg << 15));
      if (live) atomicAdd(&lc[v], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < NV && lc[threadIdx.x]) atomicAdd(counts + threadIdx.x, lc[threadIdx.x]);
}
// record p -> slot of its window's row: ranks within a batch by LDS atomics, one global reservation per (batch, window)
template <class G>   // (a template only so that every translation unit including this header may hold a copy)
__global__ void __launch_bounds__(1024) k_fixed_place(const uint8_t* __restrict__ v8, const uint16_t* __restrict__ dg16, uint32_t* __restrict__ cursor,
                                                      uint16_t* __restrict__ digits, uint32_t* __restrict__ remap, uint32_t E, uint32_t Ep, uint32_t NV) {
  __shared__ uint32_t lc[128], lb[128];
  for (uint32_t p0 = blockIdx.x * 1024u; p0 < E; p0 += gridDim.x * 1024u) {
    if (threadIdx.x < 128) lc[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t p = p0 + threadIdx.x;
    const uint32_t v = p < E ? v8[p] : 0xFFu;
    uint32_t r = 0;
    if (v != 0xFFu) r = atomicAdd(&lc[v], 1u);
    __syncthreads();
    if (threadIdx.x < NV && lc[threadIdx.x]) lb[threadIdx.x] = atomicAdd(cursor + threadIdx.x, lc[threadIdx.x]);
    __syncthreads();
    if (v != 0xFFu) {
      const size_t at = (size_t)v * Ep + lb[v] + r;
      digits[at] = dg16[p];
      remap[at] = p;
    }
    __syncthreads();
  }
}

}  // namespace celo
