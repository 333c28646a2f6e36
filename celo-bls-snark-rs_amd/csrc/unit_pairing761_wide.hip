// Translation unit: the single-product LATENCY path of the BW6-761 pairing check (Groth16 `verify`: one product of four pairs,
// crates/epoch-snark/src/api/verifier.rs:35 reached from crates/bls-snark-sys/src/snark/mod.rs:23-45).
//
// The throughput kernels give a product to one lane group, start to finish: for ONE product that is one wave walking 63 + 189
// Miller iterations per pair, then a 449-bit and a 575-bit exponentiation back to back - 65 ms of a 70 ms `verify` (round 2
// profile).  The pieces are independent, so here they run side by side in different WAVES (different loops must not share a
// wave: divergent lanes would run them one after the other):
//   k_wide_miller   block 0: the f_{x+1} loops of all pairs, block 1: the f_{x^3-x^2-x} loops (one lane group per pair and loop)
//   k_wide_easy     one group: f = prod f1_i * frob(prod f2_i), the easy part m = f^((q^3-1)(q+1)), and m^q
//   k_wide_pow      block 0: m^R0, block 1: (m^q)^R1, signed-digit (NAF) ladders: a third fewer multiplications than binary
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

__global__ void __launch_bounds__(64) LANES_OCC k_wide_miller(const uint64_t* __restrict__ g1, const uint8_t* __restrict__ inf1, const uint64_t* __restrict__ g2,
                                                              const uint8_t* __restrict__ inf2, uint32_t* __restrict__ f_out, uint32_t k, uint32_t blocks_per_loop) {
  const int g = LP::QB::group();
  const uint32_t which = blockIdx.x / blocks_per_loop;                       // 0: f1 loops, 1: f2 loops (wave-uniform)
  const uint32_t i = (blockIdx.x % blocks_per_loop) * LP::GROUPS + (uint32_t)g;
  if (g >= LP::GROUPS || i >= k) return;
  const LP::F px = LP::load_p(g1 + (size_t)i * LP::G1W, 0), py = LP::load_p(g1 + (size_t)i * LP::G1W, 1);
  const LP::QB::V Qc = LP::load_q(g2 + (size_t)i * LP::G2W);
  Tow::E12 f = which == 0 ? Pair::miller_f1(px, py, Qc) : Pair::miller_f2(px, py, Qc);
  if ((inf1 && inf1[i]) || (inf2 && inf2[i])) f = Tow::one12();
  LP::store12(f_out + (size_t)(which * k + i) * lanes_gt_words<LP>(), f);
}
// f_in: k f1 values then k f2 values.  out: [0] = m (or the Miller value f itself when !do_fe), [1] = m^q
__global__ void __launch_bounds__(64) LANES_OCC k_wide_easy(const uint32_t* __restrict__ f_in, uint32_t k, uint32_t* __restrict__ out, int do_fe) {
  if (LP::QB::group() != 0) return;
  constexpr int W = lanes_gt_words<LP>();
  Tow::E12 a = LP::load12(f_in), b = LP::load12(f_in + (size_t)k * W);
  for (uint32_t i = 1; i < k; i++) {
    a = Tow::mul12(a, LP::load12(f_in + (size_t)i * W));
    b = Tow::mul12(b, LP::load12(f_in + (size_t)(k + i) * W));
  }
  Tow::E12 f = Tow::mul12(a, Pair::frob1(b));
  if (do_fe) {
    f = Pair::easy_part(f);
    LP::store12(out + W, Pair::frob1(f));
  }
  LP::store12(out, f);
}
__global__ void __launch_bounds__(64) LANES_OCC k_wide_pow(const uint32_t* __restrict__ m_in, const int8_t* __restrict__ naf0, int len0, int neg0,
                                                           const int8_t* __restrict__ naf1, int len1, int neg1, uint32_t* __restrict__ out) {
  if (LP::QB::group() != 0) return;
  constexpr int W = lanes_gt_words<LP>();
  const bool second = blockIdx.x != 0;                                       // wave-uniform
  const Tow::E12 f = LP::load12(m_in + (second ? W : 0));
  LP::store12(out + (second ? W : 0), Pair::pow_naf(f, second ? naf1 : naf0, second ? len1 : len0, second ? neg1 != 0 : neg0 != 0));
}
__global__ void __launch_bounds__(64) LANES_OCC k_wide_final(const uint32_t* __restrict__ p, int do_fe, uint8_t* __restrict__ is_one, uint64_t* __restrict__ gt_ark) {
  if (LP::QB::group() != 0) return;
  constexpr int W = lanes_gt_words<LP>();
  Tow::E12 r = LP::load12(p);
  if (do_fe) r = Tow::mul12(r, LP::load12(p + W));
  const bool one = Tow::is_one12(r);
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

// One product of k pairs, everything enqueued on `s`.  d_f: room for 2k + 4 GT values.  Returns non-zero if the tables cannot be set up.
int wide_product_761(const uint64_t* d_g1, const uint8_t* d_i1, const uint64_t* d_g2, const uint8_t* d_i2, uint32_t k, uint32_t* d_f, uint8_t* d_one,
                     uint64_t* d_gt, int do_fe, hipStream_t s) {
  NafTables t;
  if (naf_tables(t)) return 1;
  constexpr int W = lanes_gt_words<LP>();
  const uint32_t bpl = (k + LP::GROUPS - 1) / LP::GROUPS;
  uint32_t* d_m = d_f + (size_t)2 * k * W;      // m, m^q
  uint32_t* d_p = d_m + 2 * W;                  // m^R0, (m^q)^R1
  hipLaunchKernelGGL(k_wide_miller, dim3(2 * bpl), dim3(64), 0, s, d_g1, d_i1, d_g2, d_i2, d_f, k, bpl);
  hipLaunchKernelGGL(k_wide_easy, dim3(1), dim3(64), 0, s, d_f, k, d_m, do_fe);
  if (do_fe) hipLaunchKernelGGL(k_wide_pow, dim3(2), dim3(64), 0, s, d_m, t.d0, t.len0, T761::R0_NEG ? 1 : 0, t.d1, t.len1, T761::R1_NEG ? 1 : 0, d_p);
  hipLaunchKernelGGL(k_wide_final, dim3(1), dim3(64), 0, s, do_fe ? d_p : d_m, do_fe, d_one, d_gt);
  return 0;
}
}  // namespace celo
