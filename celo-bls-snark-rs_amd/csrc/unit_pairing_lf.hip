// Translation unit: lane-parallel (six lanes per pairing) BLS12-377 final exponentiation kernel (pairing_lanes.h).
#include "pairing_lanes_kernels.h"
namespace celo {
void final_exp_w3_377(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, hipStream_t s);   // unit_pairing377_wide.hip
void final_exp_w2_377(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, hipStream_t s);
// up to 3072 products (three per wave: at most one wave per SIMD): every Fq12 operation's independent products side by side - measured
// 1.47 against 2.17 ms at 2048 products.  Above, SIMDs take a second wave and the side-by-side form's extra instructions cost more than
// its shorter chains save (4096 products: 2.39 against 2.24 ms; 6144: 2.50 against 2.31): one six-lane group per product from there on
void LaneLaunch377::final_exp(const uint32_t* prod, uint8_t* is_one, uint64_t* gt, uint32_t m, int do_fe, hipStream_t s) {
  constexpr bool no_w3 = false;       // (round 4's A/B switch CELO_NO_W3_FINAL is gone: the thresholds below are the measured ones)
  if (do_fe && m <= 3072 && !no_w3) { final_exp_w3_377(prod, is_one, gt, m, s); return; }
  // ... up to 5120 on TWO groups per product, five products per wave: still one wave per SIMD (config 3's 4096 verdicts)
  if (do_fe && m <= 5120 && !no_w3) { final_exp_w2_377(prod, is_one, gt, m, s); return; }
  hipLaunchKernelGGL((k_final_exp_slots<LPH377>), dim3((m + LPH377::GROUPS - 1) / LPH377::GROUPS), dim3(64), Slots<LPH377>::LDS_BYTES, s, prod, is_one, gt, m, do_fe);
}
}
