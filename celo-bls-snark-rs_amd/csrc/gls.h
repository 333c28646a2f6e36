// Base-x digits of a scalar (x = 0x8508c00000000001, the BLS12-377 curve parameter): the integer side of the GLS split of the batched
// G2 MSM (msm.h: k_gls_expand), host + device so that this integer code is checked on the CPU (tests/test_host_field.py).
#pragma once
#include <cstdint>
#include "fp.h"

namespace celo {
// u (NU words + one zero word on top) <- u mod x in u[0], u[1];  q (NU - 1 words) <- floor(u / x).  Knuth's algorithm D in base 2^32
// with the two-digit divisor (x >> 32, x & 0xffffffff) = (0x8508c000, 1), already normalised (top bit set).  Every bound is a
// compile-time constant: the arrays are registers on the device.
template <int NU> HD void gls_divmod_x(uint32_t (&u)[NU + 1], uint32_t (&q)[NU - 1]) {
  constexpr uint32_t V1 = 0x8508c000u, V0 = 0x00000001u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = NU - 2; j >= 0; j--) {
    const uint64_t num = ((uint64_t)u[j + 2] << 32) | u[j + 1];
    uint64_t qhat = num / V1, rhat = num % V1;
    while (qhat >= (uint64_t(1) << 32) || qhat * V0 > ((rhat << 32) | u[j])) {
      qhat--;
      rhat += V1;
      if (rhat >= (uint64_t(1) << 32)) break;
    }
    // u[j .. j+2] -= qhat * (V1 2^32 + V0)
    const uint64_t p0 = qhat * V0, p1 = qhat * V1;
    int64_t t = (int64_t)u[j] - (int64_t)(p0 & 0xffffffffu);
    u[j] = (uint32_t)t;
    int64_t borrow = t >> 32;                                        // 0 or -1
    t = (int64_t)u[j + 1] - (int64_t)(p0 >> 32) - (int64_t)(p1 & 0xffffffffu) + borrow;
    u[j + 1] = (uint32_t)t;
    borrow = t >> 32;
    t = (int64_t)u[j + 2] - (int64_t)(p1 >> 32) + borrow;
    u[j + 2] = (uint32_t)t;
    if (t < 0) {                                                     // qhat was one too large: add the divisor back
      qhat--;
      uint64_t c = (uint64_t)u[j] + V0;
      u[j] = (uint32_t)c;
      c = (uint64_t)u[j + 1] + V1 + (c >> 32);
      u[j + 1] = (uint32_t)c;
      u[j + 2] += (uint32_t)(c >> 32);
    }
    q[j] = (uint32_t)qhat;
  }
}
// k (NW significant words) = d[0] + d[1] x + .. + d[ND-1] x^(ND-1): ND - 1 divisions, the last digit is what is left (the caller
// picks ND so that it fits 64 bits: ND = 2 for k < 2^126, 3 for k < 2^189, 4 for k < 2^253; it is below x whenever k < x^ND).
template <int NW, int ND, int J = 0> HD void gls_digits_rec(uint32_t (&u)[NW + 1], uint32_t d[4][2]) {
  if constexpr (J == ND - 1 || NW < 2) {
    d[J][0] = u[0];
    d[J][1] = NW >= 2 ? u[NW >= 2 ? 1 : 0] : 0u;
    for (int j = J + 1; j < 4; j++) { d[j][0] = 0; d[j][1] = 0; }
  } else {
    uint32_t q[NW - 1];
    gls_divmod_x<NW>(u, q);
    d[J][0] = u[0]; d[J][1] = u[1];
    uint32_t v[NW];
    for (int i = 0; i < NW - 1; i++) v[i] = q[i];
    v[NW - 1] = 0;
    gls_digits_rec<NW - 1, ND, J + 1>(v, d);
  }
}
template <int NW, int ND> HD void gls_digits_base_x(const uint32_t* k, uint32_t d[4][2]) {
  uint32_t u[NW + 1];
  for (int i = 0; i < NW; i++) u[i] = k[i];
  u[NW] = 0;
  gls_digits_rec<NW, ND>(u, d);
}
// GLV split for G1 of BLS12-377: k = k0 + k1 x^2 with 0 <= k0 < x^2 < 2^127 and k1 = floor(k / x^2) < 2^127 for k < r (two divisions by
// x: k = r1 + x q1, q1 = r2 + x q2, so k0 = r1 + x r2 and k1 = q2).  [k]P = [k0]P + [k1]([x^2]P) and [x^2]P = (beta x, -y) on the subgroup.
template <int NW> HD void glv_split_x2(const uint32_t* k, uint32_t k0[4], uint32_t k1[4]) {
  static_assert(NW >= 5 && NW <= 8, "scalar words");
  uint32_t u[NW + 1], q1[NW - 1];
  for (int i = 0; i < NW; i++) u[i] = k[i];
  u[NW] = 0;
  gls_divmod_x<NW>(u, q1);
  const uint32_t r1l = u[0], r1h = u[1];
  uint32_t v[NW], q2[NW - 2];
  for (int i = 0; i < NW - 1; i++) v[i] = q1[i];
  v[NW - 1] = 0;
  gls_divmod_x<NW - 1>(v, q2);
  const uint32_t r2l = v[0], r2h = v[1];
  for (int i = 0; i < 4; i++) k1[i] = i < NW - 2 ? q2[i] : 0u;
  // k0 = r1 + x r2,  x = 0x8508c000 2^32 + 1:  x r2 = r2 + (0x8508c000 r2) 2^32
  constexpr uint64_t V1 = 0x8508c000u;
  const uint64_t p0 = V1 * r2l, p1 = V1 * r2h;                 // at words 1 and 2
  uint64_t c = (uint64_t)r1l + r2l;
  k0[0] = (uint32_t)c;
  c = (c >> 32) + (uint64_t)r1h + r2h + (uint32_t)p0;
  k0[1] = (uint32_t)c;
  c = (c >> 32) + (p0 >> 32) + (uint32_t)p1;
  k0[2] = (uint32_t)c;
  c = (c >> 32) + (p1 >> 32);
  k0[3] = (uint32_t)c;
}
}  // namespace celo
