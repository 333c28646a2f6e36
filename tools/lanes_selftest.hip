// Device self-test of the lane-parallel pairing pieces against the one-lane twins (both on the GPU).
// build: make -C celo-bls-snark-rs_amd/csrc ../build/lanes_selftest
#include "pairing_lanes_kernels.h"
#include <vector>
#include <cstring>
using namespace celo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 99; } } while (0)
#ifdef SELFTEST_HEX
typedef LPH377 LPX;          // six lanes per pairing
#else
typedef LP377 LPX;           // three lanes per pairing
#endif
typedef LPX::QB QBX;
typedef LPX::Pair QPair;
typedef LPX::Tow QTow;
constexpr int NLANES = LPX::GROUPS == 21 ? 3 : 6;
#define lanes_load LPX::load12
#define lanes_store LPX::store12

// op 0: mul12, 1: sqr12, 2: cyclotomic, 3: mul_by_034 (s from y's first three coefficients), 4: inv12, 5: frob1, 6: conj, 7: identity (load/store)
__global__ void k_lanes_op(int op, const uint32_t* x, const uint32_t* y, uint32_t* out) {
  QTow::E12 a = lanes_load(x), b = lanes_load(y), r;
  switch (op) {
    case 0: r = QTow::mul12(a, b); break;
    case 1: r = QTow::sqr12(a); break;
    case 2: r = QTow::cyclotomic_sqr(a); break;
    case 3: { QBX::V s0 = LPX::load_v(y), s3 = LPX::load_v(y + 32), s4 = LPX::load_v(y + 64); r = a; QTow::mul_by_034(r, s0, s3, s4); } break;
    case 4: r = QTow::inv12(a); break;
    case 5: r = QPair::frob12<1>(a); break;
    case 6: r = QTow::conj12(a); break;
    default: r = a; break;
  }
  lanes_store(out, r);
}
__global__ void k_lane_op(int op, const uint32_t* x, const uint32_t* y, uint32_t* out) {
  Fq12 a = f12_load(x), b = f12_load(y), r;
  switch (op) {
    case 0: f12_mul(r, a, b); break;
    case 1: f12_sqr(r, a); break;
    case 2: f12_cyclotomic_sqr(r, a); break;
    case 3: { r = a; f12_mul_by_034(r, b.c0.c0, b.c0.c1, b.c0.c2); } break;
    case 4: f12_inv(r, a); break;
    case 5: f12_frob<1>(r, a); break;
    case 6: r = f12_conj(a); break;
    default: r = a; break;
  }
  // canonicalise for comparison
  uint64_t tmp[72];
  f12_to_ark(r, tmp);
  for (int i = 0; i < 72; i++) ((uint64_t*)out)[i] = tmp[i];
}
__global__ void k_lanes_canon(const uint32_t* in, uint64_t* out) {
  QTow::E12 r = lanes_load(in);
  LPX::to_ark12(r, out);
}
// point steps: in: R (3 Fq2 at x, x+32, x+64), Q (y, y+32); out: R' (3 Fq2) then line (3 Fq2) as ark u64 (6*12)
__global__ void k_lanes_step(int add, const uint32_t* x, const uint32_t* y, uint64_t* out) {
  int q = QBX::lane();
  QBX::V Rc = LPX::load_v(x + 32 * q);
  QBX::V Qc = LPX::load_v(y + 32 * (q & 1));
  QPair::Line l;
  if (add) QPair::add_step(Rc, Qc, l); else QPair::double_step(Rc, l);
  LPX::to_ark_v(Rc, out + 12 * q);
  if (q == 0) { LPX::to_ark_v(l.c0, out + 36); LPX::to_ark_v(l.c1, out + 48); LPX::to_ark_v(l.c2, out + 60); }
}
__global__ void k_lane_step(int add, const uint32_t* x, const uint32_t* y, uint64_t* out) {
  G2Proj r = {Fq2::load(x), Fq2::load(x + 32), Fq2::load(x + 64)};
  Fq2 qx = Fq2::load(y), qy = Fq2::load(y + 32);
  Ell l;
  if (add) pairing_add_step(r, qx, qy, l); else pairing_double_step(r, l);
  r.x.to_ark(out); r.y.to_ark(out + 12); r.z.to_ark(out + 24);
  l.c0.to_ark(out + 36); l.c1.to_ark(out + 48); l.c2.to_ark(out + 60);
}
__global__ void k_fill(uint32_t* x, uint64_t seed) {  // six pseudo-random Fq2 (normalised 28-bit limbs below p's top limb)
  for (int c = 0; c < 6; c++)
    for (int h = 0; h < 2; h++) {
      for (int i = 0; i < 16; i++) {
        seed = seed * 6364136223846793005ULL + 1442695040888963407ULL;
        uint32_t v = (uint32_t)(seed >> 33) & 0x0FFFFFFF;
        if (i == 13) v &= 0xFFF;
        if (i >= 14) v = 0;
        x[c * 32 + h * 16 + i] = v;
      }
    }
}
// truncated Miller loops: `iters` top iterations of the loop, lane-parallel vs one-lane
__global__ void k_lanes_miller(int iters, const uint32_t* x, const uint32_t* y, uint32_t* out) {
  int q = QBX::lane();
  Fq px = Fq::load(x), py = Fq::load(x + 16);
  QBX::V Qc = LPX::load_v(y + 32 * (q & 1));
  QBX::V Rc = QBX::sel<2>(QBX::one(), Qc);
  QTow::E12 f = QTow::one12();
  QPair::Line l;
#pragma unroll 1
  for (int i = 62; i > 62 - iters; i--) {
    f = QTow::sqr12(f);
    QPair::step_double(Rc, f, px, py);
    if ((T377::X >> i) & 1) QPair::step_add(Rc, Qc, f, px, py);
  }
  lanes_store(out, f);
}
__global__ void k_lane_miller(int iters, const uint32_t* x, const uint32_t* y, uint64_t* out) {
  Fq px = Fq::load(x), py = Fq::load(x + 16);
  Fq2 qx = Fq2::load(y), qy = Fq2::load(y + 32);
  G2Proj r = {qx, qy, Fq2::one()};
  Fq12 f = f12_one();
  Ell l;
#pragma unroll 1
  for (int i = 62; i > 62 - iters; i--) {
    Fq12 t; f12_sqr(t, f); f = t;
    pairing_double_step(r, l); pairing_ell(f, l, px, py);
    if ((T377::X >> i) & 1) { pairing_add_step(r, qx, qy, l); pairing_ell(f, l, px, py); }
  }
  f12_to_ark(f, out);
}
int main() {
  uint32_t *x, *y, *oq;
  uint64_t *c1, *c2;
  CK(hipMalloc(&x, 192 * 4)); CK(hipMalloc(&y, 192 * 4)); CK(hipMalloc(&oq, 192 * 4)); CK(hipMalloc(&c1, 72 * 8)); CK(hipMalloc(&c2, 72 * 8));
  k_fill<<<1, 1>>>(x, 12345); k_fill<<<1, 1>>>(y, 999);
  int bad = 0;
  const char* names[] = {"mul12", "sqr12", "cyclotomic", "mul_by_034", "inv12", "frob1", "conj", "identity"};
  for (int op = 0; op < 8; op++) {
    k_lanes_op<<<1, NLANES>>>(op, x, y, oq);
    k_lanes_canon<<<1, NLANES>>>(oq, c1);
    k_lane_op<<<1, 1>>>(op, x, y, (uint32_t*)c2);
    std::vector<uint64_t> a(72), b(72);
    CK(hipMemcpy(a.data(), c1, 576, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), c2, 576, hipMemcpyDeviceToHost));
    int ok = memcmp(a.data(), b.data(), 576) == 0;
    printf("%-12s %s\n", names[op], ok ? "ok" : "MISMATCH");
    if (!ok) { bad++; for (int c = 0; c < 6; c++) printf("   coeff %d: %s\n", c, memcmp(a.data() + 12 * c, b.data() + 12 * c, 96) ? "diff" : "same"); }
  }
  for (int add = 0; add < 2; add++) {
    k_lanes_step<<<1, NLANES>>>(add, x, y, c1);
    k_lane_step<<<1, 1>>>(add, x, y, c2);
    std::vector<uint64_t> a(72), b(72);
    CK(hipMemcpy(a.data(), c1, 576, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), c2, 576, hipMemcpyDeviceToHost));
    int ok = memcmp(a.data(), b.data(), 576) == 0;
    printf("%-12s %s\n", add ? "add_step" : "double_step", ok ? "ok" : "MISMATCH");
    if (!ok) { bad++; const char* nm[] = {"X", "Y", "Z", "l.c0", "l.c1", "l.c2"}; for (int c = 0; c < 6; c++) printf("   %s: %s\n", nm[c], memcmp(a.data() + 12 * c, b.data() + 12 * c, 96) ? "diff" : "same"); }
  }
  for (int iters : {1, 2, 3, 5, 8, 63}) {
    k_lanes_miller<<<1, NLANES>>>(iters, x, y, oq);
    k_lanes_canon<<<1, NLANES>>>(oq, c1);
    k_lane_miller<<<1, 1>>>(iters, x, y, c2);
    std::vector<uint64_t> a(72), b(72);
    CK(hipMemcpy(a.data(), c1, 576, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), c2, 576, hipMemcpyDeviceToHost));
    int ok = memcmp(a.data(), b.data(), 576) == 0;
    printf("miller[%2d]   %s\n", iters, ok ? "ok" : "MISMATCH");
    if (!ok) { bad++; for (int c = 0; c < 6; c++) printf("   coeff %d: %s\n", c, memcmp(a.data() + 12 * c, b.data() + 12 * c, 96) ? "diff" : "same"); }
  }
  printf("hip status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return bad;
}
