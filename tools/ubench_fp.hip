// Field-arithmetic throughput microbenchmark (gfx950): how close do the real multiply bodies get to the
// v_mad_u64_u32 issue peak, and does the size of the straight-line loop body (instruction cache) matter?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../celo-bls-snark-rs_amd/csrc/curve.h"
#include "../celo-bls-snark-rs_amd/csrc/fp2.h"
using namespace celo;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef Fp<P377> F;

template <int MODE> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k4(uint32_t* out, int iters, int desync) {
  F a = F::one(), b = F::one();
  a.l[0] += threadIdx.x; b.l[1] += blockIdx.x & 0xff;
  Xyzz<F> acc = Xyzz<F>::from_affine({a, b});
  Affine<F> p = {b, a};
  acc.ZZ = a; acc.ZZZ = b;
  if (desync) {  // stagger the waves: each wave idles a different amount before entering the loop
    unsigned w = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u;
    unsigned spins = (w >> 20) & 0xfff;
    for (unsigned i = 0; i < spins; i++) __builtin_amdgcn_s_sleep(8);
  }
  for (int i = 0; i < iters; i++) { xyzz_madd(acc, p); p.x.l[0] ^= acc.X.l[3] & 1; }
  a = F::add(acc.X, acc.Y);
  uint32_t r = 0;
  for (int i = 0; i < F::L; i++) r ^= a.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int PREFETCH> __global__ void __launch_bounds__(256) kg(uint32_t* out, const uint32_t* __restrict__ pts, uint32_t npts, int iters) {
  F a = F::one(), b = F::one();
  a.l[0] += threadIdx.x; b.l[1] += blockIdx.x & 0xff;
  Xyzz<F> acc = Xyzz<F>::from_affine({a, b});
  acc.ZZ = a; acc.ZZZ = b;
  uint32_t h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
  if (PREFETCH) {
    Affine<F> nxt = {F::load(pts + (size_t)(h % npts) * 32), F::load(pts + (size_t)(h % npts) * 32 + 16)};
    for (int i = 0; i < iters; i++) {
      Affine<F> p = nxt;
      h = h * 1664525u + 1013904223u;
      const uint32_t* q = pts + (size_t)(h % npts) * 32;
      nxt = {F::load(q), F::load(q + 16)};
      xyzz_madd(acc, p);
    }
  } else {
    for (int i = 0; i < iters; i++) {
      h = h * 1664525u + 1013904223u;
      const uint32_t* q = pts + (size_t)(h % npts) * 32;
      Affine<F> p = {F::load(q), F::load(q + 16)};
      xyzz_madd(acc, p);
    }
  }
  a = F::add(acc.X, acc.Y);
  uint32_t r = 0;
  for (int i = 0; i < F::L; i++) r ^= a.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> __global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
  F a = F::one(), b = F::one();
  a.l[0] += threadIdx.x; b.l[1] += blockIdx.x & 0xff;
  if (MODE == 0) {            // 1 mul per iteration (~600-instruction body)
    for (int i = 0; i < iters; i++) a = F::mul(a, b);
  } else if (MODE == 1) {     // 10 muls per iteration, all different code copies (~6000-instruction body)
    F c = b, d = a;
    for (int i = 0; i < iters; i++) {
      a = F::mul(a, b); c = F::mul(c, a); d = F::mul(d, c); b = F::mul(b, d); a = F::mul(a, c);
      c = F::mul(c, d); d = F::mul(d, b); b = F::mul(b, a); a = F::mul(a, d); c = F::mul(c, b);
    }
    a = F::add(a, F::add(c, d));
  } else if (MODE == 2) {     // full mixed add on register-resident data
    Xyzz<F> acc = Xyzz<F>::from_affine({a, b});
    Affine<F> p = {b, a};
    acc.ZZ = a; acc.ZZZ = b;
    for (int i = 0; i < iters; i++) { xyzz_madd(acc, p); p.x.l[0] ^= acc.X.l[3] & 1; }
    a = F::add(acc.X, acc.Y);
  } else if (MODE == 3) {     // sqr only
    for (int i = 0; i < iters; i++) a = F::sqr(a);
  }
  uint32_t r = 0;
  for (int i = 0; i < F::L; i++) r ^= a.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> int run(const char* name, double muls_per_iter, int iters, uint32_t* dout, int cus) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double total_muls = (double)blocks * 256 * iters * muls_per_iter;
    printf("%-22s waves/SIMD(req)=%d  %8.3f ms  %7.2f G fp-mul/s  (%.1f ns per wave-mul per SIMD)\n", name, wps, ms, total_muls / ms / 1e6,
           ms * 1e6 / ((double)iters * muls_per_iter * wps));
  }
  return 0;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  uint32_t* dout; CK(hipMalloc(&dout, (size_t)cus * 8 * 256 * 4));
  run<0>("mul x1 loop", 1, 4000, dout, cus);
  run<3>("sqr x1 loop", 1, 4000, dout, cus);
  run<1>("mul x10 loop", 10, 400, dout, cus);
  run<2>("xyzz_madd loop", 10, 400, dout, cus);
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int desync = 0; desync < 2; desync++) for (int wps : {4, 8}) {
      int blocks = cus * wps, iters = 400;
      hipLaunchKernelGGL(k4<0>, dim3(blocks), dim3(256), 0, 0, dout, iters, desync);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k4<0>, dim3(blocks), dim3(256), 0, 0, dout, iters, desync);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("madd loop 4waves/EU desync=%d blocks/CU=%d  %8.3f ms  %7.2f G fp-mul-equiv/s\n", desync, wps, ms, (double)blocks * 256 * iters * 10 / ms / 1e6);
    }
  }
  {
    uint32_t npts = 1u << 20; uint32_t* pts; CK(hipMalloc(&pts, (size_t)npts * 128));
    CK(hipMemset(pts, 0x11, (size_t)npts * 128));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pf = 0; pf < 2; pf++) for (int wps : {4, 8}) {
      int blocks = cus * wps, iters = 400;
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        if (pf) hipLaunchKernelGGL(kg<1>, dim3(blocks), dim3(256), 0, 0, dout, pts, npts, iters);
        else hipLaunchKernelGGL(kg<0>, dim3(blocks), dim3(256), 0, 0, dout, pts, npts, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("madd+gather prefetch=%d blocks/CU=%d  %8.3f ms  %7.2f G fp-mul-equiv/s\n", pf, wps, ms, (double)blocks * 256 * iters * 10 / ms / 1e6);
    }
  }
  return 0;
}
