// Translation unit: the multiplier roofline, measured in the process that reports it (celo_amd_ubench_fp).
//
// bench.py's `valu_roofline` prices the dominant kernels against the chip-wide rate of the library's OWN field-product bodies in a
// register-resident loop - the instruction stream k_accumulate spends 82 % of its issue slots on.  Rounds 1-4 took
// that rate from tools/ubench_fp.hip run on some earlier box (78 G mul/s, 94 G sqr/s for the 377-bit field); the kernels of a bench run
// clock at whatever the box and its power state give (2.0 GHz under the accumulation in profiles/r4_summary.md), so a peak from another
// box at another clock is not this run's peak.  This entry point runs the same loops - ten dependent-free products per iteration, eight
// waves per SIMD requested - on the calling thread's device and returns what they measure, together with the shader clock the loop ran
// at (s_memtime ticks against the constant 100 MHz s_memrealtime, taken by every wave around its loop).
#include <hip/hip_runtime.h>
#include <vector>
#include "fp.h"
#include "runtime.h"

namespace celo {

template <class F, int MODE>
__global__ void __launch_bounds__(256) k_ubench_fp(uint32_t* __restrict__ out, uint64_t* __restrict__ ticks, int iters) {
  F a = F::one(), b = F::one();
  a.l[0] += threadIdx.x; b.l[1] += blockIdx.x & 0xff;
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  if (MODE == 0) {            // ten products per iteration, all different code copies (the loop tools/ubench_fp.hip calls "mul x10")
    F c = b, d = a;
    for (int i = 0; i < iters; i++) {
      a = F::mul(a, b); c = F::mul(c, a); d = F::mul(d, c); b = F::mul(b, d); a = F::mul(a, c);
      c = F::mul(c, d); d = F::mul(d, b); b = F::mul(b, a); a = F::mul(a, d); c = F::mul(c, b);
    }
    a = F::add(a, F::add(c, d));
  } else {                    // squarings
    for (int i = 0; i < iters; i++) {
      a = F::sqr(a); a = F::sqr(a); a = F::sqr(a); a = F::sqr(a); a = F::sqr(a);
      a = F::sqr(a); a = F::sqr(a); a = F::sqr(a); a = F::sqr(a); a = F::sqr(a);
    }
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  uint32_t r = 0;
  for (int i = 0; i < F::L; i++) r ^= a.l[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) {
    const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    ticks[2 * wv] = c1 - c0;
    ticks[2 * wv + 1] = r1 - r0;
  }
}

template <class F, int MODE>
static int ubench_one(uint32_t* d_out, uint64_t* d_ticks, std::vector<uint64_t>& h_ticks, int blocks, int iters, float* gops, float* ms_out, float* mhz, hipStream_t st,
                      hipEvent_t e0, hipEvent_t e1) {
  hipLaunchKernelGGL((k_ubench_fp<F, MODE>), dim3(blocks), dim3(256), 0, st, d_out, d_ticks, iters / 8);      // warm-up: code fetch, clocks ramping up
  if (hipEventRecord(e0, st) != hipSuccess) return 1;
  hipLaunchKernelGGL((k_ubench_fp<F, MODE>), dim3(blocks), dim3(256), 0, st, d_out, d_ticks, iters);
  if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) return 1;
  float ms = 0;
  if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 1;
  *gops = (float)((double)blocks * 256.0 * iters * 10.0 / ((double)ms * 1e6));
  *ms_out = ms;
  if (mhz) {
    if (hipMemcpy(h_ticks.data(), d_ticks, h_ticks.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    double sc = 0, sr = 0;
    for (size_t w = 0; w < h_ticks.size() / 2; w++) { sc += (double)h_ticks[2 * w]; sr += (double)h_ticks[2 * w + 1]; }
    *mhz = sr > 0 ? (float)(sc / sr * 100.0) : 0.f;
  }
  return 0;
}

// out[0..3] = G products/s, chip-wide: Fq-377 mul, Fq-377 sqr, Fq-761 mul, Fq-761 sqr;  out[4] = the shader clock during the Fq-377 mul loop in
// MHz as s_memtime / s_memrealtime read it (100.0 if s_memtime does not count shader clocks on this part);  out[5..8] = the four loops' kernel ms
int ubench_fp_run(float out[9]) {
  if (int rc = api_enter()) return rc;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, api_device()) != hipSuccess) return 100;
  const int blocks = p.multiProcessorCount * 8;       // 8 workgroups of 4 waves per CU: eight waves per SIMD requested (the 14-limb loops fit; the 28-limb ones get what fits)
  uint32_t* d_out = nullptr; uint64_t* d_ticks = nullptr;
  hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 1;
  std::vector<uint64_t> h_ticks((size_t)blocks * 4 * 2);
  do {
    if (hipMalloc(&d_out, (size_t)blocks * 256 * 4) != hipSuccess) break;
    if (hipMalloc(&d_ticks, h_ticks.size() * 8) != hipSuccess) break;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
    if (ubench_one<Fp<P377>, 0>(d_out, d_ticks, h_ticks, blocks, 400, &out[0], &out[5], &out[4], st, e0, e1)) break;
    if (ubench_one<Fp<P377>, 1>(d_out, d_ticks, h_ticks, blocks, 400, &out[1], &out[6], nullptr, st, e0, e1)) break;
    if (ubench_one<Fp<P761>, 0>(d_out, d_ticks, h_ticks, blocks, 100, &out[2], &out[7], nullptr, st, e0, e1)) break;
    if (ubench_one<Fp<P761>, 1>(d_out, d_ticks, h_ticks, blocks, 100, &out[3], &out[8], nullptr, st, e0, e1)) break;
    rc = 0;
  } while (0);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  if (d_out) (void)hipFree(d_out);
  if (d_ticks) (void)hipFree(d_ticks);
  return rc;
}
}  // namespace celo
