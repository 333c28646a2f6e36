// Microbenchmark: issue rate of the VALU instructions a big-integer Montgomery multiply can be built from
// on gfx950.  Prints wave-instructions per cycle per SIMD and the implied cycles per wave64 instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 65536;
constexpr int UNROLL = 8;  // independent chains

template <int OP> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a[UNROLL], b[UNROLL];
  uint64_t c[UNROLL];
  double d[UNROLL], e[UNROLL], f[UNROLL];
  for (int i = 0; i < UNROLL; i++) {
    a[i] = seed * (threadIdx.x + 1) + i * 7919u;
    b[i] = seed ^ (threadIdx.x * 2654435761u + i);
    c[i] = ((uint64_t)a[i] << 32) | b[i];
    d[i] = 1.0 + 1e-9 * (threadIdx.x + i); e[i] = 1.0 - 1e-9 * i; f[i] = 1e-3 * i;
  }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if constexpr (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
      if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 4) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(e[i]), "v"(f[i]));
      if constexpr (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(b[i]), "v"(a[i]) : "vcc");
      if constexpr (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[i]) : "v"(c[(i + 1) % UNROLL]));
      if constexpr (OP == 9) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % UNROLL]));
      if constexpr (OP == 10) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "s10", "s11");
      if constexpr (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 12) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
      if constexpr (OP == 13) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 14) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
      if constexpr (OP == 15) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % UNROLL]));
      if constexpr (OP == 16) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a[i]) : "v"(b[i]));
      if constexpr (OP == 20) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[0]) : "v"(a[i]), "v"(b[i]) : "vcc");
      if constexpr (OP == 21) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[i&1]) : "v"(a[i]), "v"(b[i]) : "vcc");
      if constexpr (OP == 22) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(b[i]));
      if constexpr (OP == 17) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % UNROLL]));
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < UNROLL; i++) r ^= a[i] ^ b[i] ^ (uint32_t)c[i] ^ (uint32_t)(c[i] >> 32) ^ (uint32_t)__double_as_longlong(d[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP> int run(const char* name, int insts_per_iter, uint32_t* dout, int cus, double mhz) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wpe : {1, 4, 8}) {  // waves per SIMD (blocks of 256 threads = 1 wave per SIMD)
    int blocks = cus * wpe;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dout, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, dout, 12345u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double wave_insts_per_simd = (double)ITERS * UNROLL * insts_per_iter * wpe;
    double cycles = ms * 1e-3 * mhz * 1e6;
    printf("%-28s waves/SIMD=%d  %8.3f ms  cycles/wave-inst=%6.2f  (lane-ops/clk/CU=%6.1f)\n", name, wpe, ms,
           cycles / wave_insts_per_simd, wave_insts_per_simd * 64 * 4 / cycles);
  }
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  double mhz = p.clockRate / 1000.0;
  printf("device %s  CUs=%d clock=%.0f MHz (cycle figures assume this clock; DVFS may run lower)\n", p.gcnArchName, cus, mhz);
  uint32_t* dout;
  CK(hipMalloc(&dout, (size_t)cus * 8 * 256 * 4));
  run<7>("v_add_u32", 1, dout, cus, mhz);
  run<0>("v_mad_u64_u32 (vcc)", 1, dout, cus, mhz);
  run<10>("v_mad_u64_u32 (sgpr carry)", 1, dout, cus, mhz);
  run<14>("v_mad_i64_i32", 1, dout, cus, mhz);
  run<20>("mad_u64 dependent chain x1", 1, dout, cus, mhz);
  run<21>("mad_u64 dependent chain x2", 1, dout, cus, mhz);
  run<22>("v_add_u32 dependent chain", 1, dout, cus, mhz);
  run<1>("v_mul_lo_u32", 1, dout, cus, mhz);
  run<2>("v_mul_hi_u32", 1, dout, cus, mhz);
  run<3>("v_mad_u32_u24", 1, dout, cus, mhz);
  run<11>("v_mul_u32_u24", 1, dout, cus, mhz);
  run<4>("v_mul_hi_u32_u24", 1, dout, cus, mhz);
  run<5>("v_fma_f64", 1, dout, cus, mhz);
  run<12>("v_mul_f64", 1, dout, cus, mhz);
  run<6>("v_add_co+v_addc_co pair", 2, dout, cus, mhz);
  run<8>("v_lshl_add_u64", 1, dout, cus, mhz);
  run<9>("v_dot4_u32_u8", 1, dout, cus, mhz);
  run<17>("v_dot2_u32_u16", 1, dout, cus, mhz);
  run<13>("v_pk_mul_lo_u16", 1, dout, cus, mhz);
  run<15>("v_add3_u32", 1, dout, cus, mhz);
  run<16>("v_alignbit_b32", 1, dout, cus, mhz);
  return 0;
}
