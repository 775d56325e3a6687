// valu_rate.hip -- issue rate of the VALU instructions the Smith-Waterman kernels are built from
// (gfx950).  Each lane runs 8 independent dependency chains of one instruction; the grid fills every
// SIMD with 8 waves.  Prints wave-instructions per clock per CU (4 SIMDs): 2.0 = one per 2 cycles per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define AS_US2(v) __builtin_bit_cast(us2, (uint32_t)(v))
#define AS_U32(v) __builtin_bit_cast(uint32_t, (v))

template <int OP>
__device__ inline uint32_t op(uint32_t a, uint32_t b, uint32_t c) {   // inline asm: the optimiser must not fold the chains
  if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 1) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (OP == 2) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 3) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 4) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a) : "v"(b));
  if (OP == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (OP == 6) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));
  if (OP == 7) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 8) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 9) asm volatile("v_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 10) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 11) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 12) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 13) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (OP == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 15) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 16) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a));
  if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
  if (OP == 18) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 19) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 20) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 22) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
  if (OP == 23) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));
  if (OP == 24) asm volatile("v_max3_f16 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (OP == 25) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (OP == 26) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 27) asm volatile("v_max_f16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 28) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a) : "v"(b));
  if (OP == 29) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (OP == 30) asm volatile("v_sub_u16 %0, %0, %1 clamp" : "+v"(a) : "v"(b));
  if (OP == 31) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (OP == 32) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
  if (OP == 33) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a) : "v"(b));
  return a;
}

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t b, uint32_t c) {
  uint32_t x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 8 + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = op<OP>(x[i], b, c);
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, uint32_t *d, int ncu, double mhz) {
  const int iters = 4096, blocks = ncu * 8;     // 8 blocks x 4 waves = 32 waves per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 16, 3u, 0x0c020100u);
  (void)hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, 0x0c020100u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 32;     // wave-instructions
  const double per_clk_cu = winstr / (ms * 1e-3) / (mhz * 1e6) / ncu;
  printf("%-22s %8.3f ms  %6.3f wave-instr/clk/CU  (%.2f cyc per instr per SIMD)\n", name, ms, per_clk_cu, 4.0 / per_clk_cu);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;
  printf("%s: %d CUs, %.0f MHz\n", p.name, ncu, mhz);
  uint32_t *d;
  hipMalloc(&d, (size_t)ncu * 8 * 256 * 4);
  run<0>("v_add_u32", d, ncu, mhz);
  run<7>("v_max_i32", d, ncu, mhz);
  run<1>("v_max3_i32", d, ncu, mhz);
  run<2>("v_pk_add_u16", d, ncu, mhz);
  run<3>("v_pk_max_u16", d, ncu, mhz);
  run<4>("v_pk_sub_u16 clamp", d, ncu, mhz);
  run<5>("v_perm_b32", d, ncu, mhz);
  run<6>("v_add_u32 dpp row_shr", d, ncu, mhz);
  run<8>("v_pk_max_i16", d, ncu, mhz);
  run<9>("v_max_u16", d, ncu, mhz);
  run<10>("v_pk_add_i16", d, ncu, mhz);
  run<11>("v_pk_add_f16", d, ncu, mhz);
  run<12>("v_pk_max_f16", d, ncu, mhz);
  run<13>("v_max3_f32", d, ncu, mhz);
  run<14>("v_max_f32", d, ncu, mhz);
  run<15>("v_add_f32", d, ncu, mhz);
  run<16>("v_cvt_f32_ubyte0", d, ncu, mhz);
  run<17>("v_cndmask_b32", d, ncu, mhz);
  run<18>("v_max_u32", d, ncu, mhz);
  run<19>("v_max_i16", d, ncu, mhz);
  run<20>("v_sub_u16", d, ncu, mhz);
  run<21>("v_and_b32", d, ncu, mhz);
  run<22>("v_mov_b32", d, ncu, mhz);
  run<23>("v_mov_b32 dpp", d, ncu, mhz);
  run<24>("v_max3_f16", d, ncu, mhz);
  run<25>("v_pk_fma_f16", d, ncu, mhz);
  run<26>("v_sub_u32", d, ncu, mhz);
  run<27>("v_max_f16", d, ncu, mhz);
  run<28>("v_pk_min_f16", d, ncu, mhz);
  run<29>("v_add3_u32", d, ncu, mhz);
  run<30>("v_sub_u16 clamp", d, ncu, mhz);
  run<31>("v_med3_i32", d, ncu, mhz);
  run<32>("v_lshl_add_u32", d, ncu, mhz);
  run<33>("v_pk_mul_f16", d, ncu, mhz);
  hipFree(d);
  return 0;
}
