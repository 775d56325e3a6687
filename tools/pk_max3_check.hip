// pk_max3_check.hip -- v_pk_maximum3_f16 on u16 bit patterns against the integer maximum of three (10^6 random triples incl. the
// half-float denormal range); build: hipcc --offload-arch=gfx950 -O2 -o pk_max3_check pk_max3_check.hip.  Cited in DESIGN.md 4.1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ inline uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__global__ void k(const uint32_t *in, uint32_t *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pk_max3(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
}
int main() {
  const int n = 1 << 20;
  uint32_t *h = (uint32_t *)malloc(12 * n), *o = (uint32_t *)malloc(4 * n), *di, *dout;
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < 3 * n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    uint32_t lo = (uint32_t)(s & 0xffff) % 31000u, hi = (uint32_t)((s >> 20) & 0xffff) % 31000u;
    if (i % 7 == 0) { lo %= 700; hi %= 1100; }         // small values: f16 denormal range
    h[i] = lo | (hi << 16);
  }
  hipMalloc(&di, 12 * n); hipMalloc(&dout, 4 * n);
  hipMemcpy(di, h, 12 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, di, dout, n);
  hipMemcpy(o, dout, 4 * n, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n; i++) {
    uint32_t a = h[3 * i], b = h[3 * i + 1], c = h[3 * i + 2];
    uint32_t lo = a & 0xffff; if ((b & 0xffff) > lo) lo = b & 0xffff; if ((c & 0xffff) > lo) lo = c & 0xffff;
    uint32_t hi = a >> 16; if ((b >> 16) > hi) hi = b >> 16; if ((c >> 16) > hi) hi = c >> 16;
    if (o[i] != (lo | (hi << 16))) { if (bad < 5) printf("mismatch %08x %08x %08x -> %08x want %08x\n", a, b, c, o[i], lo | (hi << 16)); bad++; }
  }
  printf("bad %ld of %d\n", bad, n);
  return bad != 0;
}
