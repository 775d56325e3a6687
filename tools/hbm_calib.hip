// hbm_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of the seeding and
// candidate kernels, on known byte counts (MI355X_MICROARCH.md, HBM section: only wide coalesced streams are calibrated;
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
//   k_stream_read16   every lane reads 16 B, consecutive lanes consecutive addresses (the guide's reference pattern)
//   k_probe8          every lane reads 8 B at an independent random 8-byte-aligned offset of a 1 GiB table (k_seed's idx probes)
//   k_list4           groups of 8 lanes read 8 consecutive 4-byte words at a random offset (k_cands' position lists)
//   k_stream_write16  every lane writes 16 B, consecutive (reference pattern for WRITE_SIZE)
//   k_record48        every lane writes one 48-byte record, consecutive lanes consecutive records (the ranked-candidate pool)
// Each kernel moves BYTES bytes of payload exactly once; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... and --pmc WRITE_SIZE --kernel-trace ...  (separate passes)
// and divide the counter (KB) by the payload: tools/refresh_profiles.py does that and writes profiles/rNN_hbm_calibration.txt.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

static const size_t TABLE = 1ull << 30;          // 1 GiB: four times the Infinity Cache
static const size_t PAYLOAD = 1ull << 28;        // 256 MiB moved per kernel

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void k_stream_read16(const uint4 *src, size_t n16, uint4 *sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678u && acc.y == 1u) sink[0] = acc;
}
__global__ void k_probe8(const uint2 *tab, size_t ntab8, size_t nprobe, uint2 *sink) {
  uint2 acc = make_uint2(0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nprobe; i += (size_t)gridDim.x * blockDim.x) { const uint2 v = tab[mix(i) % ntab8]; acc.x ^= v.x; acc.y ^= v.y; }
  if (acc.x == 0x12345678u && acc.y == 1u) sink[0] = acc;
}
__global__ void k_list4(const uint32_t *tab, size_t ntab4, size_t nword, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nword; i += (size_t)gridDim.x * blockDim.x) acc ^= tab[(mix(i >> 3) % (ntab4 - 8)) + (i & 7)];
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_stream_write16(uint4 *dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
struct Rec48 { uint64_t a, b; uint32_t c[8]; };
__global__ void k_record48(Rec48 *dst, size_t nrec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
    Rec48 r; r.a = i; r.b = ~i; for (int k = 0; k < 8; k++) r.c[k] = (uint32_t)(i + k);
    dst[i] = r;
  }
}

int main() {
  void *tab = nullptr, *out = nullptr;
  if (hipMalloc(&tab, TABLE) != hipSuccess || hipMalloc(&out, PAYLOAD + 4096) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  (void)hipMemset(tab, 1, TABLE);
  (void)hipMemset(out, 0, PAYLOAD + 4096);
  (void)hipDeviceSynchronize();
  const dim3 grid(256 * 16), block(256);
  hipLaunchKernelGGL(k_stream_read16, grid, block, 0, 0, (const uint4 *)tab, PAYLOAD / 16, (uint4 *)out);
  hipLaunchKernelGGL(k_probe8, grid, block, 0, 0, (const uint2 *)tab, TABLE / 8, PAYLOAD / 8, (uint2 *)out);
  hipLaunchKernelGGL(k_list4, grid, block, 0, 0, (const uint32_t *)tab, TABLE / 4, PAYLOAD / 4, (uint32_t *)out);
  hipLaunchKernelGGL(k_stream_write16, grid, block, 0, 0, (uint4 *)out, PAYLOAD / 16);
  hipLaunchKernelGGL(k_record48, grid, block, 0, 0, (Rec48 *)out, PAYLOAD / 48);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
  printf("payload per kernel: %zu bytes (k_record48: %zu)\n", PAYLOAD, (PAYLOAD / 48) * 48);
  (void)hipFree(tab); (void)hipFree(out);
  return 0;
}
