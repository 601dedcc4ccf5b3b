// Dependent-chain issue rate of v_mad_u64_u32 on gfx950: does acc = a*b + acc back-to-back stall a single wave?
// Decides whether a carry-free (28-bit limb) column accumulation can run one accumulator per column.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 4096;
template <int CH>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint64_t acc[CH]; uint32_t a[8], b[8];
  for (int i = 0; i < CH; ++i) acc[i] = seed + i;
  for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 3 + i) | 1; b[i] = a[i] * 2654435761u; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a[(r + c) & 7]), "v"(b[(r * 3 + c) & 7]) : "vcc");
  }
  uint32_t x = 0; for (int i = 0; i < CH; ++i) x ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32);
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int CH> void run(uint32_t* d, int cus) {
  for (int wps : {1, 2, 4}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, d, 7u); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, d, 7u + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double per_simd_ns = ms * 1e6 / ((double)ITERS * 16 * wps);
    printf("chains %d waves/SIMD %d : %.3f ms, %.2f ns per mad per SIMD (%.1f cycles @2.4GHz), %.1f T lane-mads/s\n", CH, wps, ms, per_simd_ns, per_simd_ns * 2.4,
           (double)blocks * 256 * ITERS * 16 / ms / 1e9);
  }
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); int cus = p.multiProcessorCount;
  uint32_t* d; CK(hipMalloc(&d, (size_t)cus * 8 * 256 * 4));
  run<1>(d, cus); run<2>(d, cus); run<4>(d, cus); run<8>(d, cus);
  return 0;
}
