// Throughput + known answers of the 9 x 29-bit Fr Montgomery product (tools/gen_fr29_asm.py, R = 2^261) next to the 8 x 32-bit one (fp.h Fr::mul, R = 2^256).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aleo_amd/csrc -o tools/ubench/fr29_mul_bench tools/ubench/fr29_mul_bench.hip
// The KAT lines are checked against big integers by tools/ubench/fr29_kat_check.py (reads this program's output).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fp.h"
#include "fr29_mont_gen.h"
using namespace aleo_mi355x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int CHAINS>
__global__ void __launch_bounds__(256) k_chain29(const uint32_t* in, uint32_t* out, int iters) {
  uint32_t x[CHAINS][9], b[9];
  size_t t = blockIdx.x * 256 + threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) { for (int i = 0; i < 9; ++i) x[c][i] = in[((t * 4 + c) % 4096) * 9 + i] & 0x1fffffffu; x[c][8] &= 0xfffff; }
  for (int i = 0; i < 9; ++i) b[i] = in[((t * 4 + 3) % 4096) * 9 + i] & 0x1fffffffu;
  b[8] &= 0xfffff;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) mont29_mul_inplace(x[c], b);
  }
  uint32_t acc = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 9; ++i) acc ^= x[c][i];
  out[t] = acc;
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_chain32(const uint32_t* in, uint32_t* out, int iters) {
  Fr x[CHAINS], b;
  size_t t = blockIdx.x * 256 + threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) { for (int i = 0; i < 8; ++i) x[c].v[i] = in[((t * 4 + c) % 4096) * 9 + i]; x[c].v[7] &= 0x0fffffffu; }
  for (int i = 0; i < 8; ++i) b.v[i] = in[((t * 4 + 3) % 4096) * 9 + i];
  b.v[7] &= 0x0fffffffu;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = Fr::mul(x[c], b);
  }
  uint32_t acc = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) acc ^= x[c].v[i];
  out[t] = acc;
}
__global__ void k_kat(const uint32_t* a, const uint32_t* b, uint32_t* r, int n) {
  for (int k = 0; k < n; ++k) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; ++i) { x[i] = a[9 * k + i]; y[i] = b[9 * k + i]; }
    mont29_mul_inplace(x, y);
    for (int i = 0; i < 9; ++i) r[9 * k + i] = x[i];
  }
}
template <typename K> void run(const char* name, K kern, int chains, uint32_t* d_in, uint32_t* d_out, int cus) {
  const int iters = 4000;
  for (int wps : {1, 2, 4}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_in, d_out, 10);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double muls = (double)blocks * 256 * iters * chains;
    printf("%-8s chains/lane %d waves/SIMD %d : %8.3f ms  %8.2f G products/s\n", name, chains, wps, ms, muls / ms / 1e6);
  }
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  std::vector<uint32_t> h(4096 * 9); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
  uint32_t *d_in, *d_out; CK(hipMalloc(&d_in, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
  CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  // known answers: pair 0 normalised limbs; pair 1 a loose multiplicand (limbs up to 2^31.4, as a difference u + C - x leaves them); pair 2 largest normalised operands
  const int NK = 3; std::vector<uint32_t> a(9 * NK), b(9 * NK), r(9 * NK);
  for (int i = 0; i < 9; ++i) { a[i] = h[i] & 0x1fffffffu; b[i] = h[9 + i] & 0x1fffffffu; } a[8] &= 0xfffff; b[8] &= 0xfffff;
  for (int i = 0; i < 9; ++i) { a[9 + i] = (h[18 + i] & 0x7fffffffu) + 0x50000000u; b[9 + i] = h[27 + i] & 0x1fffffffu; }
  for (int i = 0; i < 9; ++i) { a[18 + i] = 0x1fffffffu; b[18 + i] = 0x1fffffffu; }
  uint32_t *da, *db, *dr; CK(hipMalloc(&da, 36 * NK)); CK(hipMalloc(&db, 36 * NK)); CK(hipMalloc(&dr, 36 * NK));
  CK(hipMemcpy(da, a.data(), 36 * NK, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 36 * NK, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_kat, dim3(1), dim3(1), 0, 0, da, db, dr, NK); CK(hipMemcpy(r.data(), dr, 36 * NK, hipMemcpyDeviceToHost));
  for (int k = 0; k < NK; ++k) {
    printf("KAT a"); for (int i = 0; i < 9; ++i) printf(" %x", a[9 * k + i]); printf("\nKAT b"); for (int i = 0; i < 9; ++i) printf(" %x", b[9 * k + i]);
    printf("\nKAT r"); for (int i = 0; i < 9; ++i) printf(" %x", r[9 * k + i]); printf("\n");
  }
  run("mul29", k_chain29<1>, 1, d_in, d_out, cus); run("mul29", k_chain29<2>, 2, d_in, d_out, cus);
  run("mul32", k_chain32<1>, 1, d_in, d_out, cus); run("mul32", k_chain32<2>, 2, d_in, d_out, cus);
  return 0;
}
