// Throughput + a known-answer print of the 14 x 28-bit Montgomery product (tools/gen_fp28_asm.py) next to the 12 x 32-bit one.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aleo_amd/csrc -I tools/ubench -o fq28_mul_bench tools/ubench/fq28_mul_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fp.h"
#include "fp28_mont_gen.h"
using namespace aleo_mi355x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int CHAINS, bool SQR>
__global__ void __launch_bounds__(256) k_chain28(const uint32_t* in, uint32_t* out, int iters) {
  uint32_t x[CHAINS][14], b[14];
  size_t t = blockIdx.x * 256 + threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 14; ++i) x[c][i] = in[((t * 4 + c) % 4096) * 14 + i] & 0x0fffffffu;
  for (int i = 0; i < 14; ++i) b[i] = in[((t * 4 + 3) % 4096) * 14 + i] & 0x0fffffffu;
  for (int c = 0; c < CHAINS; ++c) x[c][13] &= 0xfff; b[13] &= 0xfff;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { if (SQR) mont28_sqr_inplace(x[c]); else mont28_mul_inplace(x[c], b); }
  }
  uint32_t acc = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 14; ++i) acc ^= x[c][i];
  out[t] = acc;
}
__global__ void k_kat(const uint32_t* a, const uint32_t* b, uint32_t* r) {
  uint32_t x[14], y[14];
  for (int i = 0; i < 14; ++i) { x[i] = a[i]; y[i] = b[i]; }
  mont28_mul_inplace(x, y);
  for (int i = 0; i < 14; ++i) r[i] = x[i];
  for (int i = 0; i < 14; ++i) x[i] = a[i];
  mont28_sqr_inplace(x);
  for (int i = 0; i < 14; ++i) r[14 + i] = x[i];
}

template <int CHAINS, bool SQR> void run(const char* name, uint32_t* d_in, uint32_t* d_out, int cus) {
  const int iters = 2000;
  for (int wps : {1, 2, 3, 4}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL((k_chain28<CHAINS, SQR>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 10);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain28<CHAINS, SQR>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double muls = (double)blocks * 256 * iters * CHAINS;
    printf("%-8s chains/lane %d waves/SIMD %d : %8.3f ms  %8.2f G products/s\n", name, CHAINS, wps, ms, muls / ms / 1e6);
  }
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  std::vector<uint32_t> h(4096 * 14); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
  uint32_t *d_in, *d_out; CK(hipMalloc(&d_in, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
  CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  // known answer: a, b = first two rows, limbs masked to 28 bits (top limb 12 bits): printed for an offline big-integer check
  std::vector<uint32_t> a(14), b(14), r(28);
  for (int i = 0; i < 14; ++i) { a[i] = h[i] & 0x0fffffffu; b[i] = h[14 + i] & 0x0fffffffu; } a[13] &= 0xfff; b[13] &= 0xfff;
  uint32_t *da, *db, *dr; CK(hipMalloc(&da, 56)); CK(hipMalloc(&db, 56)); CK(hipMalloc(&dr, 112));
  CK(hipMemcpy(da, a.data(), 56, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 56, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_kat, dim3(1), dim3(1), 0, 0, da, db, dr); CK(hipMemcpy(r.data(), dr, 112, hipMemcpyDeviceToHost));
  printf("KAT a"); for (auto v : a) printf(" %x", v); printf("\nKAT b"); for (auto v : b) printf(" %x", v);
  printf("\nKAT mul"); for (int i = 0; i < 14; ++i) printf(" %x", r[i]); printf("\nKAT sqr"); for (int i = 0; i < 14; ++i) printf(" %x", r[14 + i]); printf("\n");
  run<1, false>("mul28", d_in, d_out, cus); run<2, false>("mul28", d_in, d_out, cus);
  run<1, true>("sqr28", d_in, d_out, cus); run<2, true>("sqr28", d_in, d_out, cus);
  return 0;
}
