// Fq Montgomery-product throughput on gfx950: dependent chains of Fq::mul per lane, 1..4 waves per SIMD,
// one or two independent chains per lane.  Prices the generated asm (tools/gen_fp_asm.py) in G products/s.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aleo_amd/csrc -o fq_mul_bench tools/ubench/fq_mul_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fp.h"
using namespace aleo_mi355x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <class F, int CHAINS>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* in, uint32_t* out, int iters) {
  F x[CHAINS], b;
  size_t t = blockIdx.x * 256 + threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < F::N; ++i) x[c].v[i] = in[(t * 4 + c) % 4096 * 12 + i] >> 4;
  for (int i = 0; i < F::N; ++i) b.v[i] = in[((t * 4 + 3) % 4096) * 12 + i] >> 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = F::mul(x[c], b);
  }
  uint32_t acc = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < F::N; ++i) acc ^= x[c].v[i];
  out[t] = acc;
}

template <class F, int CHAINS> void run(const char* name, uint32_t* d_in, uint32_t* d_out, int cus) {
  const int iters = 2000;
  for (int wps : {1, 2, 3, 4, 6, 8}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL((k_chain<F, CHAINS>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 10);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain<F, CHAINS>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double muls = (double)blocks * 256 * iters * CHAINS;
    printf("%-6s chains/lane %d waves/SIMD %d : %8.3f ms  %8.2f G products/s   (%.1f ns per product per wave)\n", name, CHAINS, wps, ms, muls / ms / 1e6,
           ms * 1e6 / (iters * CHAINS));
  }
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  std::vector<uint32_t> h(4096 * 12); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
  uint32_t *d_in, *d_out; CK(hipMalloc(&d_in, h.size() * 4)); CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
  CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  run<Fq, 1>("Fq", d_in, d_out, cus); run<Fq, 2>("Fq", d_in, d_out, cus);
  run<Fr, 1>("Fr", d_in, d_out, cus); run<Fr, 2>("Fr", d_in, d_out, cus);
  return 0;
}
