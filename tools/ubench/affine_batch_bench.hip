// EXPERIMENT: what an affine addition costs when the inversion is shared by a per-lane batch.  Every lane owns B independent
// additions (x1, y1) + (x2, y2): pass 1 multiplies the B differences x2 - x1 into prefix products (spilled to HBM), one Fermat chain inverts the
// last of them, pass 2 walks back (two products per element), forms the chord (one product, one square, one product) and stores (x3, y3) —
// six products and 1/B of a 567-product inversion per addition, ≈ 560 B of HBM traffic.  Beside it, the XYZZ mixed addition of the accumulation kernel
// over the same number of operands (ten products, 112 B).  Operands are pseudo-random field elements (the formulas do not care about the curve);
// the back-substituted inverses are checked (inv_i * (x2 - x1) == 1) on a sample.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I aleo_amd/csrc -o tools/ubench/affine_batch_bench tools/ubench/affine_batch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ec.h"
#include "fp28.h"
using namespace aleo_mi355x;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __constant__ uint32_t Q_MINUS_2[12] = {0xffffffffu, 0x8508bfffu, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu,
                                                  0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u};
__device__ __noinline__ void f28_mul_ni(F28* r, const F28* a, const F28* b) { *r = f28_mul(*a, *b); }
__device__ __noinline__ void f28_inverse_ni(F28* io) {      // a^(q-2), Montgomery form in and out (< 2q, exact digits)
  F28 a = *io, acc = f28_const(ONE28);
  for (int bit = 376; bit >= 0; --bit) {
    f28_mul_ni(&acc, &acc, &acc);
    if ((Q_MINUS_2[bit >> 5] >> (bit & 31)) & 1u) f28_mul_ni(&acc, &acc, &a);
  }
  *io = acc;
}
__device__ __forceinline__ bool same_one(const F28& a) { Fq x = Fq::reduce(f28_to_fq(a)), o = Fq::reduce(Fq::one()); uint32_t d = 0; for (int i = 0; i < 12; ++i) d |= x.v[i] ^ o.v[i]; return d == 0; }

// element i of lane t of array k: ((k * B + i) * lanes + t) * 56 bytes  (coalesced across lanes)
__global__ void k_fill(char* buf, size_t elems) {
  size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; if (e >= elems) return;
  uint64_t s = 0x9e3779b97f4a7c15ull * (e + 1); F28 v;
  for (int i = 0; i < 14; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; v.v[i] = (uint32_t)(s >> 36); }      // 28-bit digits
  v.v[13] &= 0x7ffu;                                       // < 2^375 < q
  store_f28(buf + e * 56, v);
}

template <int B>
__global__ void __launch_bounds__(256) k_affine_batch(const char* __restrict__ in, char* __restrict__ spill, char* __restrict__ out, uint32_t lanes, uint32_t* bad) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x; if (t >= lanes) return;
  auto at = [&](const char* base, int k, int i) { return base + (((size_t)k * B + i) * lanes + t) * 56; };
  F28 p = f28_const(ONE28);
  for (int i = 0; i < B; ++i) {
    const F28 d = f28_sub<4, 1>(load_f28(at(in, 2, i)), load_f28(at(in, 0, i)));      // x2 - x1: class L3
    p = f28_mul(p, d);
    store_f28((char*)at(spill, 0, i), p);
  }
  f28_inverse_ni(&p);
  F28 inv = p;
  for (int i = B - 1; i >= 0; --i) {
    const F28 x1 = load_f28(at(in, 0, i)), x2 = load_f28(at(in, 2, i));
    const F28 d = f28_sub<4, 1>(x2, x1);
    const F28 prev = i ? load_f28(at(spill, 0, i - 1)) : f28_const(ONE28);
    const F28 inv_i = f28_mul(inv, prev);
    inv = f28_mul(inv, d);
    if (i == B / 2 && (t & 1023u) == 0 && !same_one(f28_mul(inv_i, d))) atomicAdd(bad, 1u);
    const F28 y1 = load_f28(at(in, 1, i));
    const F28 lam = f28_mul(f28_sub<4, 1>(load_f28(at(in, 3, i)), y1), inv_i);
    const F28 x3 = f28_normalise(f28_sub<6, 2>(f28_sub<4, 1>(f28_sqr(lam), x1), x2));
    const F28 y3 = f28_sub<4, 1>(f28_mul(lam, f28_sub<16, 1>(x1, x3)), y1);
    store_f28((char*)at(out, 0, i), x3); store_f28((char*)at(out, 1, i), y3);
  }
}

template <int B>
__global__ void __launch_bounds__(256) k_xyzz_chain(const char* __restrict__ in, char* __restrict__ out, uint32_t lanes) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x; if (t >= lanes) return;
  auto at = [&](const char* base, int k, int i) { return base + (((size_t)k * B + i) * lanes + t) * 56; };
  XYZZ28 acc; acc.X = load_f28(at(in, 0, 0)); acc.Y = load_f28(at(in, 1, 0)); acc.ZZ = f28_const(ONE28); acc.ZZZ = f28_const(ONE28);
  for (int i = 0; i < B; ++i) (void)xyzz28_madd_fast(acc, load_f28(at(in, 2, i)), load_f28(at(in, 3, i)));
  store_f28((char*)at(out, 0, 0), acc.X); store_f28((char*)at(out, 1, 0), acc.Y); store_f28((char*)at(out, 0, 1), acc.ZZ); store_f28((char*)at(out, 1, 1), acc.ZZZ);
}

template <int B> void run(uint32_t lanes) {
  const size_t n = (size_t)B * lanes;
  char *in, *spill, *out; uint32_t* bad;
  CK(hipMalloc(&in, 4 * n * 56)); CK(hipMalloc(&spill, n * 56)); CK(hipMalloc(&out, 2 * n * 56)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, 0, in, 4 * n); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms_a = 0, ms_x = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_affine_batch<B>, dim3((lanes + 255) / 256), dim3(256), 0, 0, in, spill, out, lanes, bad); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_a, e0, e1));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_xyzz_chain<B>, dim3((lanes + 255) / 256), dim3(256), 0, 0, in, out, lanes); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_x, e0, e1));
  }
  uint32_t h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
  printf("lanes 2^%d  B %4d : affine batch %8.3f ms = %6.2f G additions/s (%.2f ns per addition) | XYZZ mixed %8.3f ms = %6.2f G additions/s | ratio %.2f | inverse check failures %u\n",
         31 - __builtin_clz(lanes), B, ms_a, n / ms_a / 1e6, ms_a * 1e6 / n, ms_x, n / ms_x / 1e6, ms_x / ms_a, h);
  CK(hipFree(in)); CK(hipFree(spill)); CK(hipFree(out)); CK(hipFree(bad));
}

int main() {
  for (uint32_t lg : {17u, 18u}) { run<32>(1u << lg); run<64>(1u << lg); run<128>(1u << lg); run<256>(1u << lg); }
  return 0;
}
