// EXPERIMENT (round 3): an Fq Montgomery product on the FP64 pipe — 8 limbs of 52 bits held as doubles, every 52 x 52-bit
// partial product split into its high and low 52 bits by two v_fma_f64 in round-toward-zero mode (the 2^104 bias trick), columns accumulated
// as 64-bit integers of the results' bit patterns (biases removed once per column), word-by-word Montgomery reduction with R = 2^416.
// Prints a known answer (checked offline against a * b * 2^-416 mod q) and products/s next to the 14 x 28-bit v_mad_u64_u32 block
// (tools/ubench/fq28_mul_bench.hip: 81 G products/s).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o fq52_fma_bench tools/ubench/fq52_fma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// q = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001 in 52-bit limbs, least significant first
#define Q52_INIT {0x8c00000000001ull, 0x4430000000850ull, 0xa094800170b5dull, 0x138f1ef3622fbull, 0xb1a22d9f300f5ull, 0x3b05c06ca1493ull, 0xa4617c510eac6ull, 0x1ae3ull}
#define NINV52 0x8bfffffffffffull          // -q^-1 mod 2^52  (q = 1 mod 2^46: the low bits of the inverse are those of -q)

static constexpr unsigned long long MASK52 = (1ull << 52) - 1, LO_OFF = 0x433ull << 52, HI_OFF = 0x467ull << 52;

__device__ __forceinline__ unsigned long long dbits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ double bits_to_double52(unsigned long long low52) { return __longlong_as_double((long long)(low52 | LO_OFF)) - 4503599627370496.0; }

// r = a * b * 2^-416 mod q (r < 2q for a, b < 2q); limbs are integers < 2^52 held in doubles
__device__ __forceinline__ void mont52_mul(double* r, const double* a, const double* b, const double* q, double ninv) {
  const double C1 = 20282409603651670423947251286016.0;                 // 2^104
  const double C2 = 20282409603651674927546878656512.0;                 // 2^104 + 2^52
  unsigned long long col[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) col[j] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double hi = __builtin_fma(a[j], b[i], C1), lo = __builtin_fma(a[j], b[i], C2 - hi);
      col[j] += dbits(lo); col[j + 1] += dbits(hi);
    }
    const double x = bits_to_double52(col[0] & MASK52);                 // the biases have no low 52 bits
    const double mh = __builtin_fma(x, ninv, C1), m = __builtin_fma(x, ninv, C2 - mh) - 4503599627370496.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double hi = __builtin_fma(m, q[j], C1), lo = __builtin_fma(m, q[j], C2 - hi);
      col[j] += dbits(lo); col[j + 1] += dbits(hi);
    }
    // absolute column i is complete: 2(i+1) low halves and 2i high halves went into it (mod 2^64 arithmetic; the true value is < 2^58)
    const unsigned long long t = col[0] - (unsigned long long)(2 * (i + 1)) * LO_OFF - (unsigned long long)(2 * i) * HI_OFF;
    col[1] += t >> 52;
#pragma unroll
    for (int j = 0; j < 8; ++j) col[j] = col[j + 1];
    col[8] = 0;
  }
  unsigned long long carry = 0;
#pragma unroll
  for (int k = 8; k < 16; ++k) {                                          // absolute columns 8..15: 2(15-k) low halves, 2(16-k) high halves each
    const unsigned long long t = col[k - 8] - (unsigned long long)(2 * (15 - k)) * LO_OFF - (unsigned long long)(2 * (16 - k)) * HI_OFF + carry;
    r[k - 8] = bits_to_double52(t & MASK52); carry = t >> 52;
  }
}

template <int CHAINS>
__global__ void __launch_bounds__(256) k_chain52(const double* in, double* out, int iters) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");           // MODE.fp_round[3:2] (f64/f16) = round toward zero; as asm: the compiler's mode-register pass undoes the builtin
  const unsigned long long Q[8] = Q52_INIT;
  double q[8]; for (int i = 0; i < 8; ++i) q[i] = (double)Q[i];
  const double ninv = (double)NINV52;
  double x[CHAINS][8], b[8];
  const size_t t = blockIdx.x * 256 + threadIdx.x;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) x[c][i] = in[((t * 4 + c) % 4096) * 8 + i];
  for (int i = 0; i < 8; ++i) b[i] = in[((t * 4 + 3) % 4096) * 8 + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { double r[8]; mont52_mul(r, x[c], b, q, ninv); for (int i = 0; i < 8; ++i) x[c][i] = r[i]; }
  }
  double acc = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) acc += x[c][i];
  out[t] = acc;
}
__global__ void k_kat(const double* a, const double* b, double* r) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
  const unsigned long long Q[8] = Q52_INIT;
  double q[8]; for (int i = 0; i < 8; ++i) q[i] = (double)Q[i];
  double x[8], y[8], o[8];
  for (int i = 0; i < 8; ++i) { x[i] = a[i]; y[i] = b[i]; }
  mont52_mul(o, x, y, q, (double)NINV52);
  for (int i = 0; i < 8; ++i) r[i] = o[i];
  for (int it = 0; it < 100; ++it) { mont52_mul(o, x, y, q, (double)NINV52); for (int i = 0; i < 8; ++i) x[i] = o[i]; }      // a b^100 R^-100: a chain, as the benchmark runs it
  for (int i = 0; i < 8; ++i) r[8 + i] = x[i];
}

template <int CHAINS> void run(double* d_in, double* d_out, int cus) {
  const int iters = 1000;
  for (int wps : {1, 2, 3, 4}) {
    int blocks = cus * wps;
    hipLaunchKernelGGL((k_chain52<CHAINS>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, 10);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chain52<CHAINS>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mul52fma chains/lane %d waves/SIMD %d : %8.3f ms  %8.2f G products/s\n", CHAINS, wps, ms, (double)blocks * 256 * iters * CHAINS / ms / 1e6);
  }
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<double> h(4096 * 8); uint64_t s = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < h.size(); ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; uint64_t v = (s >> 12) & ((1ull << 52) - 1); if (i % 8 == 7) v &= (1ull << 12) - 1; h[i] = (double)v; }   // values < 2^376 < q
  double *d_in, *d_out; CK(hipMalloc(&d_in, h.size() * 8)); CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
  CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  double r[16], *dr; CK(hipMalloc(&dr, 128));
  hipLaunchKernelGGL(k_kat, dim3(1), dim3(1), 0, 0, d_in, d_in + 8, dr); CK(hipMemcpy(r, dr, 128, hipMemcpyDeviceToHost));
  printf("KAT a"); for (int i = 0; i < 8; ++i) printf(" %llx", (unsigned long long)h[i]); printf("\nKAT b"); for (int i = 0; i < 8; ++i) printf(" %llx", (unsigned long long)h[8 + i]);
  printf("\nKAT mul"); for (int i = 0; i < 8; ++i) printf(" %llx", (unsigned long long)r[i]); printf("\nKAT chain100"); for (int i = 0; i < 8; ++i) printf(" %llx", (unsigned long long)r[8 + i]); printf("\n");
  run<1>(d_in, d_out, cus); run<2>(d_in, d_out, cus);
  return 0;
}
