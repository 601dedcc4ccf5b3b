// Integer-VALU issue-rate micro-benchmark for gfx950 (MI355X).
// Decides the limb radix for the Fq/Fr Montgomery kernels (SURVEY.md §7 "Hard parts").
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_int valu_int.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 512;
constexpr int UNROLL = 16;  // independent instructions per iteration (8 chains x 2)

// Each body issues 16 instructions over 8 independent register chains.
#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

#define K_BEGIN(name) \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
  uint32_t a[8], b[8]; uint64_t d[8]; double f[8]; \
  for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 17 + i) | 1; b[i] = a[i] * 2654435761u + i; d[i] = ((uint64_t)a[i] << 32) | b[i]; f[i] = 1.0 + 1e-9 * (double)(a[i] & 1023); } \
  uint32_t ca = seed | 3, cb = (seed * 77u) | 5; double cf = 1.0000001; (void)ca; (void)cb; (void)cf; \
  uint64_t t0 = __builtin_amdgcn_s_memtime(); \
  for (int it = 0; it < ITERS; ++it) {

#define K_END \
  } \
  uint64_t t1 = __builtin_amdgcn_s_memtime(); \
  uint32_t acc = 0; for (int i = 0; i < 8; ++i) { acc ^= a[i] ^ b[i] ^ (uint32_t)d[i] ^ (uint32_t)(d[i] >> 32) ^ (uint32_t)__double_as_longlong(f[i]); } \
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc; \
  if (threadIdx.x == 0) out[gridDim.x * blockDim.x + blockIdx.x] = (uint32_t)(t1 - t0); \
}

#define I_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(ca), "v"(cb) : "vcc");
K_BEGIN(k_mad_u64_u32) BODY8(I_MAD64) K_END

#define I_MAD64S(i) asm volatile("v_mad_u64_u32 %0, %3, %1, %2, %0" : "+v"(d[i]) : "v"(ca), "v"(cb), "s"(0ull) : );
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(cb));
K_BEGIN(k_mul_lo_u32) BODY8(I_MULLO) K_END

#define I_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(cb));
K_BEGIN(k_mul_hi_u32) BODY8(I_MULHI) K_END

#define I_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(cb), "v"(ca));
K_BEGIN(k_mad_u32_u24) BODY8(I_MAD24) K_END

#define I_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(cb));
K_BEGIN(k_mul_hi_u32_u24) BODY8(I_MULHI24) K_END

#define I_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(cb));
K_BEGIN(k_add_u32) BODY8(I_ADD) K_END

#define I_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(ca), "v"(cb) : "vcc");
K_BEGIN(k_add_co_addc_pair) BODY8(I_ADDCO) K_END

#define I_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(cb), "v"(ca));
K_BEGIN(k_add3_u32) BODY8(I_ADD3) K_END

#define I_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(a[i]) : "v"(cb));
K_BEGIN(k_alignbit_b32) BODY8(I_ALIGN) K_END

#define I_LSHR64(i) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(d[i]));
K_BEGIN(k_lshrrev_b64) BODY8(I_LSHR64) K_END

#define I_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
K_BEGIN(k_fma_f64) BODY8(I_FMA64) K_END

#define I_MUL64F(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
K_BEGIN(k_mul_f64) BODY8(I_MUL64F) K_END

#define I_DOT2(i) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(cb), "v"(ca));
K_BEGIN(k_dot2_u32_u16) BODY8(I_DOT2) K_END

#define I_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(cb), "v"(ca));
K_BEGIN(k_dot4_u32_u8) BODY8(I_DOT4) K_END

#define I_MADU16(i) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(cb), "v"(ca));
K_BEGIN(k_mad_u32_u16) BODY8(I_MADU16) K_END

#define I_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(cb) : );
K_BEGIN(k_cndmask_b32) BODY8(I_CNDMASK) K_END

#define I_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i]) : "v"(d[(i+1)&7]));
K_BEGIN(k_lshl_add_u64) BODY8(I_LSHLADD64) K_END

#define I_PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(cb));
K_BEGIN(k_pk_add_u16) BODY8(I_PKADD) K_END

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int instr_per_slot; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s, CUs %d, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_add_co+v_addc_co (pair)", k_add_co_addc_pair, 2},
    {"v_cndmask_b32", k_cndmask_b32, 1}, {"v_alignbit_b32", k_alignbit_b32, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_pk_add_u16", k_pk_add_u16, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_mad_u32_u16", k_mad_u32_u16, 1},
    {"v_dot2_u32_u16", k_dot2_u32_u16, 1}, {"v_dot4_u32_u8", k_dot4_u32_u8, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
  };
  uint32_t* out; CK(hipMalloc(&out, sizeof(uint32_t) * (size_t)(cus * 8 * 256 + cus * 8 + 1024)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-28s %8s %12s %12s %14s\n", "instruction", "waves/SIMD", "cyc/inst(1wave s_memtime)", "us", "G lane-ops/s");
  for (auto& e : es) {
    for (int wps : {1, 2, 4}) {  // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
      int blocks = cus * wps;
      e.k<<<blocks, 256>>>(out, 12345u);  // warm
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const int reps = 5;
      for (int r = 0; r < reps; ++r) e.k<<<blocks, 256>>>(out, 12345u + r);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      std::vector<uint32_t> cyc(blocks);
      CK(hipMemcpy(cyc.data(), out + (size_t)blocks * 256, blocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
      double avg = 0; for (auto c : cyc) avg += c; avg /= blocks;
      double n_inst = (double)ITERS * UNROLL * e.instr_per_slot;
      // s_memtime counts shader cycles(?) at a fixed 100MHz-derived rate; report raw ticks per instruction per wave
      double ticks_per_inst = avg / n_inst;
      double laneops = n_inst * 64.0 * 4 * blocks / (ms * 1e-3) / 1e9;
      printf("%-28s %8d %12.3f %12.2f %14.1f\n", e.name, wps, ticks_per_inst, ms * 1e3, laneops);
    }
  }
  return 0;
}
