// fr29.h — Fr on 9 x 29-bit limbs (Montgomery, R = 2^261) for the large-tile NTT kernels (ntt.hip).
//
// Why a second representation (the same reason as fp28.h for Fq): with 32-bit limbs every v_mad_u64_u32 of a product needs carry catches —
// 128 mads + ~150 v_addc per product, 128 G products/s measured; with 29-bit limbs a 64-bit accumulator holds a whole column, 162 mads +
// 44 simple instructions, 155-177 G products/s (tools/ubench/fr29_mul_bench.hip, profiles/r03_fr29_mul_bench.txt).  Additions are lazy
// (9 v_add, limbs grow), differences add a padded multiple of r first (FR29_PAD) so that no limb goes negative, and the 9 x 29 = 261 bits
// leave room for values up to 445 r: a butterfly needs no conditional subtraction, the product brings everything back below 1.1 r.
//
// Classes.  "normalised": limbs 0..7 < 2^29 (the top limb holds what is left).  A product accepts a multiplicand with limbs < 2^31.4 and a
// normalised multiplier; value bounds a < A r, b < B r give a normalised result < (A B / 445 + 1) r.
// The data in HBM stays what it is (8 x 32-bit words, Montgomery with R = 2^256): repacking the words into 29-bit limbs does not change the
// number, and multiplying a number X by a table entry w * 2^261 under this product gives X * w — the TABLES carry the second Montgomery form.
#pragma once
#include "fp.h"
#include "fr29_mont_gen.h"

namespace aleo_mi355x {

struct F29 { uint32_t v[9]; };
static constexpr uint32_t M29 = 0x1fffffffu;

// 8 x 32-bit words -> 9 x 29-bit limbs of the same number (and back; `to` wants a normalised value below 2^256)
__device__ __forceinline__ F29 f29_from_words(const uint32_t (&w)[8]) {
  F29 r;
  r.v[0] = w[0] & M29;
#pragma unroll
  for (int i = 1; i < 8; ++i) r.v[i] = __funnelshift_r(w[i - 1], w[i], 32 - 3 * i) & M29;      // bits [29 i, 29 i + 29) = (w[i] : w[i-1]) >> (32 - 3 i)
  r.v[8] = w[7] >> 8;
  return r;
}
__device__ __forceinline__ void f29_to_words(const F29& a, uint32_t (&w)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = (a.v[i] >> (3 * i)) | (a.v[i + 1] << (29 - 3 * i));           // word i starts at bit 32 i = limb i bit 3 i
}
__device__ __forceinline__ F29 f29_from_fr(const Fr& a) { return f29_from_words(a.v); }
__device__ __forceinline__ Fr f29_to_fr(const F29& a) { Fr r; f29_to_words(a, r.v); return r; }

__device__ __forceinline__ F29 f29_add(const F29& a, const F29& b) { F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
  return r; }
// a - b + 19 r: b with limbs <= 2^30 - 2 and a value below 18 r (top limb!); a with limbs <= 2^30
__device__ __forceinline__ F29 f29_sub_pad(const F29& a, const F29& b) { F29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + FR29_PAD[i] - b.v[i];
  return r; }
__device__ __forceinline__ void f29_normalise(F29& a) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const uint32_t t = a.v[i] + c; c = t >> 29; a.v[i] = t & M29; }
  a.v[8] += c;
}
// normalised value below 445 r -> normalised value below 3 r: subtracts q r with q = mulhi(top limb, FR29_QMAGIC) <= floor(value / r)
__device__ __forceinline__ void f29_reduce_partial(F29& a) {
  const uint32_t q = __umulhi(a.v[8], FR29_QMAGIC);
  int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc += (int64_t)a.v[i] - (int64_t)((uint64_t)q * FR29_P[i]); a.v[i] = (uint32_t)acc & M29; acc >>= 29; }
  a.v[8] = (uint32_t)(acc + (int64_t)a.v[8] - (int64_t)((uint64_t)q * FR29_P[8]));
}
__device__ __forceinline__ F29 f29_mul(const F29& a, const F29& b) { F29 r = a; mont29_mul_inplace(r.v, b.v); return r; }

// table entries: packed (32 bytes, a canonical number) or unpacked (9 limbs at a 48-byte stride: the inner twiddles, read 7 per 8 elements per group)
__device__ __forceinline__ F29 f29_load_packed(const void* p) { return f29_from_fr(load_fp<Fr>(p)); }
__device__ __forceinline__ F29 f29_load48(const void* p) {
  const uint4* q = (const uint4*)p; const uint4 a = q[0], b = q[1]; F29 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w; r.v[8] = ((const uint32_t*)p)[8];
  return r;
}
struct F29Arg { uint32_t v[9]; };

}  // namespace aleo_mi355x
