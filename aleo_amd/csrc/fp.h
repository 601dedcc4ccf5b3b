// fp.h — device-side prime-field arithmetic for BLS12-377 Fr (8 x u32) and Fq (12 x u32) on gfx950.
//
// Replaces (on the device) snarkvm-fields 0.14.5 fields/src/fp_256.rs / fp_384.rs  [UPSTREAM-RECALL;
// pins: /root/reference/Cargo.lock:2652].  Same Montgomery representation (R = 2^256 / 2^384, little-endian
// limbs), so buffers cross the C ABI without conversion: a u64 limb is two consecutive u32 limbs.
//
// Lazy reduction.  R/p is ~152.6 for Fq and ~13.7 for Fr, so values are carried UNREDUCED in [0, k*p):
//   mul(a, b)  with a < A*p, b < B*p   ->  result < (A*B*p/R + 1) * p        (no final subtraction)
//   add(a, b)                           ->  plain limb addition, no reduction
//   sub<K>(a, b) with b < K*p           ->  a + K*p - b                      (never negative)
// Every call site states its bounds; reduce() brings a value < 16p (Fq) / < 4p (Fr) back to canonical form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fp_mont_gen.h"

namespace aleo_mi355x {

// ---- compile-time helpers on limb arrays ------------------------------------------------------
template <int N> struct Limbs { uint32_t v[N]; };

template <int N> constexpr Limbs<N> limbs_mul_small(const Limbs<N>& a, uint32_t k) {
  Limbs<N> r{}; uint64_t c = 0;
  for (int i = 0; i < N; ++i) { uint64_t t = (uint64_t)a.v[i] * k + c; r.v[i] = (uint32_t)t; c = t >> 32; }
  return r;
}

struct FqParams {
  static constexpr int N = 12;
  // q = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001
  static constexpr Limbs<12> P = {{0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu,
                                   0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u}};
  // R mod q (Montgomery one)
  static constexpr Limbs<12> ONE = {{0xffffff68u, 0x02cdffffu, 0x7fffffb1u, 0x51409f83u, 0x8a7d3ff2u, 0x9f7db3a9u,
                                     0x6e7c6305u, 0x7b4e97b7u, 0x803c84e8u, 0x4cf495bfu, 0xe2fdf49au, 0x008d6661u}};
  // R^2 mod q
  static constexpr Limbs<12> R2 = {{0x9400cd22u, 0xb786686cu, 0xb00431b1u, 0x0329fcaau, 0x62d6b46du, 0x22a5f111u,
                                    0x827dc3acu, 0xbfdf7d03u, 0x41790bf9u, 0x837e92f0u, 0x1e914b88u, 0x006dfccbu}};
  static constexpr int MAX_K = 16;   // largest multiple of p that reduce() accepts (16q < 2^384)
};

struct FrParams {
  static constexpr int N = 8;
  // r = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
  static constexpr Limbs<8> P = {{0x00000001u, 0x0a118000u, 0xd0000001u, 0x59aa76feu, 0x5c37b001u, 0x60b44d1eu, 0x9a2ca556u, 0x12ab655eu}};
  static constexpr Limbs<8> ONE = {{0xfffffff3u, 0x7d1c7fffu, 0x6ffffff2u, 0x7257f50fu, 0x512c0feeu, 0x16d81575u, 0x2bbb9a9du, 0x0d4bda32u}};
  static constexpr Limbs<8> R2 = {{0xb861857bu, 0x25d577bau, 0x8860591fu, 0xcc2c27b5u, 0xe5dc8593u, 0xa7cc008fu, 0xeff1c939u, 0x011fdae7u}};
  static constexpr int MAX_K = 8;    // 8r < 2^256 (r < 2^253)
};

// ---- the field element ---------------------------------------------------------------------
template <class Pm> struct Fp {
  static constexpr int N = Pm::N;
  uint32_t v[N];

  __device__ __forceinline__ static Fp zero() { Fp r; for (int i = 0; i < N; ++i) r.v[i] = 0; return r; }
  __device__ __forceinline__ static Fp one() { Fp r; for (int i = 0; i < N; ++i) r.v[i] = Pm::ONE.v[i]; return r; }
  __device__ __forceinline__ static Fp r2() { Fp r; for (int i = 0; i < N; ++i) r.v[i] = Pm::R2.v[i]; return r; }

  // Montgomery product, lazily reduced (see header).
  __device__ __forceinline__ static Fp mul(const Fp& a, const Fp& b) {
    Fp r = a;    // the asm block works in place (a <- a*b); the copy is elided when `a` is dead afterwards
    if constexpr (N == 12) mont_mul_12_inplace(r.v, b.v); else mont_mul_8_inplace(r.v, b.v);
    return r;
  }
  // Montgomery square: N(N+1)/2 limb products instead of N^2 (fp_mont_gen.h).  Needs a < 2^(32N-1): every lazy bound in
  // use (Fq < 16q < 2^381, Fr < 8r < 2^256 only up to 4r < 2^255) — callers keep Fr squares below 4r.
  __device__ __forceinline__ static Fp sqr(const Fp& a) {
    Fp r = a;
    if constexpr (N == 12) mont_sqr_12_inplace(r.v); else mont_sqr_8_inplace(r.v);
    return r;
  }

  // a + b, no reduction.  Caller guarantees the sum stays below 2^(32N).
  __device__ __forceinline__ static Fp add(const Fp& a, const Fp& b) {
    Fp r = a;    // generated carry chain works in place
    if constexpr (N == 12) fp_add_12(r.v, b.v); else fp_add_8(r.v, b.v);
    return r;
  }
  // a - b as integers mod 2^(32N) (used as a building block; callers add a multiple of p).
  __device__ __forceinline__ static Fp sub_raw(const Fp& a, const Fp& b, uint32_t& borrow_out) {
    Fp r; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint64_t t = (uint64_t)a.v[i] - b.v[i] - br; r.v[i] = (uint32_t)t; br = (uint32_t)(t >> 32) & 1u;
    }
    borrow_out = br; return r;
  }
  // a + K*p - b   (b < K*p  =>  result in [0, a + K*p))
  template <int K> __device__ __forceinline__ static Fp sub(const Fp& a, const Fp& b) {
    static_assert(K == 1 || K == 2 || K == 4 || K == 8, "sub<K>: K*p literals are generated for K in {1,2,4,8}");
    constexpr Limbs<N> kp = limbs_mul_small<N>(Pm::P, (uint32_t)K);
    uint32_t kpe[N]; for (int i = 0; i < N; ++i) kpe[i] = kp.v[i];
    Fp r = a;
    if constexpr (N == 12) {
      if constexpr (K == 1) fp_sub_12_k1(r.v, b.v, kpe); else if constexpr (K == 2) fp_sub_12_k2(r.v, b.v, kpe);
      else if constexpr (K == 4) fp_sub_12_k4(r.v, b.v, kpe); else fp_sub_12_k8(r.v, b.v, kpe);
    } else {
      if constexpr (K == 1) fp_sub_8_k1(r.v, b.v, kpe); else if constexpr (K == 2) fp_sub_8_k2(r.v, b.v, kpe);
      else if constexpr (K == 4) fp_sub_8_k4(r.v, b.v, kpe); else fp_sub_8_k8(r.v, b.v, kpe);
    }
    return r;
  }
  __device__ __forceinline__ static Fp dbl(const Fp& a) { return add(a, a); }

  // if (a >= K*p) a -= K*p
  template <int K> __device__ __forceinline__ static Fp cond_sub(const Fp& a) {
    static_assert(K == 1 || K == 2 || K == 4 || K == 8, "cond_sub<K>: generated for K in {1,2,4,8}");
    constexpr Limbs<N> kp = limbs_mul_small<N>(Pm::P, (uint32_t)K);
    uint32_t kpe[N]; for (int i = 0; i < N; ++i) kpe[i] = kp.v[i];
    Fp r = a;    // one borrow chain + one select per limb (fp_mont_gen.h); the C form costs ~7 instructions per limb
    if constexpr (N == 12) {
      if constexpr (K == 1) fp_cond_sub_12_k1(r.v, kpe); else if constexpr (K == 2) fp_cond_sub_12_k2(r.v, kpe);
      else if constexpr (K == 4) fp_cond_sub_12_k4(r.v, kpe); else fp_cond_sub_12_k8(r.v, kpe);
    } else {
      if constexpr (K == 1) fp_cond_sub_8_k1(r.v, kpe); else if constexpr (K == 2) fp_cond_sub_8_k2(r.v, kpe);
      else if constexpr (K == 4) fp_cond_sub_8_k4(r.v, kpe); else fp_cond_sub_8_k8(r.v, kpe);
    }
    return r;
  }
  // canonical representative of a value < 16p (Fq) / < 8p (Fr)
  __device__ __forceinline__ static Fp reduce(const Fp& a) {
    Fp r = a;
    if constexpr (Pm::MAX_K >= 16) r = cond_sub<8>(r);
    r = cond_sub<4>(r); r = cond_sub<2>(r); r = cond_sub<1>(r);
    return r;
  }
  __device__ __forceinline__ bool is_zero_raw() const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) o |= v[i];
    return o == 0;
  }
  __device__ __forceinline__ bool equals_raw(const Limbs<N>& k) const {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) o |= v[i] ^ k.v[i];
    return o == 0;
  }
  // a == 0 (mod p) for a < 2p: only 0 and p qualify.
  __device__ __forceinline__ bool is_zero_mod_lt2p() const { return is_zero_raw() || equals_raw(Pm::P); }
  // a == 0 (mod p) for a < MAX_K * p
  __device__ __forceinline__ bool is_zero_mod() const { Fp r = reduce(*this); return r.is_zero_raw(); }

  // Montgomery <-> canonical
  __device__ __forceinline__ static Fp to_mont(const Fp& a) { return reduce(mul(a, r2())); }
  __device__ __forceinline__ static Fp from_mont(const Fp& a) {
    Fp o = zero(); o.v[0] = 1; return reduce(mul(a, o));
  }
};

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

// 16-byte vector load/store helpers (coalescing unit on CDNA4: 16 B per lane)
template <class F> __device__ __forceinline__ F load_fp(const void* p) {
  F r; const uint4* s = (const uint4*)p;
#pragma unroll
  for (int i = 0; i < F::N / 4; ++i) { uint4 t = s[i]; r.v[4 * i] = t.x; r.v[4 * i + 1] = t.y; r.v[4 * i + 2] = t.z; r.v[4 * i + 3] = t.w; }
  return r;
}
template <class F> __device__ __forceinline__ void store_fp(void* p, const F& a) {
  uint4* d = (uint4*)p;
#pragma unroll
  for (int i = 0; i < F::N / 4; ++i) d[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

}  // namespace aleo_mi355x
