// fp28.h — Fq on 14 x 28-bit limbs (R' = 2^392): the representation of the bucket-accumulation inner loop.
//
// Why a second representation (fp.h keeps 12 x 32-bit limbs, snarkVM's layout, for everything that crosses the ABI).
// With 32-bit limbs every v_mad_u64_u32 needs a v_addc to catch its carry-out: 288 + 288 of the 649 instructions of a
// product.  With 28-bit limbs a 64-bit accumulator holds a whole column, so a product is 392 mads + 71 simple instructions
// (tools/gen_fp28_asm.py): 81 G products/s against 60.7 (profiles/r01_fq28_mul_bench.txt).  Additions and subtractions
// become carry-free limb-wise operations; what they cost instead is bookkeeping of LIMB bounds next to the value bounds.
//
// Value bounds: R'/q ~ 2^15.2, so a product of a < A*q, b < B*q is < (A*B/38000 + 1)*q — every product below is < 2q.
// Limb bounds ("class Lk": every limb < k * 2^28; L1 = exact base-2^28 digits, what a product or normalise() returns).
// A column of a product holds at most 14 terms a_i*b_j plus 14 terms m_i*p_j (< 2^56 each) plus a carry-in < 2^36, all in
// one 64-bit accumulator, so   14 * ka * kb + 14 < 256   for a La x Lb product (e.g. L3 x L3: 140, L4 x L3: 182);
// a squaring doubles the cross terms: 7 * 2 * k^2 + 14 < 256 (L3: 140, L4: 238); muladd adds both products' terms.
// sub<K, S>(a, b) = a - b + K*q limb-wise, with K*q spread so that every limb of the constant is >= S * 2^28 - S:
// needs b's limbs < S * 2^28 - S + (K*q)_i and b < K*q as a value; the result's limbs are < a_max + (S + 1) * 2^28.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fp.h"
#include "ec.h"
#include "fp28_mont_gen.h"

namespace aleo_mi355x {

struct F28 {
  static constexpr int N = 14;
  static constexpr uint32_t MASK = 0x0fffffffu;
  uint32_t v[N];
};

struct Limbs14 { uint32_t v[14]; };
static constexpr Limbs14 Q28 = {{0x00000001u, 0x008c0000u, 0x00000085u, 0x05d44300u, 0x0800170bu, 0x02fba094u, 0x0f1ef362u, 0x000f5138u, 0x0a22d9f3u,
                                 0x0a1493b1u, 0x0b05c06cu, 0x010eac63u, 0x0a4617c5u, 0x00001ae3u}};
static constexpr Limbs14 ONE28 = {{0x0fff67acu, 0x020fffffu, 0x0fb0d727u, 0x0e9203ffu, 0x0249b0e4u, 0x0e172345u, 0x0955d771u, 0x02bf89aau, 0x0b2833b2u,
                                   0x098e116bu, 0x07dc5c97u, 0x00d43e93u, 0x02e3314bu, 0x000003b4u}};      // 2^392 mod q
static constexpr Limbs14 R32_28 = {{0x0fffff68u, 0x0cdfffffu, 0x0fffb102u, 0x09f837ffu, 0x0ff25140u, 0x0a98a7d3u, 0x059f7db3u, 0x06e7c630u, 0x0b4e97b7u,
                                    0x03c84e87u, 0x0495bf80u, 0x0f49a4cfu, 0x0661e2fdu, 0x000008d6u}};     // 2^384 mod q (plain digits)
// 2^392 mod q as 12 x 32-bit limbs: Fq::mul(X, K256_32) turns x * 2^384 into x * 2^392
static constexpr Limbs<12> K256_32 = {{0xffff67acu, 0x2720ffffu, 0x3fffb0d7u, 0xb0e4e920u, 0x72345249u, 0x55d771e1u, 0x2bf89aa9u, 0xbb2833b2u, 0x9798e116u,
                                       0xe937dc5cu, 0x314b0d43u, 0x003b42e3u}};

// K*q with the borrow spread described above: c_0 = d_0 + S*2^28, c_i = d_i + S*2^28 - S, c_13 = d_13 - S
constexpr Limbs14 spread_kq(uint32_t K, uint32_t S) {
  Limbs14 d{}; uint64_t carry = 0;
  for (int i = 0; i < 14; ++i) { uint64_t t = (uint64_t)Q28.v[i] * K + carry; d.v[i] = (uint32_t)(t & F28::MASK); carry = t >> 28; }
  Limbs14 c{};
  for (int i = 0; i < 14; ++i) c.v[i] = d.v[i] + (i < 13 ? S << 28 : 0u) - (i > 0 ? S : 0u);
  return c;
}

__device__ __forceinline__ F28 f28_const(const Limbs14& k) { F28 r; for (int i = 0; i < 14; ++i) r.v[i] = k.v[i]; return r; }
__device__ __forceinline__ F28 f28_mul(const F28& a, const F28& b) { F28 r = a; mont28_mul_inplace(r.v, b.v); return r; }
__device__ __forceinline__ F28 f28_sqr(const F28& a) { F28 r = a; mont28_sqr_inplace(r.v); return r; }
// (a*b + c*d) * 2^-392 under ONE Montgomery reduction (588 mads instead of 2 x 392)
__device__ __forceinline__ F28 f28_muladd(const F28& a, const F28& b, const F28& c, const F28& d) { F28 r = a; mont28_muladd_inplace(r.v, b.v, c.v, d.v); return r; }
__device__ __forceinline__ F28 f28_add(const F28& a, const F28& b) {
  F28 r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}
template <uint32_t K, uint32_t S> __device__ __forceinline__ F28 f28_sub(const F28& a, const F28& b) {
  constexpr Limbs14 c = spread_kq(K, S);
  static_assert(c.v[13] < 0x80000000u, "K too small for this spread");
  F28 r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.v[i] = a.v[i] + c.v[i] - b.v[i];
  return r;
}
// carry-propagate to exact base-2^28 digits (the value must be < 2^392; the top limb keeps what is left)
__device__ __forceinline__ F28 f28_normalise(const F28& a) {
  F28 r = a;
#pragma unroll
  for (int i = 0; i < 13; ++i) { r.v[i + 1] += r.v[i] >> 28; r.v[i] &= F28::MASK; }
  return r;
}
// a == 0 (mod q) for an exact-digit value < 2q
__device__ __forceinline__ bool f28_is_zero_mod_lt2q(const F28& a) {
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 14; ++i) { z |= a.v[i]; e |= a.v[i] ^ Q28.v[i]; }
  return z == 0 || e == 0;
}

// ---- conversions (12 x 32-bit Montgomery R = 2^384  <->  14 x 28-bit Montgomery R' = 2^392) ------------------------
__device__ __forceinline__ F28 relimb_32_to_28(const Fq& x) {      // same integer, value < 2^384
  F28 r;
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int bit = 28 * i, j = bit >> 5, sh = bit & 31;
    uint32_t w = j < 12 ? x.v[j] >> sh : 0u;
    if (sh > 4 && j + 1 < 12) w |= x.v[j + 1] << (32 - sh);
    r.v[i] = i < 13 ? (w & F28::MASK) : w;
  }
  return r;
}
__device__ __forceinline__ Fq relimb_28_to_32(const F28& a) {      // exact digits, value < 2^384
  Fq r;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int bit = 32 * j, i = bit / 28, sh = bit % 28;
    uint32_t w = a.v[i] >> sh;
    if (i + 1 < 14) w |= a.v[i + 1] << (28 - sh);
    if (sh > 24 && i + 2 < 14) w |= a.v[i + 2] << (56 - sh);
    r.v[j] = w;
  }
  return r;
}
// x * 2^384 (canonical or lazily reduced, < 16q)  ->  x * 2^392, exact digits, canonical
__device__ __forceinline__ F28 f28_from_fq(const Fq& x) {
  Fq k; for (int i = 0; i < 12; ++i) k.v[i] = K256_32.v[i];
  return relimb_32_to_28(Fq::reduce(Fq::mul(x, k)));
}
// x * 2^392 (class N or L3, value < 64q)  ->  x * 2^384 as a lazily reduced Fq (< 2q)
__device__ __forceinline__ Fq f28_to_fq(const F28& a) { return relimb_28_to_32(f28_mul(f28_const(R32_28), a)); }

// 112-byte rows: x'[14] | y'[14]
__device__ __forceinline__ void load_affine28(const void* p, F28& x, F28& y) {
  const uint4* s = (const uint4*)p; uint32_t w[28];
#pragma unroll
  for (int i = 0; i < 7; ++i) { uint4 t = s[i]; w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w; }
#pragma unroll
  for (int i = 0; i < 14; ++i) { x.v[i] = w[i]; y.v[i] = w[14 + i]; }
}
__device__ __forceinline__ void store_affine28(void* p, const F28& x, const F28& y) {
  uint32_t w[28];
#pragma unroll
  for (int i = 0; i < 14; ++i) { w[i] = x.v[i]; w[14 + i] = y.v[i]; }
  uint4* d = (uint4*)p;
#pragma unroll
  for (int i = 0; i < 7; ++i) d[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ---- the mixed addition of the accumulation loop (EFD madd-2008-s, same formulas as ec.h xyzz_madd_fast) -----------------
// Invariant of acc between additions: X exact digits < 12q; Y < 2q as exact digits (or the first point's 2q - y, class L2);
// ZZ, ZZZ exact digits < 2q.  x2, y2: table entries, canonical exact digits (y2 may be the L2 negation 2q - y).
// Returns false — acc untouched — when P == +-acc (ZZ3 == 0 mod q); the caller finishes the slice with the general 32-bit code.
struct XYZZ28 { F28 X, Y, ZZ, ZZZ; };

__device__ __forceinline__ bool xyzz28_madd_fast(XYZZ28& acc, const F28& x2, const F28& y2) {
  // operand order: the product block works in place on its FIRST argument, so the argument that dies there goes first
  F28 U2 = f28_mul(x2, acc.ZZ);                              // < 2q
  F28 S2 = f28_mul(y2, acc.ZZZ);                             // L2 x L1 -> < 2q
  F28 P = f28_sub<16, 1>(U2, acc.X);                         // U2 + 16q - X1 < 18q, class L3
  F28 R = f28_sub<4, 2>(S2, acc.Y);                          // Y1 limbs < 2^29 - 2; S2 + 4q - Y1 < 6q, class L4
  F28 PP = f28_sqr(P);                                       // L3 squared: 140 < 256; 324/38000 + 1 -> < 2q
  F28 ZZ3 = f28_mul(acc.ZZ, PP);                             // < 2q (copy: acc must survive a failed check)
  if (__builtin_expect(f28_is_zero_mod_lt2q(ZZ3), 0)) return false;
  F28 PPP = f28_mul(P, PP);                                  // L3 x L1 -> < 2q
  F28 Q = f28_mul(acc.X, PP);                                // < 2q
  F28 RR = f28_sqr(R);                                       // L4 squared: 238 < 256 -> < 2q
  F28 t0 = f28_sub<4, 1>(RR, PPP);                           // < 6q, class L3
  F28 X3 = f28_normalise(f28_sub<6, 2>(t0, f28_add(Q, Q)));  // 2Q: class L2, < 4q; X3 < 12q; exact digits (it is the next P's subtrahend)
  F28 t1 = f28_sub<16, 1>(Q, X3);                            // Q + 16q - X3 < 18q, class L3
  F28 nY = f28_sub<4, 2>(f28_const(Limbs14{}), acc.Y);       // 4q - Y1 <= 4q, class L3
  acc.Y = f28_muladd(R, t1, nY, PPP);                        // R*t1 - Y1*PPP: 14*(4*3 + 3*1) + 14 = 224 < 256; (108 + 8)/38000 + 1 -> < 2q
  acc.X = X3;
  acc.ZZ = ZZ3;
  acc.ZZZ = f28_mul(acc.ZZZ, PPP);                           // < 2q
  return true;
}

// ---- XYZZ points stored in the 28-bit form: 224 bytes, X[14] | Y[14] | ZZ[14] | ZZZ[14] --------------------------------
// Stored invariant: X exact digits < 12q; Y class L3, < 6q; ZZ, ZZZ exact digits < 2q; infinity <=> ZZ all zero.
__device__ __forceinline__ F28 load_f28(const void* p) {
  F28 r; const uint2* s = (const uint2*)p;
#pragma unroll
  for (int i = 0; i < 7; ++i) { uint2 t = s[i]; r.v[2 * i] = t.x; r.v[2 * i + 1] = t.y; }
  return r;
}
__device__ __forceinline__ void store_f28(void* p, const F28& a) {
  uint2* d = (uint2*)p;
#pragma unroll
  for (int i = 0; i < 7; ++i) d[i] = make_uint2(a.v[2 * i], a.v[2 * i + 1]);
}
__device__ __forceinline__ bool f28_is_zero_raw(const F28& a) { uint32_t z = 0; for (int i = 0; i < 14; ++i) z |= a.v[i]; return z == 0; }
__device__ __forceinline__ void store_xyzz28(void* p, const XYZZ28& a) {
  char* c = (char*)p; store_f28(c, a.X); store_f28(c + 56, a.Y); store_f28(c + 112, a.ZZ); store_f28(c + 168, a.ZZZ);
}
__device__ __forceinline__ void store_xyzz28_from32(void* p, const XYZZ& a, bool inf) {     // off the hot path
  XYZZ28 r;
  if (inf || a.ZZ.is_zero_raw() || a.ZZ.is_zero_mod()) { r.X = f28_const(Limbs14{}); r.Y = r.X; r.ZZ = r.X; r.ZZZ = r.X; }
  else { r.X = f28_from_fq(a.X); r.Y = f28_from_fq(a.Y); r.ZZ = f28_from_fq(a.ZZ); r.ZZZ = f28_from_fq(a.ZZZ); }
  store_xyzz28(p, r);
}
__device__ __forceinline__ XYZZ load_xyzz_from28(const void* p) {
  const char* c = (const char*)p; XYZZ r;
  F28 zz = load_f28(c + 112);
  if (f28_is_zero_raw(zz)) return xyzz_infinity();
  r.X = f28_to_fq(load_f28(c)); r.Y = f28_to_fq(load_f28(c + 56)); r.ZZ = f28_to_fq(zz); r.ZZZ = f28_to_fq(load_f28(c + 168));
  return r;
}

// ---- lane-pair cooperative addition in the 28-bit form (same level plan as ec.h xyzz_add_pair) ---------------------------
__device__ __forceinline__ F28 f28_xchg(const F28& a) {
  F28 r;
#ifdef ALEO_XCHG_BPERMUTE
#pragma unroll
  for (int i = 0; i < 14; ++i) r.v[i] = (uint32_t)__shfl_xor((int)a.v[i], 1);      // (rounds 1-3: compiled to ds_bpermute_b32 — an LDS-crossbar round trip per limb)
#else
#pragma unroll
  for (int i = 0; i < 14; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xF, 0xF, true);
#endif
  return r;
}
__device__ __forceinline__ F28 f28_sel(bool take_b, const F28& a, const F28& b) {
  F28 r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.v[i] = take_b ? b.v[i] : a.v[i];
  return r;
}
// 2P for a stored 28-bit XYZZ point (EFD dbl-2008-s-1, a = 0), the same-point case of the pair addition.  Both lanes of the pair run
// the whole doubling on the same operands (8.5 product times, no exchanges) and each stores its half of the result: the case is
// rare on dense data but systematic on sparse data, where the running sum of a bucket chunk meets an unchanged `run` a second time
// right after its first non-empty bucket (acc == run -> 2 run).  It used to leave through a 32-bit out-of-line routine on one lane
// (~40 us, once per chunk): 4 x 2^15 sparse buckets took 339 us against 86 us for dense ones.
// Bounds: X exact < 12q, Y class L3 < 6q, ZZ / ZZZ exact < 2q (the stored invariant); results satisfy it again.
__device__ __forceinline__ void xyzz28_double_both(const char* pa, char* out) {
  const bool odd = threadIdx.x & 1;
  const F28 X = load_f28(pa), Y = load_f28(pa + 56), ZZ = load_f28(pa + 112), ZZZ = load_f28(pa + 168);
  const F28 U = f28_normalise(f28_add(Y, Y));                            // 2Y < 12q, exact digits
  const F28 V = f28_sqr(U);                                              // 144/38000 + 1 -> < 2q
  const F28 ZZ3 = f28_mul(ZZ, V);                                        // < 2q
  if (__builtin_expect(f28_is_zero_mod_lt2q(ZZ3), 0)) {                  // Y == 0: a 2-torsion point doubles to the identity
    uint4* d4 = (uint4*)(out + (odd ? 112 : 0));
#pragma unroll
    for (int i = 0; i < 7; ++i) d4[i] = make_uint4(0, 0, 0, 0);
    return;
  }
  const F28 W = f28_mul(U, V);                                           // < 2q
  const F28 S = f28_mul(X, V);                                           // < 2q
  const F28 XX = f28_sqr(X);                                             // < 2q
  const F28 M = f28_add(f28_add(XX, XX), XX);                            // 3 X^2: class L3, < 6q
  const F28 MM = f28_sqr(M);                                             // L3 squared: 140 < 256; 36/38000 + 1 -> < 2q
  const F28 X3 = f28_normalise(f28_sub<4, 2>(MM, f28_add(S, S)));        // MM + 4q - 2S < 6q, exact digits
  const F28 t = f28_sub<16, 1>(S, X3);                                   // S + 16q - X3 < 18q, class L3
  const F28 nY = f28_sub<8, 4>(f28_const(Limbs14{}), Y);                 // 8q - Y <= 8q, class L5
  const F28 Y3 = f28_muladd(M, t, nY, W);                                // M t - W Y: 14 * (3*3 + 5*1) + 14 = 210 < 256; (108 + 16)/38000 + 1 -> < 2q
  const F28 ZZZ3 = f28_mul(ZZZ, W);                                      // < 2q
  if (odd) { store_f28(out, X3); store_f28(out + 56, Y3); }
  else { store_f28(out + 112, ZZ3); store_f28(out + 168, ZZZ3); }
}
// exact-digit value equals k*q for k in {1, 2, 3}
__device__ __forceinline__ bool f28_is_small_multiple_of_q(const F28& a) {
  constexpr Limbs14 q1 = spread_kq(1, 0), q2 = spread_kq(2, 0), q3 = spread_kq(3, 0);
  uint32_t e1 = 0, e2 = 0, e3 = 0;
#pragma unroll
  for (int i = 0; i < 14; ++i) { e1 |= a.v[i] ^ q1.v[i]; e2 |= a.v[i] ^ q2.v[i]; e3 |= a.v[i] ^ q3.v[i]; }
  return e1 == 0 || e2 == 0 || e3 == 0;
}
__device__ __forceinline__ void xyzz28_add_pair(const char* pa, const char* pb, char* out) {
  const bool odd = threadIdx.x & 1;
  const F28 zzA = load_f28(pa + 112), zzB = load_f28(pb + 112);
  const bool infA = f28_is_zero_raw(zzA), infB = f28_is_zero_raw(zzB);
  if (infA || infB) {              // pair-uniform: both lanes see the same two points
    const char* src = infB ? pa : pb;          // A + O = A ; O + B = B ; O + O = O (either)
    if (src != out) {              // each lane copies half of the 224 bytes
      const uint4* s4 = (const uint4*)(src + (odd ? 112 : 0)); uint4* d4 = (uint4*)(out + (odd ? 112 : 0));
#pragma unroll
      for (int i = 0; i < 7; ++i) d4[i] = s4[i];
    }
    return;
  }
  const char* own = odd ? pb : pa; const char* oth = odd ? pa : pb;
  F28 u = f28_mul(load_f28(own), odd ? zzA : zzB);                              // a: U1 = X1*ZZ2   b: U2 = X2*ZZ1   (< 2q)
  F28 s = f28_mul(load_f28(oth + 168), load_f28(own + 56));                     // a: S1 = ZZZ2*Y1  b: S2 = ZZZ1*Y2  (L1 x L3)
  F28 pu = f28_xchg(u), ps = f28_xchg(s);
  F28 U1 = f28_sel(odd, u, pu), U2 = f28_sel(odd, pu, u), S1 = f28_sel(odd, s, ps), S2 = f28_sel(odd, ps, s);
  F28 P = f28_sub<4, 1>(U2, U1), R = f28_sub<4, 1>(S2, S1);                      // < 6q, class L3
  F28 t3 = f28_sqr(f28_sel(odd, P, R));                                        // a: PP   b: RR   (L3 squared: 140 < 256)
  F28 t4 = f28_mul(load_f28(pa + (odd ? 168 : 112)), load_f28(pb + (odd ? 168 : 112)));   // a: ZZ1*ZZ2   b: ZZZ1*ZZZ2
  F28 pt3 = f28_xchg(t3);
  F28 PP = f28_sel(odd, t3, pt3), RR = f28_sel(odd, pt3, t3);
  F28 t5 = f28_mul(f28_sel(odd, P, U1), PP);                                   // a: PPP  b: Q
  F28 pt5 = f28_xchg(t5);
  F28 PPP = f28_sel(odd, t5, pt5), Q = f28_sel(odd, pt5, t5);
  F28 X3 = f28_normalise(f28_sub<6, 2>(f28_sub<4, 1>(RR, PPP), f28_add(Q, Q)));  // < 12q, exact digits
  F28 t6 = f28_mul(f28_sel(odd, t4, S1), f28_sel(odd, PP, PPP));               // a: ZZ3  b: SP
  F28 pt4 = f28_xchg(t4);                                                      // a receives ZZZ12
  F28 t7 = f28_mul(f28_sel(odd, pt4, R), f28_sel(odd, PPP, f28_sub<16, 1>(Q, X3)));   // a: ZZZ3  b: Rt (L3 x L3: 140 < 256)
  // same-x case: ZZ3 == 0 mod q, seen by the even lane.  Same point (S1 == S2 mod q: S2 + 2q - S1 is q, 2q or 3q) -> doubling;
  // opposite points -> the identity.  Both in the 28-bit form, in line.
  int z = (!odd && f28_is_zero_mod_lt2q(t6)) ? 1 : 0;
  z = __shfl(z, (int)(threadIdx.x & 63u & ~1u));
  if (__builtin_expect(z, 0)) {
    if (f28_is_small_multiple_of_q(f28_normalise(f28_sub<2, 1>(S2, S1)))) xyzz28_double_both(pa, out);
    else {
      uint4* d4 = (uint4*)(out + (odd ? 112 : 0));
#pragma unroll
      for (int i = 0; i < 7; ++i) d4[i] = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  if (odd) { store_f28(out, X3); store_f28(out + 56, f28_sub<4, 1>(t7, t6)); }   // Y3 = Rt - SP: class L3, < 6q
  else { store_f28(out + 112, t6); store_f28(out + 168, t7); }
}

// ---- lane-QUAD cooperative addition: the same fourteen products in four levels instead of seven ---------------------------------------
// For the chains that are pure latency (slice tree, chunk sums, masked sums, segment folds at <= one wave per SIMD).  Lanes 4k..4k+3 share one
// addition A + B; lane q of the quad computes
//   level 1   q0: U1 = X1 ZZ2     q1: U2 = X2 ZZ1      q2: S1 = ZZZ2 Y1     q3: S2 = ZZZ1 Y2
//   level 2   q0: PP = P P        q1: ZZ1 ZZ2          q2: RR = R R         q3: ZZZ1 ZZZ2           (P = U2 - U1 in q0, R = S2 - S1 in q2)
//   level 3   q0: PPP = P PP      q1: Q = U1 PP        (q2, q3 repeat q1's product: no divergence)
//   level 4   q0: R (Q - X3)      q1: ZZ3 = ZZ12 PP    q2: S1 PPP           q3: ZZZ3 = ZZZ12 PPP    (X3 = RR - PPP - 2Q in every lane)
// with quad_perm DPP moves between the levels (values never leave the quad's registers).  Same formulas, bounds and stored invariants as
// xyzz28_add_pair; the rare same-x case (doubling / identity) is handed to the pair form on lanes 0, 1 of the quad.
template <int S0, int S1, int S2, int S3> __device__ __forceinline__ F28 f28_qperm(const F28& a) {
  F28 r;
#pragma unroll
  for (int i = 0; i < 14; ++i) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], S0 | (S1 << 2) | (S2 << 4) | (S3 << 6), 0xF, 0xF, true);
  return r;
}
__device__ __forceinline__ void xyzz28_add_quad(const char* pa, const char* pb, char* out) {
  const uint32_t q = threadIdx.x & 3u; const bool odd = q & 1u, hi = q & 2u;
  const F28 zzA = load_f28(pa + 112), zzB = load_f28(pb + 112);
  const bool infA = f28_is_zero_raw(zzA), infB = f28_is_zero_raw(zzB);
  if (infA || infB) {              // quad-uniform: all four lanes see the same two points
    const char* src = infB ? pa : pb;          // A + O = A ; O + B = B ; O + O = O (either)
    if (src != out) {              // each lane copies a quarter of the 224 bytes
      const uint2* s2 = (const uint2*)(src + 56 * q); uint2* d2 = (uint2*)(out + 56 * q);
#pragma unroll
      for (int i = 0; i < 7; ++i) d2[i] = s2[i];
    }
    return;
  }
  const char* own = odd ? pb : pa; const char* oth = odd ? pa : pb;
  // level 1 (operand order as in the pair form: X_own * ZZ_other, ZZZ_other * Y_own)
  const F28 l1a = load_f28(hi ? oth + 168 : own), l1b = hi ? load_f28(own + 56) : (odd ? zzA : zzB);
  const F28 t1 = f28_mul(l1a, l1b);                                             // q0: U1  q1: U2  q2: S1  q3: S2   (< 2q)
  const F28 D = f28_sub<4, 1>(f28_qperm<1, 1, 3, 3>(t1), t1);                   // q0: P = U2 - U1   q2: R = S2 - S1   (< 6q, class L3; q1, q3: unused)
  // level 2
  const F28 m2a = load_f28(pa + (hi ? 168 : 112)), m2b = load_f28(pb + (hi ? 168 : 112));
  const F28 t2 = f28_mul(f28_sel(odd, D, m2a), f28_sel(odd, D, m2b));           // q0: PP  q1: ZZ1 ZZ2  q2: RR  q3: ZZZ1 ZZZ2   (L3 x L3: 140 < 256)
  // level 3
  const F28 U1 = f28_qperm<0, 0, 0, 0>(t1), PP = f28_qperm<0, 0, 0, 0>(t2);
  const F28 t3 = f28_mul(f28_sel(q == 0, U1, D), PP);                           // q0: PPP = P PP   q1 (q2, q3): Q = U1 PP
  const F28 RR = f28_qperm<2, 2, 2, 2>(t2), PPP = f28_qperm<0, 0, 0, 0>(t3), Q = f28_qperm<1, 1, 1, 1>(t3);
  const F28 X3 = f28_normalise(f28_sub<6, 2>(f28_sub<4, 1>(RR, PPP), f28_add(Q, Q)));   // < 12q, exact digits (every lane)
  // level 4
  const F28 R = f28_qperm<2, 2, 2, 2>(D);
  const F28 a4 = f28_sel(q == 0, f28_sel(q == 2, t2, t1), R);                   // q0: R   q1: ZZ12   q2: S1   q3: ZZZ12
  const F28 b4 = f28_sel(q == 0, f28_sel(q == 1, PPP, PP), f28_sub<16, 1>(Q, X3));   // q0: Q - X3 (L3)   q1: PP   q2, q3: PPP
  const F28 t4 = f28_mul(a4, b4);                                               // q0: Rt   q1: ZZ3   q2: SP   q3: ZZZ3
  // same-x case: ZZ3 == 0 mod q, seen by lane 1 of the quad
  int z = (q == 1 && f28_is_zero_mod_lt2q(t4)) ? 1 : 0;
  z = __builtin_amdgcn_mov_dpp(z, 1 | (1 << 2) | (1 << 4) | (1 << 6), 0xF, 0xF, true);
  if (__builtin_expect(z, 0)) {
    if (q < 2) xyzz28_add_pair(pa, pb, out);      // lanes 0, 1 of the quad are a lane pair: the pair form doubles or writes the identity
    return;
  }
  const F28 SP = f28_qperm<2, 2, 2, 2>(t4);
  if (q == 0) { store_f28(out, X3); store_f28(out + 56, f28_sub<4, 1>(t4, SP)); }   // Y3 = Rt - SP: class L3, < 6q
  else if (q != 2) store_f28(out + (q == 1 ? 112 : 168), t4);
}

}  // namespace aleo_mi355x
