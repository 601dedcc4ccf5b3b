// g2.hip — BLS12-377 G2 multi-scalar multiplication for MI355X (gfx950): y^2 = x^3 + b' over Fq2 = Fq[u] / (u^2 + 5).
//
// Replaces snarkvm-algorithms 0.14.5 `VariableBase::msm::<G2Affine>` -> `standard::msm` [UPSTREAM-RECALL: variable_base/mod.rs sends
// every curve but BLS12-377 G1 to the standard Pippenger] over snarkvm-curves bls12_377/{fq2,g2}.rs (pin: /root/reference/Cargo.lock:2637).
// The prover never runs a G2 MSM (G2 appears in the verifying key and in SRS setup), so this path is built for
// parity and reach, not for the last percent: it shares the whole scalar side with G1 — signed-digit windows, the two-level
// counting sort, slice sizing and ordering (msm_sort_phase: none of it depends on the group) — and swaps the group law:
// extended Jacobian (XYZZ) over Fq2 on the 32-bit-limb Montgomery blocks of fp.h, one lane per addition, out-of-line field
// calls (a 2 x 12-limb product inlined ten times per addition would not fit the register file).
//
// Layouts at the boundary (snarkVM in-memory): G2Affine {x: Fq2 (c0, c1), y: Fq2, infinity: bool} = 4 x 48 bytes Montgomery +
// flag byte at 192, stride 200 (or 192 without the flag); result G2Projective (Jacobian) {x, y, z: Fq2} = 288 bytes,
// returned affine-normalised (x, y, 1) or (1, 1, 0) for the identity.  HBM: bases n x 192 B; partial sums 384 B per slice.
#include "ctx.h"
#include "ec.h"
#include "fp28.h"
#include "host_field.hpp"
#include "msm_common.h"
#include <vector>

namespace aleo_mi355x {

// ---- Fq2 on the device: components stay below 2q between operations ----------------------------------------------------
struct Fq2 { Fq a, b; };                       // a + b u,  u^2 = -5
struct G2Affine { Fq2 x, y; };                 // 192 bytes
struct XYZZ2 { Fq2 X, Y, ZZ, ZZZ; };           // 384 bytes; infinity <=> ZZ stored as raw zero

__device__ __forceinline__ Fq lt2q(const Fq& x) { return Fq::cond_sub<2>(x); }                       // < 4q -> < 2q
__device__ __forceinline__ Fq2 fq2_zero() { Fq2 r; r.a = Fq::zero(); r.b = Fq::zero(); return r; }
__device__ __forceinline__ Fq2 fq2_one() { Fq2 r; r.a = Fq::one(); r.b = Fq::zero(); return r; }
__device__ __forceinline__ Fq2 fq2_add(const Fq2& x, const Fq2& y) { Fq2 r; r.a = lt2q(Fq::add(x.a, y.a)); r.b = lt2q(Fq::add(x.b, y.b)); return r; }
__device__ __forceinline__ Fq2 fq2_sub(const Fq2& x, const Fq2& y) { Fq2 r; r.a = lt2q(Fq::sub<2>(x.a, y.a)); r.b = lt2q(Fq::sub<2>(x.b, y.b)); return r; }
__device__ __forceinline__ Fq2 fq2_dbl(const Fq2& x) { return fq2_add(x, x); }
__device__ __forceinline__ Fq fq_times5_canonical(const Fq& v) {      // 5 v mod q, v < 2q
  Fq d2 = Fq::dbl(v), d4 = Fq::dbl(d2);                               // < 4q, < 8q
  return Fq::reduce(Fq::add(d4, v));                                  // < 10q -> canonical
}
// (x.a + x.b u)(y.a + y.b u) = (x.a y.a - 5 x.b y.b) + (x.a y.b + x.b y.a) u   (Karatsuba: three base-field products)
__device__ __noinline__ void fq2_mul_ni(Fq2* r, const Fq2* px, const Fq2* py) {
  const Fq2 x = *px, y = *py;
  Fq v0 = Fq::mul(x.a, y.a), v1 = Fq::mul(x.b, y.b);                  // < 2q
  Fq s = Fq::mul(Fq::add(x.a, x.b), Fq::add(y.a, y.b));               // (< 4q)(< 4q): 16/152 + 1 -> < 2q
  Fq2 o;
  o.b = lt2q(Fq::sub<2>(lt2q(Fq::sub<2>(s, v0)), v1));
  o.a = lt2q(Fq::sub<1>(v0, fq_times5_canonical(v1)));                // v0 + q - 5 v1 < 3q
  *r = o;
}
// (a + b u)^2 = (a + b)(a - 5b) + 4ab  +  2ab u   (two base-field products)
__device__ __noinline__ void fq2_sqr_ni(Fq2* r, const Fq2* px) {
  const Fq2 x = *px;
  Fq m = Fq::mul(x.a, x.b);                                           // < 2q
  Fq w = Fq::mul(Fq::add(x.a, x.b), Fq::sub<1>(x.a, fq_times5_canonical(x.b)));      // (< 4q)(< 3q) -> < 2q
  Fq2 o;
  o.b = lt2q(Fq::dbl(m));
  o.a = Fq::reduce(Fq::add(w, Fq::dbl(Fq::dbl(m))));                  // < 10q -> canonical
  *r = o;
}
__device__ __forceinline__ Fq2 fq2_mul(const Fq2& x, const Fq2& y) { Fq2 r; fq2_mul_ni(&r, &x, &y); return r; }
__device__ __forceinline__ Fq2 fq2_sqr(const Fq2& x) { Fq2 r; fq2_sqr_ni(&r, &x); return r; }
__device__ __forceinline__ bool fq2_is_zero(const Fq2& x) { return x.a.is_zero_mod_lt2p() && x.b.is_zero_mod_lt2p(); }
__device__ __forceinline__ bool fq2_is_zero_raw(const Fq2& x) { return x.a.is_zero_raw() && x.b.is_zero_raw(); }

__device__ __forceinline__ Fq2 load_fq2(const void* p) { Fq2 r; r.a = load_fp<Fq>(p); r.b = load_fp<Fq>((const char*)p + 48); return r; }
__device__ __forceinline__ void store_fq2(void* p, const Fq2& x) { store_fp<Fq>(p, x.a); store_fp<Fq>((char*)p + 48, x.b); }
__device__ __forceinline__ XYZZ2 g2_infinity() { XYZZ2 r; r.X = fq2_zero(); r.Y = r.X; r.ZZ = r.X; r.ZZZ = r.X; return r; }
__device__ __forceinline__ bool g2_is_inf(const XYZZ2& p) { return fq2_is_zero_raw(p.ZZ); }
__device__ __forceinline__ XYZZ2 load_xyzz2(const void* p) {
  const char* c = (const char*)p; XYZZ2 r; r.X = load_fq2(c); r.Y = load_fq2(c + 96); r.ZZ = load_fq2(c + 192); r.ZZZ = load_fq2(c + 288); return r;
}
__device__ __forceinline__ void store_xyzz2(void* p, const XYZZ2& a) {
  char* c = (char*)p; store_fq2(c, a.X); store_fq2(c + 96, a.Y); store_fq2(c + 192, a.ZZ); store_fq2(c + 288, a.ZZZ);
}

// 2P (EFD dbl-2008-s-1, a = 0)
__device__ __noinline__ void g2_double_ni(XYZZ2* io) {
  const XYZZ2 p = *io;
  if (g2_is_inf(p)) return;
  Fq2 U = fq2_dbl(p.Y), V = fq2_sqr(U), W = fq2_mul(U, V), S = fq2_mul(p.X, V);
  Fq2 xx = fq2_sqr(p.X), M = fq2_add(fq2_dbl(xx), xx);
  XYZZ2 r;
  r.X = fq2_sub(fq2_sqr(M), fq2_dbl(S));
  r.Y = fq2_sub(fq2_mul(M, fq2_sub(S, r.X)), fq2_mul(W, p.Y));
  r.ZZ = fq2_mul(V, p.ZZ); r.ZZZ = fq2_mul(W, p.ZZZ);
  if (fq2_is_zero(r.ZZ)) r = g2_infinity();                           // y == 0: a 2-torsion point
  *io = r;
}
// acc += b, both XYZZ (EFD add-2008-s); handles the identity, doubling and cancellation
__device__ __noinline__ void g2_add_ni(XYZZ2* pa, const XYZZ2* pb) {
  const XYZZ2 a = *pa, b = *pb;
  if (g2_is_inf(b)) return;
  if (g2_is_inf(a)) { *pa = b; return; }
  Fq2 U1 = fq2_mul(a.X, b.ZZ), U2 = fq2_mul(b.X, a.ZZ), S1 = fq2_mul(a.Y, b.ZZZ), S2 = fq2_mul(b.Y, a.ZZZ);
  Fq2 P = fq2_sub(U2, U1), R = fq2_sub(S2, S1);
  if (fq2_is_zero(P)) {
    if (fq2_is_zero(R)) { XYZZ2 d = a; g2_double_ni(&d); *pa = d; } else *pa = g2_infinity();
    return;
  }
  Fq2 PP = fq2_sqr(P), PPP = fq2_mul(P, PP), Q = fq2_mul(U1, PP);
  XYZZ2 r;
  r.X = fq2_sub(fq2_sub(fq2_sqr(R), PPP), fq2_dbl(Q));
  r.Y = fq2_sub(fq2_mul(R, fq2_sub(Q, r.X)), fq2_mul(S1, PPP));
  r.ZZ = fq2_mul(fq2_mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = fq2_mul(fq2_mul(a.ZZZ, b.ZZZ), PPP);
  *pa = r;
}
// acc += (x, y) affine (EFD madd-2008-s)
__device__ __noinline__ void g2_madd_ni(XYZZ2* pa, const G2Affine* pp) {
  const XYZZ2 a = *pa; const G2Affine q = *pp;
  if (g2_is_inf(a)) { XYZZ2 r; r.X = q.x; r.Y = q.y; r.ZZ = fq2_one(); r.ZZZ = fq2_one(); *pa = r; return; }
  Fq2 U2 = fq2_mul(q.x, a.ZZ), S2 = fq2_mul(q.y, a.ZZZ);
  Fq2 P = fq2_sub(U2, a.X), R = fq2_sub(S2, a.Y);
  if (fq2_is_zero(P)) {
    if (fq2_is_zero(R)) { XYZZ2 d; d.X = q.x; d.Y = q.y; d.ZZ = fq2_one(); d.ZZZ = fq2_one(); g2_double_ni(&d); *pa = d; } else *pa = g2_infinity();
    return;
  }
  Fq2 PP = fq2_sqr(P), PPP = fq2_mul(P, PP), Q = fq2_mul(a.X, PP);
  XYZZ2 r;
  r.X = fq2_sub(fq2_sub(fq2_sqr(R), PPP), fq2_dbl(Q));
  r.Y = fq2_sub(fq2_mul(R, fq2_sub(Q, r.X)), fq2_mul(a.Y, PPP));
  r.ZZ = fq2_mul(a.ZZ, PP);
  r.ZZZ = fq2_mul(a.ZZZ, PPP);
  *pa = r;
}

// ---- the accumulation on 28-bit limbs, one LANE PAIR per slice (round 4) -------------------------------------------------------------------------------
// An Fq2 value lives across the two lanes of a pair: the even lane holds the a-component, the odd lane the b-component, each as 14 x 28-bit limbs in the
// R' = 2^392 Montgomery form of fp28.h.  Sums, differences and normalisations are component-wise, so they are the F28 operations unchanged; a product
//   (x.a + x.b u)(y.a + y.b u) = (x.a y.a - 5 x.b y.b) + (x.a y.b + x.b y.a) u
// is ONE merged block per lane (mont28_muladd: two products under one reduction, 588 mads) after the lanes have swapped copies of their operands:
//   even lane:  x.a * y.a + x.b * (K q - 5 y.b)          odd lane:  x.b * y.a + x.a * y.b
// i.e. three product times of work per Fq2 product — what a one-lane Karatsuba costs — at half the registers per lane (an XYZZ accumulator over Fq2 is
// 112 limbs; one lane per slice with the ten products in line does not fit 256 VGPRs, and the out-of-line 32-bit version this replaces moved every
// operand through scratch memory).  Every operand of a product is normalised first (exact digits, class L1): a column then holds at most
// 14 (1 * 1 + 1 * 7) + 14 < 256 (fp28.h); value bounds are written at each step, with a < A q, b < B q -> a b < (A B / 38000 + 1) q.
// The mixed addition is fp28.h's xyzz28_madd_fast (EFD madd-2008-s) with F28 read as "my component"; P == +-acc (ZZ3 = 0 in BOTH components) leaves the
// loop for the general 32-bit code on the even lane, as in msm.hip.
template <uint32_t K> __device__ __forceinline__ F28 f28_neg5(const F28& v) {      // K q - 5 v for an exact-digit v with 5 v < K q: limbs < 7 * 2^28
  F28 t;
#pragma unroll
  for (int i = 0; i < 14; ++i) t.v[i] = 5u * v.v[i];
  return f28_sub<K, 6>(f28_const(Limbs14{}), t);
}
// my component of x * y; x, y: exact digits; KY: 5 * (value bound of y in q) rounded up to a K with a spread constant
template <uint32_t KY> __device__ __forceinline__ F28 fq2p_mul(const F28& x, const F28& y, bool odd) {
  const F28 ox = f28_xchg(x), oy = f28_xchg(y);
  const F28 b1 = f28_sel(odd, y, oy);                      // even: x.a * y.a        odd: x.b * y.a
  const F28 b2 = f28_sel(odd, f28_neg5<KY>(oy), y);        // even: x.b * (-5 y.b)   odd: x.a * y.b
  return f28_muladd(x, b1, ox, b2);
}
struct XYZZ2P { F28 X, Y, ZZ, ZZZ; };                      // my components of an XYZZ point over Fq2
// acc += (x2, y2).  In: acc.X exact < 12q, acc.Y exact < 6q, acc.ZZ / ZZZ exact < 2q; x2, y2 exact < 2q.  Out: the same.  false (acc untouched): P == +-acc.
__device__ __forceinline__ bool g2p_madd_fast(XYZZ2P& acc, const F28& x2, const F28& y2, bool odd) {
  const F28 U2 = fq2p_mul<16>(x2, acc.ZZ, odd);                                    // (2*2 + 2*16) / 38000 + 1 -> < 2q
  const F28 S2 = fq2p_mul<16>(y2, acc.ZZZ, odd);                                   // < 2q
  const F28 P = f28_normalise(f28_sub<16, 1>(U2, acc.X));                          // U2 + 16q - X1 < 18q
  const F28 R = f28_normalise(f28_sub<8, 1>(S2, acc.Y));                           // S2 + 8q - Y1 < 10q
  const F28 PP = fq2p_mul<96>(P, P, odd);                                          // (18*18 + 18*96) / 38000 + 1 -> < 2q
  const F28 ZZ3 = fq2p_mul<16>(acc.ZZ, PP, odd);                                   // < 2q
  {
    const bool z = f28_is_zero_mod_lt2q(ZZ3);
    if (__builtin_expect(z && (bool)__shfl_xor((int)z, 1), 0)) return false;
  }
  const F28 PPP = fq2p_mul<16>(P, PP, odd);                                        // (18*2 + 18*16) / 38000 + 1 -> < 2q
  const F28 Q = fq2p_mul<16>(acc.X, PP, odd);                                      // (12*2 + 12*16) / 38000 + 1 -> < 2q
  const F28 RR = fq2p_mul<64>(R, R, odd);                                          // (10*10 + 10*64) / 38000 + 1 -> < 2q
  const F28 t0 = f28_sub<4, 1>(RR, PPP);                                           // < 6q, limbs < 3 * 2^28
  const F28 X3 = f28_normalise(f28_sub<6, 2>(t0, f28_add(Q, Q)));                  // 2Q: limbs < 2 * 2^28, < 4q; X3 < 12q, exact digits
  const F28 t1 = f28_normalise(f28_sub<16, 1>(Q, X3));                             // Q + 16q - X3 < 18q
  const F28 RT = fq2p_mul<96>(R, t1, odd);                                         // (10*18 + 10*96) / 38000 + 1 -> < 2q
  const F28 YP = fq2p_mul<16>(acc.Y, PPP, odd);                                    // (6*2 + 6*16) / 38000 + 1 -> < 2q
  acc.Y = f28_normalise(f28_sub<4, 1>(RT, YP));                                    // RT + 4q - YP < 6q
  acc.X = X3;
  acc.ZZ = ZZ3;
  acc.ZZZ = fq2p_mul<16>(acc.ZZZ, PPP, odd);                                       // < 2q
  return true;
}
// 192-byte rows (x.a | x.b | y.a | y.b, 32-bit Montgomery) -> 224-byte rows [x.a | y.a | x.b | y.b] in the 28-bit form: each lane of a pair reads 112 contiguous bytes
__global__ void __launch_bounds__(256) k_g2_rows_to28(const char* __restrict__ src192, char* __restrict__ dst224, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const char* r = src192 + (size_t)i * 192; char* o = dst224 + (size_t)i * 224;
  store_affine28(o, f28_from_fq(load_fp<Fq>(r)), f28_from_fq(load_fp<Fq>(r + 96)));
  store_affine28(o + 112, f28_from_fq(load_fp<Fq>(r + 48)), f28_from_fq(load_fp<Fq>(r + 144)));
}
// snarkVM G2Affine rows (200 bytes: x.c0 | x.c1 | y.c0 | y.c1 | infinity byte + padding) -> 192-byte rows + one flag byte per point; *n_inf counts the
// flagged points (the host drops the flag array when it is zero).  200 = 8 * 25: rows are 8-byte aligned, so a lane moves its row as 25 eight-byte words.
__global__ void __launch_bounds__(256) k_g2_unpack200(const char* __restrict__ rows200, char* __restrict__ xy192, uint8_t* __restrict__ flags, uint32_t* __restrict__ n_inf, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const uint2* s = (const uint2*)(rows200 + (size_t)i * 200); uint2* d = (uint2*)(xy192 + (size_t)i * 192);
#pragma unroll
  for (int k = 0; k < 24; ++k) d[k] = s[k];
  const uint8_t f = (uint8_t)(s[24].x & 0xffu) ? 1 : 0;
  flags[i] = f;
  if (f) atomicAdd(n_inf, 1u);
}
int32_t g2_unpack200(Ctx* c, const void* d_rows200, void* d_xy192, void* d_flags, uint32_t* d_count, size_t n, hipStream_t s) {
  (void)c;
  HIPCHK(hipMemsetAsync(d_count, 0, 4, s));
  if (n) hipLaunchKernelGGL(k_g2_unpack200, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const char*)d_rows200, (char*)d_xy192, (uint8_t*)d_flags, d_count, (uint32_t)n);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}
struct G2Affine;
__device__ __noinline__ void g2_madd_ni(struct XYZZ2* pa, const G2Affine* pp);
__device__ __noinline__ void g2_slice_slow_path(const char* bases192, const uint32_t* run, uint32_t j, uint32_t j1, char* slot);

// ---- kernels (same bookkeeping as msm.hip, one lane per group operation) ----------------------------------------------------
__global__ void __launch_bounds__(256) k_g2_accum(const char* __restrict__ bases, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ hist,
                                                  const uint2* __restrict__ scan_local, const uint2* __restrict__ scan_blk, const uint32_t* __restrict__ total_pairs, uint32_t M,
                                                  const uint32_t* __restrict__ meta, const uint32_t* __restrict__ order, const uint32_t* __restrict__ task_g, char* __restrict__ partial) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= meta[0]) return;
  const uint32_t sid = order[t], g = task_g[sid];
  uint2 st = scan_at(scan_local, scan_blk, g);
  uint32_t cnt = hist[g], m = slices_of(cnt, pick_rule(total_pairs, M)), k = sid - st.y;
  uint32_t j0 = (uint32_t)(((uint64_t)k * cnt) / m), j1 = (uint32_t)(((uint64_t)(k + 1) * cnt) / m);
  const uint32_t* run = sorted + st.x;
  XYZZ2 acc = g2_infinity();
  for (uint32_t j = j0; j < j1; ++j) {
    const uint32_t e = run[j];
    G2Affine p; const char* row = bases + (size_t)(e & 0x7fffffffu) * 192;
    p.x = load_fq2(row); p.y = load_fq2(row + 96);
    if (e >> 31) { p.y.a = Fq::sub<1>(Fq::zero(), p.y.a); p.y.b = Fq::sub<1>(Fq::zero(), p.y.b); }      // q - y per component (canonical inputs)
    g2_madd_ni(&acc, &p);
  }
  store_xyzz2(partial + (size_t)sid * 384, acc);
}
// the rest of a slice whose fast loop met P == +-acc: the general 32-bit code, continuing from the sum the pair stored in the slice's slot
__device__ __noinline__ void g2_slice_slow_path(const char* bases192, const uint32_t* run, uint32_t j, uint32_t j1, char* slot) {
  XYZZ2 acc = load_xyzz2(slot);
  for (; j < j1; ++j) {
    const uint32_t e = run[j];
    G2Affine p; const char* row = bases192 + (size_t)(e & 0x7fffffffu) * 192;
    p.x = load_fq2(row); p.y = load_fq2(row + 96);
    if (e >> 31) { p.y.a = Fq::sub<1>(Fq::zero(), p.y.a); p.y.b = Fq::sub<1>(Fq::zero(), p.y.b); }
    g2_madd_ni(&acc, &p);
  }
  store_xyzz2(slot, acc);
}
// One lane pair per slice on the 28-bit form (rows28: k_g2_rows_to28); the slice sum leaves as a 32-bit XYZZ2 point (384 B: the tree and the
// reduction kernels below work on those), every lane converting and storing its own components.
__global__ void __launch_bounds__(256, 2) k_g2_accum28(const char* __restrict__ rows28, const char* __restrict__ bases192, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ hist, const uint2* __restrict__ scan_local, const uint2* __restrict__ scan_blk,
                                                    const uint32_t* __restrict__ total_pairs, uint32_t M, const uint32_t* __restrict__ meta, const uint32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ task_g, char* __restrict__ partial) {
  const uint32_t t = (blockIdx.x * 256 + threadIdx.x) >> 1; const bool odd = threadIdx.x & 1;
  if (t >= meta[0]) return;
  const uint32_t sid = order[t], g = task_g[sid];
  const uint2 st = scan_at(scan_local, scan_blk, g);
  const uint32_t cnt = hist[g], m = slices_of(cnt, pick_rule(total_pairs, M)), k = sid - st.y;
  const uint32_t j0 = (uint32_t)(((uint64_t)k * cnt) / m), j1 = (uint32_t)(((uint64_t)(k + 1) * cnt) / m);
  const uint32_t* run = sorted + st.x;
  char* slot = partial + (size_t)sid * 384;
  auto fetch = [&](uint32_t e, F28& x, F28& y) {
    load_affine28(rows28 + (size_t)(e & 0x7fffffffu) * 224 + (odd ? 112 : 0), x, y);
    if (e >> 31) y = f28_normalise(f28_sub<2, 1>(f28_const(Limbs14{}), y));      // 2q - y (canonical rows): exact digits, <= 2q
  };
  XYZZ2P acc; uint32_t j = j0; bool ok = true;
  {
    F28 x, y; fetch(run[j], x, y);
    acc.X = x; acc.Y = y; acc.ZZ = odd ? f28_const(Limbs14{}) : f28_const(ONE28); acc.ZZZ = acc.ZZ;      // (x, +-y, 1, 1): the one of Fq2 is (1, 0)
    ++j;
  }
  for (; j < j1; ++j) {
    F28 x, y; fetch(run[j], x, y);
    if (!g2p_madd_fast(acc, x, y, odd)) { ok = false; break; }
  }
  // my components as lazily reduced 32-bit Montgomery values (< 2q): X.a at 0, X.b at 48, Y at 96 / 144, ZZ at 192 / 240, ZZZ at 288 / 336
  const uint32_t o = odd ? 48u : 0u;
  store_fp<Fq>(slot + o, f28_to_fq(acc.X)); store_fp<Fq>(slot + 96 + o, f28_to_fq(acc.Y));
  store_fp<Fq>(slot + 192 + o, f28_to_fq(acc.ZZ)); store_fp<Fq>(slot + 288 + o, f28_to_fq(acc.ZZZ));
  if (!ok) {                                               // rare: repeated or opposite bases in one bucket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // both lanes' halves of the slot are in memory (one wave: its accesses to an address stay in order)
    if (!odd) g2_slice_slow_path(bases192, run, j, j1, slot);
  }
}
// partial[ft + i] += partial[ft + i + half] inside every multi-slice bucket of the list
__global__ void __launch_bounds__(256) k_g2_tree_pass(char* __restrict__ partial, const uint32_t* __restrict__ list, const uint2* __restrict__ scan_local,
                                                      const uint2* __restrict__ scan_blk, uint32_t M, const uint32_t* __restrict__ meta, uint32_t pass, uint32_t max_pairs,
                                                      uint32_t list_len) {
  const uint32_t op = blockIdx.x * 256 + threadIdx.x;
  uint32_t h = op / max_pairs, i = op % max_pairs;
  if (h >= list_len) return;
  uint32_t g = list[h];
  uint32_t ft = scan_at(scan_local, scan_blk, g).y;
  uint32_t fn = (g + 1 < M) ? scan_at(scan_local, scan_blk, g + 1).y : meta[0];
  uint32_t L = fn - ft;
  for (uint32_t p = 0; p < pass; ++p) L = (L + 1) >> 1;
  if (L <= 1) return;
  uint32_t half = (L + 1) >> 1;
  if (i >= L - half) return;
  char* pa = partial + (size_t)(ft + i) * 384;
  XYZZ2 a = load_xyzz2(pa), b = load_xyzz2(pa + (size_t)half * 384);
  g2_add_ni(&a, &b);
  store_xyzz2(pa, a);
}
// one lane per chunk of S consecutive buckets of one window: V = sum_{b in chunk} (b + 1) * S_b (running sums, then the chunk base by double-and-add)
__global__ void __launch_bounds__(256) k_g2_bucket_chunks(const char* __restrict__ partial, const uint32_t* __restrict__ hist, const uint2* __restrict__ scan_local,
                                                          const uint2* __restrict__ scan_blk, uint32_t B, uint32_t S, uint32_t nchunks_total, char* __restrict__ V) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nchunks_total) return;
  uint32_t cpw = B / S, w = t / cpw, j = t % cpw, g0 = w * B + j * S;
  XYZZ2 run = g2_infinity(), acc = g2_infinity();
  for (uint32_t k = 0; k < S; ++k) {
    const uint32_t g = g0 + (S - 1 - k);
    if (hist[g]) { XYZZ2 y = load_xyzz2(partial + (size_t)scan_at(scan_local, scan_blk, g).y * 384); g2_add_ni(&run, &y); }
    g2_add_ni(&acc, &run);
  }
  const uint32_t base = j * S;
  if (base) {
    XYZZ2 r = g2_infinity();
    for (int bit = 31 - __clz(base); bit >= 0; --bit) { g2_double_ni(&r); if ((base >> bit) & 1) g2_add_ni(&r, &run); }
    g2_add_ni(&acc, &r);
  }
  store_xyzz2(V + (size_t)t * 384, acc);
}
// V[seg*seg_len + i] += V[seg*seg_len + i + half] for i < L - half
__global__ void __launch_bounds__(256) k_g2_seg_tree_pass(char* __restrict__ V, uint32_t seg_len, uint32_t nseg, uint32_t L) {
  const uint32_t half = (L + 1) >> 1, pairs = L - half;
  const uint32_t op = blockIdx.x * 256 + threadIdx.x;
  if (op >= pairs * nseg) return;
  const uint32_t seg = op / pairs, i = op % pairs;
  char* pa = V + ((size_t)seg * seg_len + i) * 384;
  XYZZ2 a = load_xyzz2(pa), b = load_xyzz2(pa + (size_t)half * 384);
  g2_add_ni(&a, &b);
  store_xyzz2(pa, a);
}
__global__ void k_g2_gather_windows(const char* __restrict__ V, uint32_t seg_len, uint32_t W, char* __restrict__ out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= W * 24) return;
  uint32_t w = t / 24, q = t % 24;
  ((uint4*)out)[t] = ((const uint4*)(V + (size_t)w * seg_len * 384))[q];
}

// ---- host tail: Fq2 and G2 on the host -----------------------------------------------------------------------------------------
namespace host {
struct HFq2 {
  HFq a, b;
  static HFq2 zero() { HFq2 r; r.a = HFq::zero(); r.b = HFq::zero(); return r; }
  static HFq2 one() { HFq2 r; r.a = HFq::one(); r.b = HFq::zero(); return r; }
  bool is_zero() const { return a.is_zero() && b.is_zero(); }
  static HFq2 add(const HFq2& x, const HFq2& y) { HFq2 r; r.a = HFq::add(x.a, y.a); r.b = HFq::add(x.b, y.b); return r; }
  static HFq2 sub(const HFq2& x, const HFq2& y) { HFq2 r; r.a = HFq::sub(x.a, y.a); r.b = HFq::sub(x.b, y.b); return r; }
  static HFq2 dbl(const HFq2& x) { return add(x, x); }
  static HFq times5(const HFq& v) { HFq d = HFq::dbl(HFq::dbl(v)); return HFq::add(d, v); }
  static HFq2 mul(const HFq2& x, const HFq2& y) {
    HFq v0 = HFq::mul(x.a, y.a), v1 = HFq::mul(x.b, y.b);
    HFq2 r; r.a = HFq::sub(v0, times5(v1)); r.b = HFq::sub(HFq::sub(HFq::mul(HFq::add(x.a, x.b), HFq::add(y.a, y.b)), v0), v1); return r;
  }
  static HFq2 sqr(const HFq2& x) { return mul(x, x); }
  static HFq2 inv(const HFq2& x) {                     // (a - b u) / (a^2 + 5 b^2)
    HFq n = HFq::inv(HFq::add(HFq::sqr(x.a), times5(HFq::sqr(x.b))));
    HFq2 r; r.a = HFq::mul(x.a, n); r.b = HFq::neg(HFq::mul(x.b, n)); return r;
  }
};
struct HXYZZ2 {
  HFq2 X, Y, ZZ, ZZZ;
  static HXYZZ2 infinity() { HXYZZ2 r; r.X = HFq2::zero(); r.Y = r.X; r.ZZ = r.X; r.ZZZ = r.X; return r; }
  bool is_inf() const { return ZZ.is_zero(); }
};
static HXYZZ2 h2double(const HXYZZ2& p) {
  if (p.is_inf()) return p;
  HFq2 U = HFq2::dbl(p.Y), V = HFq2::sqr(U), W = HFq2::mul(U, V), S = HFq2::mul(p.X, V);
  HFq2 xx = HFq2::sqr(p.X), M = HFq2::add(HFq2::dbl(xx), xx);
  HXYZZ2 r;
  r.X = HFq2::sub(HFq2::sqr(M), HFq2::dbl(S));
  r.Y = HFq2::sub(HFq2::mul(M, HFq2::sub(S, r.X)), HFq2::mul(W, p.Y));
  r.ZZ = HFq2::mul(V, p.ZZ); r.ZZZ = HFq2::mul(W, p.ZZZ);
  if (r.ZZ.is_zero()) return HXYZZ2::infinity();
  return r;
}
static HXYZZ2 h2add(const HXYZZ2& a, const HXYZZ2& b) {
  if (a.is_inf()) return b;
  if (b.is_inf()) return a;
  HFq2 U1 = HFq2::mul(a.X, b.ZZ), U2 = HFq2::mul(b.X, a.ZZ), S1 = HFq2::mul(a.Y, b.ZZZ), S2 = HFq2::mul(b.Y, a.ZZZ);
  HFq2 P = HFq2::sub(U2, U1), R = HFq2::sub(S2, S1);
  if (P.is_zero()) { if (R.is_zero()) return h2double(a); return HXYZZ2::infinity(); }
  HFq2 PP = HFq2::sqr(P), PPP = HFq2::mul(P, PP), Q = HFq2::mul(U1, PP);
  HXYZZ2 r;
  r.X = HFq2::sub(HFq2::sub(HFq2::sqr(R), PPP), HFq2::dbl(Q));
  r.Y = HFq2::sub(HFq2::mul(R, HFq2::sub(Q, r.X)), HFq2::mul(S1, PPP));
  r.ZZ = HFq2::mul(HFq2::mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = HFq2::mul(HFq2::mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}
static HFq2 h2load(const uint64_t* p) { HFq2 r; std::memcpy(r.a.l, p, 48); std::memcpy(r.b.l, p + 6, 48); return r; }
static void h2store(uint64_t* p, const HFq2& x) { std::memcpy(p, x.a.l, 48); std::memcpy(p + 6, x.b.l, 48); }
static HXYZZ2 h2from_jacobian(const uint64_t* j36) {
  HXYZZ2 r; r.X = h2load(j36); r.Y = h2load(j36 + 12); HFq2 Z = h2load(j36 + 24);
  if (Z.is_zero()) return HXYZZ2::infinity();
  r.ZZ = HFq2::sqr(Z); r.ZZZ = HFq2::mul(r.ZZ, Z); return r;
}
// affine-normalised Jacobian (x, y, 1); the identity as snarkVM's Projective::zero() = (1, 1, 0)
static void h2store_jacobian_normalized(uint64_t* j36, const HXYZZ2& p) {
  const HFq2 one = HFq2::one();
  if (p.is_inf()) { h2store(j36, one); h2store(j36 + 12, one); h2store(j36 + 24, HFq2::zero()); return; }
  HFq2 zi3 = HFq2::inv(p.ZZZ), zi2 = HFq2::sqr(HFq2::mul(zi3, p.ZZ));
  h2store(j36, HFq2::mul(p.X, zi2)); h2store(j36 + 12, HFq2::mul(p.Y, zi3)); h2store(j36 + 24, one);
}
}  // namespace host

// One G2 MSM: bases d_xy (n x 192 B, device) with optional infinity flags, scalars on the device.  Plain schedule (no table).
int32_t msm_g2_run(Ctx* c, uint64_t* out_jac36, const void* d_xy, const uint8_t* d_inf, const void* d_scalars, size_t n, hipStream_t s) {
  using namespace host;
  if (n == 0) { h2store_jacobian_normalized(out_jac36, HXYZZ2::infinity()); return ALEO_MI355X_OK; }
  if (n >= (1ull << 31)) { g_last_error = "msm_g2: n exceeds 2^31"; return ALEO_MI355X_ERR_BAD_ARG; }
  MsmPlan P = make_plan(n, 0);
  SegArgs segs{}; segs.nseg = 1; segs.ptr[0] = (const char*)d_scalars; segs.n[0] = (uint32_t)n;
  int32_t rc;
  if ((rc = ensure_host_pinned(c, 64 + (size_t)P.W * 384))) return rc;
  SortPhase sp;
  if ((rc = msm_sort_phase(c, segs, n, false, d_inf, (uint32_t)n, P, false, s, &sp))) return rc;
  const uint32_t M = sp.M, cpw = P.B / P.S, nchunks = cpw * P.W;
  if ((rc = c->partial.reserve(sp.slices_max * 384))) return rc;
  if ((rc = c->vbuf.reserve(((size_t)nchunks + P.W) * 384))) return rc;
  char* partial = c->partial.as<char>(); char* V = c->vbuf.as<char>(); char* Vout = V + (size_t)nchunks * 384;
  static const bool pair28 = [] { const char* e = std::getenv("ALEO_MI355X_G2_PAIR28"); return !(e && e[0] == '0'); }();      // A/B switch: 0 = the round-2 kernel (32-bit limbs, one lane per slice, out-of-line field calls)
  if (pair28) {
    if ((rc = c->out_stage.reserve(n * 224))) return rc;                          // the bases in the 28-bit form (per call: a G2 MSM keeps nothing resident)
    hipLaunchKernelGGL(k_g2_rows_to28, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const char*)d_xy, c->out_stage.as<char>(), (uint32_t)n);
    hipLaunchKernelGGL(k_g2_accum28, dim3(2 * sp.slice_blocks), dim3(256), 0, s, c->out_stage.as<const char>(), (const char*)d_xy, sp.sorted, sp.hist, sp.scan_local, sp.scan_blk,
                       sp.total_pairs, M, sp.meta, sp.order, sp.task_g, partial);
  } else
  hipLaunchKernelGGL(k_g2_accum, dim3(sp.slice_blocks), dim3(256), 0, s, (const char*)d_xy, sp.sorted, sp.hist, sp.scan_local, sp.scan_blk, sp.total_pairs, M, sp.meta,
                     sp.order, sp.task_g, partial);
  HIPCHK(hipGetLastError());
  SliceMeta sm;
  if ((rc = msm_wait_meta(c, sp, s, &sm))) return rc;
  for (uint32_t pass = 0, L = sm.max_m; L > 1; ++pass, L = (L + 1) >> 1) {
    const uint32_t Lc = sm.super_overflow ? L : (L < 16u ? L : (16u >> (pass < 4 ? pass : 4)));
    if (sm.n_heavy && Lc > 1) {
      uint32_t mp = Lc >> 1; uint64_t threads = (uint64_t)sm.n_heavy * mp;
      hipLaunchKernelGGL(k_g2_tree_pass, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, partial, sp.heavy, sp.scan_local, sp.scan_blk, M, sp.meta, pass, mp, sm.n_heavy);
    }
    if (sm.n_super) {
      uint32_t mp = L >> 1; uint64_t threads = (uint64_t)sm.n_super * mp;
      hipLaunchKernelGGL(k_g2_tree_pass, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, partial, sp.super_list, sp.scan_local, sp.scan_blk, M, sp.meta, pass, mp, sm.n_super);
    }
  }
  hipLaunchKernelGGL(k_g2_bucket_chunks, dim3((nchunks + 255) / 256), dim3(256), 0, s, partial, sp.hist, sp.scan_local, sp.scan_blk, P.B, P.S, nchunks, V);
  for (uint32_t L = cpw; L > 1; L = (L + 1) >> 1) {
    uint32_t pairs = (L - ((L + 1) >> 1)) * P.W;
    hipLaunchKernelGGL(k_g2_seg_tree_pass, dim3((pairs + 255) / 256), dim3(256), 0, s, V, cpw, P.W, L);
  }
  hipLaunchKernelGGL(k_g2_gather_windows, dim3((P.W * 24 + 255) / 256), dim3(256), 0, s, V, cpw, P.W, Vout);
  char* h_win = (char*)c->h_pinned + 64;
  HIPCHK(hipMemcpyAsync(h_win, Vout, (size_t)P.W * 384, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  // host tail: total = sum_w 2^(c w) * S_w (Horner from the top window); device coordinates are lazily reduced (< 2q)
  auto lazy2 = [](const uint64_t* p) { HFq2 r; r.a = HFq::reduce_lazy(p); r.b = HFq::reduce_lazy(p + 6); return r; };
  HXYZZ2 total = HXYZZ2::infinity();
  for (int w = (int)P.W - 1; w >= 0; --w) {
    for (int d = 0; d < plan_win_width((int)P.c, w); ++d) total = h2double(total);
    const uint64_t* src = (const uint64_t*)(h_win + (size_t)w * 384);
    HXYZZ2 v; v.X = lazy2(src); v.Y = lazy2(src + 12); v.ZZ = lazy2(src + 24); v.ZZZ = lazy2(src + 36);
    total = h2add(total, v);
  }
  h2store_jacobian_normalized(out_jac36, total);
  return ALEO_MI355X_OK;
}

}  // namespace aleo_mi355x

using namespace aleo_mi355x;

// Slot acquisition lives in api.hip; these two entry points are defined there around msm_g2_run / the host group law:
namespace aleo_mi355x {
int32_t g2_sum_host(uint64_t* out36, const uint64_t* pts36, size_t count) {
  host::HXYZZ2 t = host::HXYZZ2::infinity();
  for (size_t i = 0; i < count; ++i) t = host::h2add(t, host::h2from_jacobian(pts36 + 36 * i));
  host::h2store_jacobian_normalized(out36, t);
  return ALEO_MI355X_OK;
}
}  // namespace aleo_mi355x
