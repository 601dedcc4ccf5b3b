// host_field.hpp — host-side (CPU) BLS12-377 arithmetic used by the PRODUCT library for the O(W) tail work
// that must not run on a single GPU lane: the final Horner combination of the window sums, affine
// normalisation of the result, partial-sum combination across GPUs, and twiddle/table generation.
// (A dependent chain of ~250 doublings costs ~2 ms on one GPU lane and ~0.1 ms on a host core.)
//
// This is independent of oracle/ (which is test infrastructure): nothing here includes or links it.
// Layouts match snarkVM 0.14.5 (fields/src/fp_256.rs, fp_384.rs; curves/.../short_weierstrass_jacobian):
// u64 limbs little-endian, Montgomery form with R = 2^(64*N).
#pragma once
#include <cstdint>
#include <cstring>
#include <immintrin.h>
#include "host_modinv.hpp"

namespace aleo_mi355x { namespace host {

typedef unsigned __int128 u128;

template <int N> struct HParams;
template <> struct HParams<4> {  // Fr
  static constexpr uint64_t P[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};
  static constexpr uint64_t ONE[4] = {0x7d1c7ffffffffff3ULL, 0x7257f50f6ffffff2ULL, 0x16d81575512c0feeULL, 0x0d4bda322bbb9a9dULL};
  static constexpr uint64_t R2[4] = {0x25d577bab861857bULL, 0xcc2c27b58860591fULL, 0xa7cc008fe5dc8593ULL, 0x011fdae7eff1c939ULL};
  static constexpr uint64_t INV = 0x0a117fffffffffffULL;
};
template <> struct HParams<6> {  // Fq
  static constexpr uint64_t P[6] = {0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL, 0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL};
  static constexpr uint64_t ONE[6] = {0x02cdffffffffff68ULL, 0x51409f837fffffb1ULL, 0x9f7db3a98a7d3ff2ULL, 0x7b4e97b76e7c6305ULL, 0x4cf495bf803c84e8ULL, 0x008d6661e2fdf49aULL};
  static constexpr uint64_t R2[6] = {0xb786686c9400cd22ULL, 0x0329fcaab00431b1ULL, 0x22a5f11162d6b46dULL, 0xbfdf7d03827dc3acULL, 0x837e92f041790bf9ULL, 0x006dfccb1e914b88ULL};
  static constexpr uint64_t INV = 0x8508bfffffffffffULL;
};

// Fully reduced Montgomery field element on the host.
template <int N> struct Wide;
template <int N> struct HFp {
  uint64_t l[N];
  using Pm = HParams<N>;

  static HFp zero() { HFp r; std::memset(r.l, 0, sizeof r.l); return r; }
  static HFp one() { HFp r; std::memcpy(r.l, Pm::ONE, sizeof r.l); return r; }
  static HFp from_u64(uint64_t v) { HFp t = zero(); t.l[0] = v; HFp r2; std::memcpy(r2.l, Pm::R2, sizeof r2.l); return mul(t, r2); }
  bool is_zero() const { uint64_t o = 0; for (int i = 0; i < N; ++i) o |= l[i]; return o == 0; }
  bool operator==(const HFp& b) const { uint64_t o = 0; for (int i = 0; i < N; ++i) o |= l[i] ^ b.l[i]; return o == 0; }

  static bool geq_p(const uint64_t* a) {
    for (int i = N - 1; i >= 0; --i) { if (a[i] > Pm::P[i]) return true; if (a[i] < Pm::P[i]) return false; }
    return true;
  }
  static void sub_p(uint64_t* a) {
    uint64_t br = 0;
    for (int i = 0; i < N; ++i) { u128 t = (u128)a[i] - Pm::P[i] - br; a[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; }
  }
  // Reduce an arbitrary N-limb value (e.g. a lazily reduced device result < 16p) to canonical form.
  static HFp reduce_lazy(const uint64_t* a) {
    HFp r; std::memcpy(r.l, a, sizeof r.l);
    while (geq_p(r.l)) sub_p(r.l);
    return r;
  }
  static HFp add(const HFp& a, const HFp& b) {
    HFp r; uint64_t c = 0;
    for (int i = 0; i < N; ++i) { u128 s = (u128)a.l[i] + b.l[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    if (c || geq_p(r.l)) sub_p(r.l);
    return r;
  }
  static HFp sub(const HFp& a, const HFp& b) {
    HFp r; uint64_t br = 0;
    for (int i = 0; i < N; ++i) { u128 s = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)s; br = (uint64_t)(s >> 64) & 1; }
    if (br) { uint64_t c = 0; for (int i = 0; i < N; ++i) { u128 s = (u128)r.l[i] + Pm::P[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
    return r;
  }
  static HFp dbl(const HFp& a) { return add(a, a); }
  static HFp neg(const HFp& a) { return a.is_zero() ? a : sub(zero(), a); }
  // Montgomery product: separated operand scanning on mulx / adc chains (Wide<N> below) — ~35 ns for Fq on a 2.1 GHz core against ~60 ns for the
  // interleaved unsigned __int128 form this file used before: the host tails of the MSMs and the transcript are chains of these
  static inline HFp mul(const HFp& a, const HFp& b);
  static inline HFp sqr(const HFp& a);                       // dedicated squaring (Wide<N>::square)
  static HFp pow(const HFp& a, const uint64_t* e, int nl) {
    HFp acc = one();
    for (int i = nl * 64 - 1; i >= 0; --i) { acc = sqr(acc); if ((e[i / 64] >> (i % 64)) & 1) acc = mul(acc, a); }
    return acc;
  }
  static HFp pow_u64(const HFp& a, uint64_t e) { return pow(a, &e, 1); }
  static HFp inv_fermat(const HFp& a) { uint64_t e[N]; std::memcpy(e, Pm::P, sizeof e); e[0] -= 2; return pow(a, e, N); }      // a^(p-2): ~570 products for Fq (the reference the fast one is checked against)
  // Bernstein-Yang divsteps on the Montgomery representative x = a R: x^-1 = a^-1 R^-1 as a plain integer, times R^3 under one Montgomery product = a^-1 R.  0 -> 0.
  static HFp inv(const HFp& a) {
    static const HFp r3 = [] { HFp r2; std::memcpy(r2.l, Pm::R2, sizeof r2.l); return mul(r2, r2); }();
    HFp y; ModInv<N>::inverse(y.l, a.l, Pm::P);
    return mul(y, r3);
  }
  static HFp to_mont(const HFp& a) { HFp r2; std::memcpy(r2.l, Pm::R2, sizeof r2.l); return mul(a, r2); }
  static HFp from_mont(const HFp& a) { HFp o = zero(); o.l[0] = 1; return mul(a, o); }
};

// ---- wide products: a dot product of W pairs costs W half-products and ONE Montgomery reduction --------------------------------------------
// Separated operand scanning on mulx / adc chains (x86-64 BMI2 + ADX: every server CPU since 2015; build.sh passes -mbmi2 -madx to the host
// pass): a row of N mulx, one carry chain for the low halves, one for the high halves.  ~35 ns per Fq product on a 2.1 GHz core against
// ~60 ns for the generic unsigned __int128 form of host_field.hpp (which stays for everything that is not a hot chain).
typedef unsigned long long ull;
template <int N> struct Wide {
  ull t[2 * N + 1];
  __attribute__((always_inline)) static inline void product(ull* __restrict o, const ull* __restrict a, const ull* __restrict b) {      // o[0..2N) = a * b
    ull lo[N], hi[N]; unsigned char c = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) lo[j] = _mulx_u64(a[j], b[0], &hi[j]);
    o[0] = lo[0];
#pragma unroll
    for (int j = 1; j < N; ++j) c = _addcarry_u64(c, lo[j], hi[j - 1], &o[j]);
    _addcarry_u64(c, hi[N - 1], 0, &o[N]);
#pragma unroll
    for (int i = 1; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < N; ++j) lo[j] = _mulx_u64(a[j], b[i], &hi[j]);
      c = 0;
#pragma unroll
      for (int j = 0; j < N; ++j) c = _addcarry_u64(c, o[i + j], lo[j], &o[i + j]);
      ull top; _addcarry_u64(c, 0, 0, &top);
      c = 0;
#pragma unroll
      for (int j = 0; j < N - 1; ++j) c = _addcarry_u64(c, o[i + j + 1], hi[j], &o[i + j + 1]);
      _addcarry_u64(c, top, hi[N - 1], &o[i + N]);                                                         // a row's product has no carry beyond limb i + N
    }
  }
  // o[0..2N) = a^2: the N (N - 1) / 2 cross products once, doubled, plus the N squares — 21 + 6 multiplications for Fq instead of 36 (the Montgomery
  // reduction behind it stays 36): the s-boxes of the transcript's Poseidon (x^17 = four squarings and a product) and the point doublings of the host tails
  __attribute__((always_inline)) static inline void square(ull* __restrict o, const ull* __restrict a) {
    ull c[2 * N];                                            // cross products sum_{i < j} a_i a_j 2^(64 (i + j)), limbs 1 .. 2N - 2
#pragma unroll
    for (int k = 0; k < 2 * N; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
      ull lo[N], hi[N]; unsigned char cy = 0;
#pragma unroll
      for (int j = i + 1; j < N; ++j) lo[j] = _mulx_u64(a[i], a[j], &hi[j]);
#pragma unroll
      for (int j = i + 1; j < N; ++j) cy = _addcarry_u64(cy, c[i + j], lo[j], &c[i + j]);
      ull top; _addcarry_u64(cy, 0, 0, &top);
      cy = 0;
#pragma unroll
      for (int j = i + 1; j < N - 1; ++j) cy = _addcarry_u64(cy, c[i + j + 1], hi[j], &c[i + j + 1]);
      _addcarry_u64(cy, top, hi[N - 1], &c[i + N]);          // c[i + N] is still zero here: the row's top limb
    }
    ull prev = 0;                                            // double
#pragma unroll
    for (int k = 1; k < 2 * N; ++k) { const ull v = c[k]; c[k] = (v << 1) | (prev >> 63); prev = v; }
    unsigned char cy = 0;                                    // + the squares
#pragma unroll
    for (int i = 0; i < N; ++i) {
      ull h; const ull l = _mulx_u64(a[i], a[i], &h);
      cy = _addcarry_u64(cy, c[2 * i], l, &o[2 * i]);
      cy = _addcarry_u64(cy, c[2 * i + 1], h, &o[2 * i + 1]);
    }
  }
  __attribute__((always_inline)) inline void set_sqr(const uint64_t* a) { square(t, (const ull*)a); t[2 * N] = 0; }
  __attribute__((always_inline)) inline void set_mul(const uint64_t* a, const uint64_t* b) { product(t, (const ull*)a, (const ull*)b); t[2 * N] = 0; }
  __attribute__((always_inline)) inline void add_mul(const uint64_t* a, const uint64_t* b) {
    ull u[2 * N]; product(u, (const ull*)a, (const ull*)b);
    unsigned char c = 0;
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) c = _addcarry_u64(c, t[i], u[i], &t[i]);
    t[2 * N] += c;
  }
  // Montgomery reduction of a value < W p^2 with W p < R (R/q = 152, R/r = 13.7; W <= 9): result < 2p before the final subtraction
  __attribute__((always_inline)) inline HFp<N> redc() {
    using Pm = HParams<N>;
    const ull* P = (const ull*)Pm::P;
    ull pending = 0;                                         // carries out of the previous row, due at limb i + N
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const ull m = t[i] * Pm::INV; ull lo[N], hi[N]; unsigned char c = 0;
#pragma unroll
      for (int j = 0; j < N; ++j) lo[j] = _mulx_u64(m, P[j], &hi[j]);
#pragma unroll
      for (int j = 0; j < N; ++j) c = _addcarry_u64(c, t[i + j], lo[j], &t[i + j]);
      const unsigned char k1 = _addcarry_u64(c, t[i + N], pending, &t[i + N]);
      c = 0;
#pragma unroll
      for (int j = 0; j < N; ++j) c = _addcarry_u64(c, t[i + j + 1], hi[j], &t[i + j + 1]);
      pending = (ull)k1 + c;
    }
    t[2 * N] += pending;
    HFp<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = t[N + i];
    if (t[2 * N] || HFp<N>::geq_p(r.l)) HFp<N>::sub_p(r.l);
    return r;
  }
};
template <int N> __attribute__((always_inline)) inline HFp<N> fmul(const HFp<N>& a, const HFp<N>& b) { Wide<N> w; w.set_mul(a.l, b.l); return w.redc(); }
template <int N> __attribute__((always_inline)) inline HFp<N> fsqr(const HFp<N>& a) { Wide<N> w; w.set_sqr(a.l); return w.redc(); }
template <int N> inline HFp<N> HFp<N>::mul(const HFp<N>& a, const HFp<N>& b) { return fmul<N>(a, b); }
template <int N> inline HFp<N> HFp<N>::sqr(const HFp<N>& a) { return fsqr<N>(a); }

using HFr = HFp<4>;
using HFq = HFp<6>;

// Fr constants of the evaluation domain (snarkvm-curves bls12_377/fr.rs)
static constexpr uint64_t FR_TWO_ADIC_ROOT_CANON[4] = {0x476ef4a4ec2a895eULL, 0x9b506ee363e3f04aULL, 0x60c69477d1a8a12fULL, 0x11d4b7f60cb92cc1ULL};
static constexpr int FR_TWO_ADICITY = 47;
static constexpr uint64_t FR_GENERATOR = 22;

// ---- G1 on the host: extended Jacobian (X, Y, ZZ, ZZZ), fully reduced coordinates -------------
struct HXYZZ {
  HFq X, Y, ZZ, ZZZ;
  static HXYZZ infinity() { HXYZZ r; r.X = HFq::zero(); r.Y = HFq::zero(); r.ZZ = HFq::zero(); r.ZZZ = HFq::zero(); return r; }
  bool is_inf() const { return ZZ.is_zero(); }
};

inline HXYZZ hdouble(const HXYZZ& p) {  // dbl-2008-s-1, a = 0
  if (p.is_inf()) return p;
  HXYZZ r;
  HFq U = HFq::dbl(p.Y), V = HFq::sqr(U), W = HFq::mul(U, V), S = HFq::mul(p.X, V);
  HFq xx = HFq::sqr(p.X), M = HFq::add(HFq::dbl(xx), xx);
  r.X = HFq::sub(HFq::sqr(M), HFq::dbl(S));
  r.Y = HFq::sub(HFq::mul(M, HFq::sub(S, r.X)), HFq::mul(W, p.Y));
  r.ZZ = HFq::mul(V, p.ZZ); r.ZZZ = HFq::mul(W, p.ZZZ);
  if (r.ZZ.is_zero()) return HXYZZ::infinity();
  return r;
}

inline HXYZZ hadd(const HXYZZ& a, const HXYZZ& b) {  // add-2008-s
  if (a.is_inf()) return b;
  if (b.is_inf()) return a;
  HFq U1 = HFq::mul(a.X, b.ZZ), U2 = HFq::mul(b.X, a.ZZ), S1 = HFq::mul(a.Y, b.ZZZ), S2 = HFq::mul(b.Y, a.ZZZ);
  HFq P = HFq::sub(U2, U1), R = HFq::sub(S2, S1);
  if (P.is_zero()) { if (R.is_zero()) return hdouble(a); return HXYZZ::infinity(); }
  HFq PP = HFq::sqr(P), PPP = HFq::mul(P, PP), Q = HFq::mul(U1, PP);
  HXYZZ r;
  r.X = HFq::sub(HFq::sub(HFq::sqr(R), PPP), HFq::dbl(Q));
  r.Y = HFq::sub(HFq::mul(R, HFq::sub(Q, r.X)), HFq::mul(S1, PPP));
  r.ZZ = HFq::mul(HFq::mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = HFq::mul(HFq::mul(a.ZZZ, b.ZZZ), PPP);
  return r;
}

// (x, y) affine of an XYZZ point; returns false for infinity
inline bool hto_affine(const HXYZZ& p, HFq& x, HFq& y) {
  if (p.is_inf()) return false;
  HFq zi3 = HFq::inv(p.ZZZ);                 // 1/Z^3
  HFq zi2 = HFq::sqr(HFq::mul(zi3, p.ZZ));    // (Z^2/Z^3)^2 = 1/Z^2
  x = HFq::mul(p.X, zi2); y = HFq::mul(p.Y, zi3);
  return true;
}

// Jacobian (X, Y, Z) as snarkVM's Projective stores it (144 bytes) <-> XYZZ
inline HXYZZ hfrom_jacobian(const uint64_t* j18) {
  HXYZZ r; HFq Z;
  std::memcpy(r.X.l, j18, 48); std::memcpy(r.Y.l, j18 + 6, 48); std::memcpy(Z.l, j18 + 12, 48);
  if (Z.is_zero()) return HXYZZ::infinity();
  r.ZZ = HFq::sqr(Z); r.ZZZ = HFq::mul(r.ZZ, Z);
  return r;
}
// Writes the affine-normalised point as Jacobian (x, y, 1); infinity as snarkVM's Projective::zero() = (1, 1, 0).
inline void hstore_jacobian_normalized(uint64_t* j18, const HXYZZ& p) {
  HFq x, y;
  if (!hto_affine(p, x, y)) {
    HFq o = HFq::one(); std::memcpy(j18, o.l, 48); std::memcpy(j18 + 6, o.l, 48); std::memset(j18 + 12, 0, 48); return;
  }
  HFq o = HFq::one();
  std::memcpy(j18, x.l, 48); std::memcpy(j18 + 6, y.l, 48); std::memcpy(j18 + 12, o.l, 48);
}

// Affine normalisation of `count` results with ONE field inversion (Montgomery's trick over the ZZZ coordinates): the tail
// of a batched call, where k inversions (~570 products each) would cost more than everything else the host does.
inline void hstore_jacobian_normalized_batch(uint64_t* j18, const HXYZZ* pts, size_t count) {
  if (count == 1) { hstore_jacobian_normalized(j18, pts[0]); return; }
  HFq acc = HFq::one(); HFq pre[64];
  for (size_t base = 0; base < count; base += 64) {
    const size_t m = count - base < 64 ? count - base : 64;
    acc = HFq::one();
    for (size_t i = 0; i < m; ++i) { pre[i] = acc; if (!pts[base + i].is_inf()) acc = HFq::mul(acc, pts[base + i].ZZZ); }
    HFq inv = HFq::inv(acc);
    for (size_t i = m; i-- > 0;) {
      const HXYZZ& p = pts[base + i]; uint64_t* o = j18 + 18 * (base + i);
      const HFq one = HFq::one();
      if (p.is_inf()) { std::memcpy(o, one.l, 48); std::memcpy(o + 6, one.l, 48); std::memset(o + 12, 0, 48); continue; }
      const HFq zi3 = HFq::mul(inv, pre[i]);                 // 1/ZZZ_i
      inv = HFq::mul(inv, p.ZZZ);
      const HFq zi2 = HFq::sqr(HFq::mul(zi3, p.ZZ));         // (ZZ/ZZZ)^2 = 1/ZZ
      const HFq x = HFq::mul(p.X, zi2), y = HFq::mul(p.Y, zi3);
      std::memcpy(o, x.l, 48); std::memcpy(o + 6, y.l, 48); std::memcpy(o + 12, one.l, 48);
    }
  }
}

}}  // namespace aleo_mi355x::host
