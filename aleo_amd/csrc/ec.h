// ec.h — BLS12-377 G1 group law on the device (y^2 = x^3 + 1 over Fq, a = 0).
//
// Replaces (on the device) snarkvm-curves 0.14.5 templates/short_weierstrass_jacobian/{affine,projective}.rs
// add_assign_mixed / add_assign / double_in_place  [UPSTREAM-RECALL; pin /root/reference/Cargo.lock:2637].
// The reference accumulates in Jacobian (X,Y,Z); the device accumulates in extended Jacobian "XYZZ"
// (X, Y, ZZ = Z^2, ZZZ = Z^3): x = X/ZZ, y = Y/ZZZ.  A mixed addition costs 8M+2S instead of 7M+4S and never
// needs Z itself.  Both describe the same group element; the C ABI hands back the affine-normalised point as
// Jacobian (x, y, 1) — SURVEY.md §0 fact 4: results are compared after normalisation, never as raw Jacobian.
//
// Lazy-reduction invariant of every stored/accumulated XYZZ point (see fp.h for the product bound):
//     X < 8q,  Y < 4q,  ZZ < 2q,  ZZZ < 2q        infinity <=> ZZ stored as raw 0
// Affine inputs are canonical (< q).  Each formula below carries its bound bookkeeping in comments.
#pragma once
#include "fp.h"

namespace aleo_mi355x {

struct AffinePt { Fq x, y; };                 // 96 bytes in memory, Montgomery form
struct XYZZ { Fq X, Y, ZZ, ZZZ; };            // 192 bytes in memory

__device__ __forceinline__ AffinePt load_affine(const void* p) {
  AffinePt r; r.x = load_fp<Fq>(p); r.y = load_fp<Fq>((const char*)p + 48); return r;
}
__device__ __forceinline__ XYZZ load_xyzz(const void* p) {
  XYZZ r; const char* c = (const char*)p;
  r.X = load_fp<Fq>(c); r.Y = load_fp<Fq>(c + 48); r.ZZ = load_fp<Fq>(c + 96); r.ZZZ = load_fp<Fq>(c + 144); return r;
}
__device__ __forceinline__ void store_xyzz(void* p, const XYZZ& a) {
  char* c = (char*)p;
  store_fp<Fq>(c, a.X); store_fp<Fq>(c + 48, a.Y); store_fp<Fq>(c + 96, a.ZZ); store_fp<Fq>(c + 144, a.ZZZ);
}
__device__ __forceinline__ XYZZ xyzz_infinity() { XYZZ r; r.X = Fq::zero(); r.Y = Fq::zero(); r.ZZ = Fq::zero(); r.ZZZ = Fq::zero(); return r; }
__device__ __forceinline__ bool xyzz_is_inf(const XYZZ& a) { return a.ZZ.is_zero_raw(); }

// y -> -y for a canonical y: q - y (y == 0 maps to q == 0 mod q; harmless under lazy reduction)
__device__ __forceinline__ Fq fq_neg_canonical(const Fq& y) { return Fq::sub<1>(Fq::zero(), y); }

// 2 * (x, y) for an affine point, result XYZZ (EFD mdbl-2008-s-1, a = 0).  x, y < 2q accepted.
__device__ __noinline__ void xyzz_double_affine(XYZZ* out, const Fq* px, const Fq* py) {
  const Fq x = *px, y = *py;
  XYZZ r;
  Fq U = Fq::dbl(y);                          // < 4q
  Fq V = Fq::sqr(U);                          // 16/152+1 -> < 2q
  Fq W = Fq::mul(U, V);                       // < 2q
  Fq S = Fq::mul(x, V);                       // < 2q
  Fq xx = Fq::sqr(x);                         // < 2q
  Fq M = Fq::add(Fq::dbl(xx), xx);            // 3x^2 < 6q
  Fq MM = Fq::sqr(M);                         // 36/152+1 -> < 2q
  r.X = Fq::sub<4>(MM, Fq::dbl(S));           // MM + 4q - 2S < 6q
  Fq t = Fq::sub<8>(S, r.X);                  // S + 8q - X3 < 10q
  Fq Mt = Fq::mul(M, t);                      // 60/152+1 -> < 2q
  Fq Wy = Fq::mul(W, y);                      // < 2q
  r.Y = Fq::sub<2>(Mt, Wy);                   // < 4q
  r.ZZ = V; r.ZZZ = W;
  if (r.ZZ.is_zero_mod_lt2p()) r = xyzz_infinity();   // y == 0: a 2-torsion point doubles to the identity
  *out = r;
}

// 2 * P for an XYZZ point (EFD dbl-2008-s-1, a = 0).
__device__ __noinline__ void xyzz_double_ni(XYZZ* io) {
  const XYZZ p = *io;
  if (xyzz_is_inf(p)) return;
  XYZZ r;
  Fq U = Fq::dbl(p.Y);                        // < 8q
  Fq V = Fq::sqr(U);                          // 64/152+1 -> < 2q
  Fq W = Fq::mul(U, V);                       // 16/152+1 -> < 2q
  Fq S = Fq::mul(p.X, V);                     // 16/152+1 -> < 2q
  Fq xx = Fq::sqr(p.X);                       // 64/152+1 -> < 2q
  Fq M = Fq::add(Fq::dbl(xx), xx);            // < 6q
  Fq MM = Fq::sqr(M);                         // < 2q
  r.X = Fq::sub<4>(MM, Fq::dbl(S));           // < 6q
  Fq t = Fq::sub<8>(S, r.X);                  // < 10q
  Fq Mt = Fq::mul(M, t);                      // < 2q
  Fq Wy = Fq::mul(W, p.Y);                    // 8/152+1 -> < 2q
  r.Y = Fq::sub<2>(Mt, Wy);                   // < 4q
  r.ZZ = Fq::mul(V, p.ZZ);                    // < 2q
  r.ZZZ = Fq::mul(W, p.ZZZ);                  // < 2q
  if (r.ZZ.is_zero_mod_lt2p()) r = xyzz_infinity();
  *io = r;
}

// acc += (x2, y2) affine, canonical coordinates (EFD madd-2008-s).  `acc_inf` is the caller-held infinity flag
// of acc (kept in a register so the hot loop never tests ZZ of a lazily reduced value for it).
__device__ __forceinline__ void xyzz_madd(XYZZ& acc, bool& acc_inf, const Fq& x2, const Fq& y2) {
  if (acc_inf) { acc.X = x2; acc.Y = y2; acc.ZZ = Fq::one(); acc.ZZZ = Fq::one(); acc_inf = false; return; }
  Fq U2 = Fq::mul(x2, acc.ZZ);                // < 2q
  Fq S2 = Fq::mul(y2, acc.ZZZ);               // < 2q
  Fq P = Fq::sub<8>(U2, acc.X);               // U2 + 8q - X1 < 10q
  Fq R = Fq::sub<4>(S2, acc.Y);               // S2 + 4q - Y1 < 6q
  Fq PP = Fq::sqr(P);                         // 100/152+1 -> < 2q
  Fq ZZ3 = Fq::mul(acc.ZZ, PP);               // < 2q
  if (__builtin_expect(ZZ3.is_zero_mod_lt2p(), 0)) {
    // P == 0 (mod q): same x.  Either the same point (double it) or the inverse (sum is infinity).
    if (R.is_zero_mod()) { XYZZ d; Fq tx = x2, ty = y2; xyzz_double_affine(&d, &tx, &ty); acc = d; acc_inf = xyzz_is_inf(d); }
    else { acc = xyzz_infinity(); acc_inf = true; }
    return;
  }
  Fq PPP = Fq::mul(P, PP);                    // 20/152+1 -> < 2q
  Fq Q = Fq::mul(acc.X, PP);                  // 16/152+1 -> < 2q
  Fq RR = Fq::sqr(R);                         // 36/152+1 -> < 2q
  Fq X3 = Fq::sub<4>(Fq::sub<2>(RR, PPP), Fq::dbl(Q));   // RR + 6q - PPP - 2Q < 8q
  Fq t = Fq::sub<8>(Q, X3);                   // Q + 8q - X3 < 10q
  Fq Rt = Fq::mul(R, t);                      // 60/152+1 -> < 2q
  Fq YP = Fq::mul(acc.Y, PPP);                // 8/152+1 -> < 2q
  acc.X = X3;
  acc.Y = Fq::sub<2>(Rt, YP);                 // < 4q
  acc.ZZ = ZZ3;
  acc.ZZZ = Fq::mul(acc.ZZZ, PPP);            // < 2q
}

// Hot-loop form of the mixed addition: acc is a finite point, no infinity flag, no doubling branch, nothing whose
// address is taken (a call in the rare branch made hipcc spill the point to scratch on EVERY iteration: 1.6 GB of
// scratch writes per 2^20 MSM, profiles/r01_pmc_*).  Returns false — leaving acc untouched — when the point has
// the same x as acc (P == +-acc); the caller then finishes its slice with the general code out of line.
__device__ __forceinline__ bool xyzz_madd_fast(XYZZ& acc, const Fq& x2, const Fq& y2) {
  Fq U2 = Fq::mul(x2, acc.ZZ);                // < 2q
  Fq S2 = Fq::mul(y2, acc.ZZZ);               // < 2q
  Fq P = Fq::sub<8>(U2, acc.X);               // < 10q
  Fq R = Fq::sub<4>(S2, acc.Y);               // < 6q
  Fq PP = Fq::sqr(P);                         // < 2q
  Fq ZZ3 = Fq::mul(acc.ZZ, PP);               // < 2q
  if (__builtin_expect(ZZ3.is_zero_mod_lt2p(), 0)) return false;
  Fq PPP = Fq::mul(P, PP);                    // < 2q
  Fq Q = Fq::mul(acc.X, PP);                  // < 2q
  Fq RR = Fq::sqr(R);                         // < 2q
  Fq X3 = Fq::sub<4>(Fq::sub<2>(RR, PPP), Fq::dbl(Q));   // < 8q
  Fq t = Fq::sub<8>(Q, X3);                   // < 10q
  Fq Rt = Fq::mul(R, t);                      // < 2q
  Fq YP = Fq::mul(acc.Y, PPP);                // < 2q
  acc.X = X3;
  acc.Y = Fq::sub<2>(Rt, YP);                 // < 4q
  acc.ZZ = ZZ3;
  acc.ZZZ = Fq::mul(acc.ZZZ, PPP);            // < 2q
  return true;
}

// acc += b, both XYZZ (EFD add-2008-s).  Infinity encoded as raw ZZ == 0 on both sides.
__device__ __forceinline__ void xyzz_add(XYZZ& acc, const XYZZ& b) {
  if (xyzz_is_inf(b)) return;
  if (xyzz_is_inf(acc)) { acc = b; return; }
  Fq U1 = Fq::mul(acc.X, b.ZZ);               // 16/152+1 -> < 2q
  Fq U2 = Fq::mul(b.X, acc.ZZ);               // < 2q
  Fq S1 = Fq::mul(acc.Y, b.ZZZ);              // 8/152+1 -> < 2q
  Fq S2 = Fq::mul(b.Y, acc.ZZZ);              // < 2q
  Fq P = Fq::sub<2>(U2, U1);                  // < 4q
  Fq R = Fq::sub<2>(S2, S1);                  // < 4q
  Fq PP = Fq::sqr(P);                         // 16/152+1 -> < 2q
  Fq ZZ3 = Fq::mul(Fq::mul(acc.ZZ, b.ZZ), PP);   // < 2q
  if (__builtin_expect(ZZ3.is_zero_mod_lt2p(), 0)) {
    if (R.is_zero_mod()) { XYZZ d = acc; xyzz_double_ni(&d); acc = d; } else acc = xyzz_infinity();
    return;
  }
  Fq PPP = Fq::mul(P, PP);                    // < 2q
  Fq Q = Fq::mul(U1, PP);                     // < 2q
  Fq RR = Fq::sqr(R);                         // 16/152+1 -> < 2q
  Fq X3 = Fq::sub<4>(Fq::sub<2>(RR, PPP), Fq::dbl(Q));   // < 8q
  Fq t = Fq::sub<8>(Q, X3);                   // < 10q
  Fq Rt = Fq::mul(R, t);                      // 40/152+1 -> < 2q
  Fq SP = Fq::mul(S1, PPP);                   // < 2q
  acc.X = X3;
  acc.Y = Fq::sub<2>(Rt, SP);                 // < 4q
  acc.ZZ = ZZ3;
  acc.ZZZ = Fq::mul(Fq::mul(acc.ZZZ, b.ZZZ), PPP);       // < 2q
}

// Out-of-line copy for call sites off the hot path (keeps kernels with several additions at one inlined body).
__device__ __noinline__ void xyzz_add_ni(XYZZ* a, const XYZZ* b) { XYZZ x = *a; const XYZZ y = *b; xyzz_add(x, y); *a = x; }

// ---- lane-pair cooperative addition ---------------------------------------------------------------------------
// A full XYZZ addition is 14 dependent-ish Fq products = ~15 us on one lane, and the phases after the bucket
// accumulation (slice tree, chunk running sums, segment folds) are chains of them with little parallelism.  The 14
// products pack perfectly into 7 levels x 2 lanes, so two adjacent lanes (2k, 2k+1) share one addition: same total
// work, half the latency.  Both lanes execute the same instruction stream on operands chosen by lane parity; values
// cross with one DPP/swizzle exchange per level (__shfl_xor 1).
//     level   even lane (a)             odd lane (b)
//       1     U1 = X1*ZZ2               U2 = X2*ZZ1
//       2     S1 = Y1*ZZZ2              S2 = Y2*ZZZ1
//       3     PP = P^2                  RR = R^2                 (P = U2-U1, R = S2-S1 on both)
//       4     Z12 = ZZ1*ZZ2             ZZZ12 = ZZZ1*ZZZ2
//       5     PPP = P*PP                Q = U1*PP
//       6     ZZ3 = Z12*PP              SP = S1*PPP
//       7     ZZZ3 = ZZZ12*PPP          Rt = R*(Q - X3)          (X3 = RR - PPP - 2Q on both)
// a stores ZZ3, ZZZ3; b stores X3, Y3 = Rt - SP.  Operands come from memory (global or LDS, generic pointers) so each
// lane loads exactly the coordinates its levels need.  out may alias pa or pb.  Both lanes must call it together.
__device__ __forceinline__ Fq fq_xchg(const Fq& a) {
  Fq r;
#pragma unroll
  for (int i = 0; i < Fq::N; ++i) r.v[i] = (uint32_t)__shfl_xor((int)a.v[i], 1);
  return r;
}
__device__ __forceinline__ Fq fq_sel(bool take_b, const Fq& a, const Fq& b) {
  Fq r;
#pragma unroll
  for (int i = 0; i < Fq::N; ++i) r.v[i] = take_b ? b.v[i] : a.v[i];
  return r;
}
__device__ __noinline__ void xyzz_add_pair_rare(const char* pa, const char* pb, char* out) {   // P == +-Q: one lane, general code
  XYZZ a = load_xyzz(pa), b = load_xyzz(pb);
  xyzz_add(a, b);
  store_xyzz(out, a);
}
__device__ __forceinline__ void xyzz_add_pair(const char* pa, const char* pb, char* out) {
  const bool odd = threadIdx.x & 1;
  const Fq zzA = load_fp<Fq>(pa + 96), zzB = load_fp<Fq>(pb + 96);
  const bool infA = zzA.is_zero_raw(), infB = zzB.is_zero_raw();
  if (infA || infB) {              // pair-uniform: both lanes see the same two points
    const char* src = infB ? pa : pb;          // A + O = A ; O + B = B ; O + O = O (either)
    if (src != out) {              // each lane copies half of the 192 bytes
      const uint4* s4 = (const uint4*)(src + (odd ? 96 : 0)); uint4* d4 = (uint4*)(out + (odd ? 96 : 0));
#pragma unroll
      for (int i = 0; i < 6; ++i) d4[i] = s4[i];
    }
    return;
  }
  const char* own = odd ? pb : pa; const char* oth = odd ? pa : pb;
  Fq u = Fq::mul(load_fp<Fq>(own), odd ? zzA : zzB);                          // a: U1   b: U2     (< 2q)
  Fq s = Fq::mul(load_fp<Fq>(own + 48), load_fp<Fq>(oth + 144));               // a: S1   b: S2
  Fq pu = fq_xchg(u), ps = fq_xchg(s);
  Fq U1 = fq_sel(odd, u, pu), U2 = fq_sel(odd, pu, u), S1 = fq_sel(odd, s, ps), S2 = fq_sel(odd, ps, s);
  Fq P = Fq::sub<2>(U2, U1), R = Fq::sub<2>(S2, S1);                            // < 4q
  Fq t3 = Fq::sqr(fq_sel(odd, P, R));                                          // a: PP   b: RR
  Fq t4 = Fq::mul(load_fp<Fq>(pa + (odd ? 144 : 96)), load_fp<Fq>(pb + (odd ? 144 : 96)));   // a: ZZ1*ZZ2   b: ZZZ1*ZZZ2
  Fq pt3 = fq_xchg(t3);
  Fq PP = fq_sel(odd, t3, pt3), RR = fq_sel(odd, pt3, t3);
  Fq t5 = Fq::mul(fq_sel(odd, P, U1), PP);                                     // a: PPP  b: Q
  Fq pt5 = fq_xchg(t5);
  Fq PPP = fq_sel(odd, t5, pt5), Q = fq_sel(odd, pt5, t5);
  Fq X3 = Fq::sub<4>(Fq::sub<2>(RR, PPP), Fq::dbl(Q));                          // < 8q
  Fq t6 = Fq::mul(fq_sel(odd, t4, S1), fq_sel(odd, PP, PPP));                  // a: ZZ3  b: SP
  Fq pt4 = fq_xchg(t4);                                                        // a receives ZZZ12
  Fq t7 = Fq::mul(fq_sel(odd, pt4, R), fq_sel(odd, PPP, Fq::sub<8>(Q, X3)));   // a: ZZZ3 b: Rt
  // same-x case (doubling / cancellation): ZZ3 == 0 mod q, seen by the even lane
  int z = (!odd && t6.is_zero_mod_lt2p()) ? 1 : 0;
  z = __shfl(z, (int)(threadIdx.x & 63u & ~1u));
  if (__builtin_expect(z, 0)) { if (!odd) xyzz_add_pair_rare(pa, pb, out); return; }
  if (odd) { store_fp<Fq>(out, X3); store_fp<Fq>(out + 48, Fq::sub<2>(t7, t6)); }
  else { store_fp<Fq>(out + 96, t6); store_fp<Fq>(out + 144, t7); }
}

// Normalise the infinity encoding before a store: a lazily reduced ZZ that is 0 mod q becomes raw 0.
__device__ __forceinline__ void xyzz_store_normalized(void* p, XYZZ a, bool inf) {
  if (inf) a = xyzz_infinity();
  store_xyzz(p, a);
}

}  // namespace aleo_mi355x
