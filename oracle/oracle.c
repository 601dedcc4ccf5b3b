/*
 * oracle.c — CPU restatement of the snarkVM 0.14.5 operators on the Aleo execute/prove hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so.  The product (aleo_amd/csrc, libaleo_mi355x.so) never includes, links or calls this.
 *
 * PARITY UNPINNED for operator *values*: the arithmetic lives in crates.io snarkvm-fields /
 * snarkvm-curves / snarkvm-algorithms =0.14.5 (pins: /root/reference/Cargo.lock:2200,2637,2652), which are
 * not under /root/reference and cannot be built here (no Rust toolchain, no network).  This file restates
 * their published algorithms [UPSTREAM-RECALL: paths below are upstream-relative] and is anchored on
 *   - the reference's call sites: rust/src/program/execute.rs:74,177 ; transfer.rs:99 (prove path),
 *   - the reference's own fixture: ten KZG commitments + field evaluations in the `proof1…` string at
 *     wasm/src/programs/transaction.rs:100  (tests/golden/reference_proof.json; checked by tests/test_oracle.py),
 *   - Python big-integer known answers (oracle/pyref.py -> tests/golden/*.json).
 *
 * Restated items:
 *   fields/src/fp_256.rs, fp_384.rs                    Fp256/Fp384 Montgomery arithmetic (4/6 x u64 limbs, LE)
 *   curves/src/bls12_377/{fr,fq,g1}.rs                 constants
 *   curves/src/templates/short_weierstrass_jacobian/   Affine{x,y,infinity}, Projective = Jacobian{x,y,z}:
 *       projective.rs: add_assign_mixed (madd-2007-bl), double_in_place (dbl-2009-l, a=0), add_assign (add-2007-bl),
 *       batch_normalization, to_affine
 *   algorithms/src/msm/variable_base/standard.rs       msm(): window c = 3 if n<32 else ln_without_floats(n)+2,
 *       (1<<c)-1 buckets per window, scalar==1 fast path in window 0, running-sum, Horner combine
 *   algorithms/src/msm/variable_base/batched.rs        same windows; buckets filled by batched affine additions
 *       (one shared inversion per batch) — restated here as msm_batched (CPU baseline; same group element)
 *   algorithms/src/fft/domain.rs                       EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place:
 *       in-order in, in-order out, omega = TWO_ADIC_ROOT^(2^(47-k)), coset shift g = 22, inverse scales by n^-1
 *
 * Build: see oracle/Makefile  (gcc -O3 -march=native -shared -fPIC, pthreads).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------------
 * Generic Montgomery field over N u64 limbs (fields/src/fp_256.rs, fp_384.rs)
 * ---------------------------------------------------------------------------------------------- */
#define DEFINE_FIELD(F, N)                                                                         \
  typedef struct { u64 l[N]; } F;                                                                  \
  static const F F##_MOD, F##_R1, F##_R2; static const u64 F##_INV;                                \
  static inline int F##_is_zero(const F* a) { u64 o = 0; for (int i = 0; i < N; ++i) o |= a->l[i]; return o == 0; } \
  static inline int F##_eq(const F* a, const F* b) { u64 o = 0; for (int i = 0; i < N; ++i) o |= a->l[i] ^ b->l[i]; return o == 0; } \
  static inline int F##_geq(const u64* a, const u64* b) {                                          \
    for (int i = N - 1; i >= 0; --i) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; } return 1; } \
  static inline void F##_sub_raw(u64* r, const u64* a, const u64* b) {                             \
    u64 br = 0; for (int i = 0; i < N; ++i) { u128 t = (u128)a[i] - b[i] - br; r[i] = (u64)t; br = (u64)(t >> 64) & 1; } } \
  static inline void F##_add(F* r, const F* a, const F* b) {                                       \
    u64 c = 0; u64 t[N]; for (int i = 0; i < N; ++i) { u128 s = (u128)a->l[i] + b->l[i] + c; t[i] = (u64)s; c = (u64)(s >> 64); } \
    if (c || F##_geq(t, F##_MOD.l)) F##_sub_raw(t, t, F##_MOD.l);                                  \
    memcpy(r->l, t, sizeof t); }                                                                   \
  static inline void F##_sub(F* r, const F* a, const F* b) {                                       \
    u64 t[N]; u64 br = 0; for (int i = 0; i < N; ++i) { u128 s = (u128)a->l[i] - b->l[i] - br; t[i] = (u64)s; br = (u64)(s >> 64) & 1; } \
    if (br) { u64 c = 0; for (int i = 0; i < N; ++i) { u128 s = (u128)t[i] + F##_MOD.l[i] + c; t[i] = (u64)s; c = (u64)(s >> 64); } } \
    memcpy(r->l, t, sizeof t); }                                                                   \
  static inline void F##_neg(F* r, const F* a) { if (F##_is_zero(a)) { *r = *a; } else { F##_sub_raw(r->l, F##_MOD.l, a->l); } } \
  static inline void F##_dbl(F* r, const F* a) { F##_add(r, a, a); }                               \
  /* CIOS Montgomery product: r = a*b*R^-1 mod p */                                                \
  static inline void F##_mul(F* r, const F* a, const F* b) {                                       \
    u64 t[N + 2]; memset(t, 0, sizeof t);                                                          \
    for (int i = 0; i < N; ++i) {                                                                  \
      u64 c = 0;                                                                                   \
      for (int j = 0; j < N; ++j) { u128 s = (u128)a->l[j] * b->l[i] + t[j] + c; t[j] = (u64)s; c = (u64)(s >> 64); } \
      u128 s = (u128)t[N] + c; t[N] = (u64)s; t[N + 1] = (u64)(s >> 64);                           \
      u64 m = t[0] * F##_INV;                                                                      \
      s = (u128)m * F##_MOD.l[0] + t[0]; c = (u64)(s >> 64);                                       \
      for (int j = 1; j < N; ++j) { s = (u128)m * F##_MOD.l[j] + t[j] + c; t[j - 1] = (u64)s; c = (u64)(s >> 64); } \
      s = (u128)t[N] + c; t[N - 1] = (u64)s; t[N] = t[N + 1] + (u64)(s >> 64);                     \
    }                                                                                              \
    if (t[N] || F##_geq(t, F##_MOD.l)) F##_sub_raw(t, t, F##_MOD.l);                               \
    memcpy(r->l, t, N * sizeof(u64)); }                                                            \
  static inline void F##_sqr(F* r, const F* a) { F##_mul(r, a, a); }                               \
  static inline void F##_to_mont(F* r, const F* a) { F##_mul(r, a, &F##_R2); }                     \
  static inline void F##_from_mont(F* r, const F* a) { F one; memset(&one, 0, sizeof one); one.l[0] = 1; F##_mul(r, a, &one); } \
  /* a^e, e given as N little-endian limbs (square-and-multiply, MSB first) */                     \
  static void F##_pow(F* r, const F* a, const u64* e, int nl) {                                    \
    F acc = F##_R1;                                                                                \
    for (int i = nl * 64 - 1; i >= 0; --i) { F##_sqr(&acc, &acc); if ((e[i / 64] >> (i % 64)) & 1) F##_mul(&acc, &acc, a); } \
    *r = acc; }                                                                                    \
  /* inverse by Fermat: a^(p-2); 0 -> 0 (callers test for zero first, as Field::inverse() returns None) */ \
  static void F##_inv(F* r, const F* a) { u64 e[N]; memcpy(e, F##_MOD.l, sizeof e); e[0] -= 2; F##_pow(r, a, e, N); }

DEFINE_FIELD(Fr, 4)
DEFINE_FIELD(Fq, 6)

/* curves/src/bls12_377/fr.rs */
static const Fr Fr_MOD = {{0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL}};
static const Fr Fr_R1 = {{0x7d1c7ffffffffff3ULL, 0x7257f50f6ffffff2ULL, 0x16d81575512c0feeULL, 0x0d4bda322bbb9a9dULL}};
static const Fr Fr_R2 = {{0x25d577bab861857bULL, 0xcc2c27b58860591fULL, 0xa7cc008fe5dc8593ULL, 0x011fdae7eff1c939ULL}};
static const u64 Fr_INV = 0x0a117fffffffffffULL;
/* TWO_ADIC_ROOT_OF_UNITY (canonical) = 22^((r-1)/2^47); SURVEY.md §0 fact 5 */
static const Fr FR_ROOT_CANON = {{0x476ef4a4ec2a895eULL, 0x9b506ee363e3f04aULL, 0x60c69477d1a8a12fULL, 0x11d4b7f60cb92cc1ULL}};
#define FR_TWO_ADICITY 47
#define FR_GENERATOR 22
#define FR_BITS 253
/* curves/src/bls12_377/fq.rs */
static const Fq Fq_MOD = {{0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL, 0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL}};
static const Fq Fq_R1 = {{0x02cdffffffffff68ULL, 0x51409f837fffffb1ULL, 0x9f7db3a98a7d3ff2ULL, 0x7b4e97b76e7c6305ULL, 0x4cf495bf803c84e8ULL, 0x008d6661e2fdf49aULL}};
static const Fq Fq_R2 = {{0xb786686c9400cd22ULL, 0x0329fcaab00431b1ULL, 0x22a5f11162d6b46dULL, 0xbfdf7d03827dc3acULL, 0x837e92f041790bf9ULL, 0x006dfccb1e914b88ULL}};
static const u64 Fq_INV = 0x8508bfffffffffffULL;

/* ------------------------------------------------------------------------------------------------
 * G1 (curves/src/templates/short_weierstrass_jacobian): y^2 = x^3 + 1
 * In-memory layouts follow snarkVM: Affine = {x: Fq, y: Fq, infinity: bool} -> 104-byte stride;
 * Projective = {x,y,z: Fq} (Jacobian), 144 bytes; all coordinates Montgomery form.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { Fq x, y; uint8_t infinity; uint8_t pad[7]; } G1Affine;   /* sizeof == 104 */
typedef struct { Fq x, y, z; } G1Proj;                                    /* Jacobian */

static inline void g1p_zero(G1Proj* p) { p->x = Fq_R1; p->y = Fq_R1; memset(&p->z, 0, sizeof p->z); } /* (1,1,0) */
static inline int g1p_is_zero(const G1Proj* p) { return Fq_is_zero(&p->z); }

/* projective.rs double_in_place, a = 0 branch (dbl-2009-l) */
static void g1p_double(G1Proj* p) {
  if (g1p_is_zero(p)) return;
  Fq a, b, c, d, e, f, t;
  Fq_sqr(&a, &p->x); Fq_sqr(&b, &p->y); Fq_sqr(&c, &b);
  Fq_add(&t, &p->x, &b); Fq_sqr(&t, &t); Fq_sub(&t, &t, &a); Fq_sub(&t, &t, &c); Fq_dbl(&d, &t);
  Fq_dbl(&e, &a); Fq_add(&e, &e, &a);
  Fq_sqr(&f, &e);
  Fq_mul(&p->z, &p->z, &p->y); Fq_dbl(&p->z, &p->z);
  Fq_sub(&p->x, &f, &d); Fq_sub(&p->x, &p->x, &d);
  Fq_sub(&t, &d, &p->x); Fq_mul(&t, &t, &e);
  Fq_dbl(&c, &c); Fq_dbl(&c, &c); Fq_dbl(&c, &c);
  Fq_sub(&p->y, &t, &c);
}

/* projective.rs add_assign_mixed (madd-2007-bl) */
static void g1p_add_mixed(G1Proj* p, const G1Affine* q) {
  if (q->infinity) return;
  if (g1p_is_zero(p)) { p->x = q->x; p->y = q->y; p->z = Fq_R1; return; }
  Fq z1z1, u2, s2, h, hh, i, j, r, v, t;
  Fq_sqr(&z1z1, &p->z);
  Fq_mul(&u2, &q->x, &z1z1);
  Fq_mul(&s2, &q->y, &p->z); Fq_mul(&s2, &s2, &z1z1);
  if (Fq_eq(&p->x, &u2) && Fq_eq(&p->y, &s2)) { g1p_double(p); return; }
  Fq_sub(&h, &u2, &p->x);
  Fq_sqr(&hh, &h);
  Fq_dbl(&i, &hh); Fq_dbl(&i, &i);
  Fq_mul(&j, &h, &i);
  Fq_sub(&r, &s2, &p->y); Fq_dbl(&r, &r);
  Fq_mul(&v, &p->x, &i);
  Fq_sqr(&t, &r); Fq_sub(&t, &t, &j); Fq_sub(&t, &t, &v); Fq_sub(&t, &t, &v);   /* X3 */
  Fq y1j; Fq_mul(&y1j, &p->y, &j); Fq_dbl(&y1j, &y1j);
  Fq_sub(&v, &v, &t); Fq_mul(&v, &v, &r); Fq_sub(&p->y, &v, &y1j);              /* Y3 */
  p->x = t;
  Fq_add(&t, &p->z, &h); Fq_sqr(&t, &t); Fq_sub(&t, &t, &z1z1); Fq_sub(&p->z, &t, &hh); /* Z3 */
}

/* projective.rs add_assign (add-2007-bl) */
static void g1p_add(G1Proj* p, const G1Proj* q) {
  if (g1p_is_zero(p)) { *p = *q; return; }
  if (g1p_is_zero(q)) return;
  Fq z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t;
  Fq_sqr(&z1z1, &p->z); Fq_sqr(&z2z2, &q->z);
  Fq_mul(&u1, &p->x, &z2z2); Fq_mul(&u2, &q->x, &z1z1);
  Fq_mul(&s1, &p->y, &q->z); Fq_mul(&s1, &s1, &z2z2);
  Fq_mul(&s2, &q->y, &p->z); Fq_mul(&s2, &s2, &z1z1);
  if (Fq_eq(&u1, &u2) && Fq_eq(&s1, &s2)) { g1p_double(p); return; }
  Fq_sub(&h, &u2, &u1);
  Fq_dbl(&i, &h); Fq_sqr(&i, &i);
  Fq_mul(&j, &h, &i);
  Fq_sub(&r, &s2, &s1); Fq_dbl(&r, &r);
  Fq_mul(&v, &u1, &i);
  Fq x3; Fq_sqr(&x3, &r); Fq_sub(&x3, &x3, &j); Fq_sub(&x3, &x3, &v); Fq_sub(&x3, &x3, &v);
  Fq_mul(&s1, &s1, &j); Fq_dbl(&s1, &s1);
  Fq_sub(&t, &v, &x3); Fq_mul(&t, &t, &r); Fq_sub(&p->y, &t, &s1);
  Fq_add(&t, &p->z, &q->z); Fq_sqr(&t, &t); Fq_sub(&t, &t, &z1z1); Fq_sub(&t, &t, &z2z2); Fq_mul(&p->z, &t, &h);
  p->x = x3;
}

/* projective.rs to_affine */
static void g1p_to_affine(G1Affine* a, const G1Proj* p) {
  memset(a, 0, sizeof *a);
  if (g1p_is_zero(p)) { a->infinity = 1; return; }   /* Affine::zero() = (0, 1?, inf): only the flag is compared */
  Fq zi, zi2;
  Fq_inv(&zi, &p->z); Fq_sqr(&zi2, &zi);
  Fq_mul(&a->x, &p->x, &zi2);
  Fq_mul(&a->y, &p->y, &zi2); Fq_mul(&a->y, &a->y, &zi);
}

/* projective.rs batch_normalization: Montgomery's trick over the non-zero z's */
static void g1p_batch_normalize(G1Affine* out, const G1Proj* in, size_t n) {
  Fq* pre = (Fq*)malloc(sizeof(Fq) * (n + 1));
  Fq acc = Fq_R1;
  for (size_t i = 0; i < n; ++i) { pre[i] = acc; if (!g1p_is_zero(&in[i])) Fq_mul(&acc, &acc, &in[i].z); }
  Fq inv; Fq_inv(&inv, &acc);
  for (size_t i = n; i-- > 0;) {
    memset(&out[i], 0, sizeof out[i]);
    if (g1p_is_zero(&in[i])) { out[i].infinity = 1; continue; }
    Fq zi, zi2; Fq_mul(&zi, &inv, &pre[i]); Fq_mul(&inv, &inv, &in[i].z);
    Fq_sqr(&zi2, &zi);
    Fq_mul(&out[i].x, &in[i].x, &zi2);
    Fq_mul(&out[i].y, &in[i].y, &zi2); Fq_mul(&out[i].y, &out[i].y, &zi);
  }
  free(pre);
}

static int g1a_on_curve(const G1Affine* a) {
  if (a->infinity) return 1;
  Fq l, r; Fq_sqr(&l, &a->y); Fq_sqr(&r, &a->x); Fq_mul(&r, &r, &a->x); Fq_add(&r, &r, &Fq_R1);
  return Fq_eq(&l, &r);
}

/* ------------------------------------------------------------------------------------------------
 * VariableBase::msm (algorithms/src/msm/variable_base/standard.rs)
 * scalars: canonical (non-Montgomery) BigInteger256, LE limbs.
 * ---------------------------------------------------------------------------------------------- */
static inline unsigned ln_without_floats(size_t a) {  /* msm/mod.rs: (log2(a) * 69 / 100) */
  unsigned lg = 0; while ((a >> lg) > 1) ++lg; return lg * 69 / 100;
}
static inline u64 scalar_window(const u64* s, unsigned w_start, unsigned c) {
  /* (scalar >> w_start) % (1 << c) over 4 limbs */
  unsigned limb = w_start / 64, off = w_start % 64;
  u64 v = s[limb] >> off;
  if (off + c > 64 && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
  return v & ((1ULL << c) - 1);
}
static inline int scalar_is_one(const u64* s) { return s[0] == 1 && (s[1] | s[2] | s[3]) == 0; }
static inline int scalar_is_zero(const u64* s) { return (s[0] | s[1] | s[2] | s[3]) == 0; }

typedef struct {
  const G1Affine* bases; const u64* scalars; size_t n; unsigned c; unsigned w_start;
  G1Proj out; int batched;
} window_job;

static void window_standard(window_job* J) {
  size_t nb = ((size_t)1 << J->c) - 1;
  G1Proj res; g1p_zero(&res);
  G1Proj* buckets = (G1Proj*)malloc(sizeof(G1Proj) * nb);
  for (size_t i = 0; i < nb; ++i) g1p_zero(&buckets[i]);
  for (size_t i = 0; i < J->n; ++i) {
    const u64* s = J->scalars + 4 * i;
    if (scalar_is_zero(s)) continue;
    if (scalar_is_one(s)) { if (J->w_start == 0) g1p_add_mixed(&res, &J->bases[i]); continue; }
    u64 d = scalar_window(s, J->w_start, J->c);
    if (d != 0) g1p_add_mixed(&buckets[d - 1], &J->bases[i]);
  }
  G1Affine* norm = (G1Affine*)malloc(sizeof(G1Affine) * nb);
  g1p_batch_normalize(norm, buckets, nb);
  G1Proj running; g1p_zero(&running);
  for (size_t b = nb; b-- > 0;) { g1p_add_mixed(&running, &norm[b]); g1p_add(&res, &running); }
  free(norm); free(buckets);
  J->out = res;
}

/* batched.rs restated: buckets are filled by rounds of pairwise affine additions that share one
 * inversion per round (Montgomery's trick); identical group element, far fewer Fq products. */
typedef struct { uint32_t bucket, idx; } bpair;
static int bpair_cmp(const void* a, const void* b) {
  const bpair* x = (const bpair*)a; const bpair* y = (const bpair*)b;
  return (x->bucket > y->bucket) - (x->bucket < y->bucket);
}
static void window_batched(window_job* J) {
  size_t nb = ((size_t)1 << J->c) - 1;
  G1Proj res; g1p_zero(&res);
  bpair* pr = (bpair*)malloc(sizeof(bpair) * (J->n ? J->n : 1)); size_t m = 0;
  for (size_t i = 0; i < J->n; ++i) {
    const u64* s = J->scalars + 4 * i;
    if (scalar_is_zero(s) || J->bases[i].infinity) continue;
    if (scalar_is_one(s)) { if (J->w_start == 0) g1p_add_mixed(&res, &J->bases[i]); continue; }
    u64 d = scalar_window(s, J->w_start, J->c);
    if (d != 0) { pr[m].bucket = (uint32_t)(d - 1); pr[m].idx = (uint32_t)i; ++m; }
  }
  qsort(pr, m, sizeof(bpair), bpair_cmp);
  /* working set: affine points tagged with their bucket; reduce every bucket to <= 1 point */
  G1Affine* cur = (G1Affine*)malloc(sizeof(G1Affine) * (m ? m : 1));
  uint32_t* tag = (uint32_t*)malloc(sizeof(uint32_t) * (m ? m : 1));
  for (size_t k = 0; k < m; ++k) { cur[k] = J->bases[pr[k].idx]; tag[k] = pr[k].bucket; }
  free(pr);
  Fq* den = (Fq*)malloc(sizeof(Fq) * (m / 2 + 1)); Fq* pre = (Fq*)malloc(sizeof(Fq) * (m / 2 + 1));
  size_t* lhs = (size_t*)malloc(sizeof(size_t) * (m / 2 + 1));
  G1Proj* spill = (G1Proj*)calloc(nb, sizeof(G1Proj));   /* doubling / cancellation cases fall back to Jacobian */
  for (size_t b = 0; b < nb; ++b) g1p_zero(&spill[b]);
  const uint32_t DEAD = 0xFFFFFFFFu;
  while (1) {
    size_t np = 0, w = 0, k = 0; int spilled = 0;
    /* pair neighbours with equal tags */
    while (k < m) {
      if (k + 1 < m && tag[k] == tag[k + 1]) {
        if (Fq_eq(&cur[k].x, &cur[k + 1].x)) {  /* P == +-Q: rare, handled outside the batch (Jacobian spill) */
          g1p_add_mixed(&spill[tag[k]], &cur[k]); g1p_add_mixed(&spill[tag[k]], &cur[k + 1]);
          tag[k] = tag[k + 1] = DEAD; spilled = 1; k += 2; continue;
        }
        lhs[np] = k; Fq_sub(&den[np], &cur[k + 1].x, &cur[k].x); ++np; k += 2;
      } else { ++k; }
    }
    if (np == 0 && !spilled) break;
    if (np) {
      Fq acc = Fq_R1;
      for (size_t p = 0; p < np; ++p) { pre[p] = acc; Fq_mul(&acc, &acc, &den[p]); }
      Fq inv; Fq_inv(&inv, &acc);
      for (size_t p = np; p-- > 0;) { Fq di; Fq_mul(&di, &inv, &pre[p]); Fq_mul(&inv, &inv, &den[p]); den[p] = di; }
    }
    /* rewrite the array: sums replace pairs, singletons carry over, spilled entries vanish */
    size_t p = 0; k = 0;
    while (k < m) {
      if (tag[k] == DEAD) { ++k; continue; }
      if (p < np && lhs[p] == k) {
        G1Affine *A = &cur[k], *B = &cur[k + 1], S; memset(&S, 0, sizeof S);
        Fq lam, t; Fq_sub(&lam, &B->y, &A->y); Fq_mul(&lam, &lam, &den[p]);
        Fq_sqr(&S.x, &lam); Fq_sub(&S.x, &S.x, &A->x); Fq_sub(&S.x, &S.x, &B->x);
        Fq_sub(&t, &A->x, &S.x); Fq_mul(&t, &t, &lam); Fq_sub(&S.y, &t, &A->y);
        uint32_t tg = tag[k]; cur[w] = S; tag[w] = tg; ++w; ++p; k += 2;
      } else { uint32_t tg = tag[k]; cur[w] = cur[k]; tag[w] = tg; ++w; ++k; }
    }
    m = w;
  }
  for (size_t k = 0; k < m; ++k) g1p_add_mixed(&spill[tag[k]], &cur[k]);
  free(cur); free(tag); free(den); free(pre); free(lhs);
  G1Proj running; g1p_zero(&running);
  for (size_t b = nb; b-- > 0;) { g1p_add(&running, &spill[b]); g1p_add(&res, &running); }
  free(spill);
  J->out = res;
}

typedef struct { window_job* jobs; int njobs; int next; pthread_mutex_t mu; } job_pool;
static void* pool_worker(void* arg) {
  job_pool* P = (job_pool*)arg;
  for (;;) {
    pthread_mutex_lock(&P->mu); int k = P->next++; pthread_mutex_unlock(&P->mu);
    if (k >= P->njobs) break;
    if (P->jobs[k].batched) window_batched(&P->jobs[k]); else window_standard(&P->jobs[k]);
  }
  return NULL;
}

/* out: Jacobian (x,y,z) Montgomery, 144 bytes.  bases: stride bytes per element (104 = snarkVM Affine, or 96
 * = x,y only with no infinity flag).  threads <= 0 -> 1.  variant 0 = standard.rs, 1 = batched.rs; 2 / 3 = the same with the points of
 * every window also split over the threads (not how the reference parallelises: used for the all-cores CPU baseline) */
int oracle_msm_g1(void* out, const void* bases_, size_t stride, const void* scalars_, size_t n, int threads, int variant) {
  if (stride != 104 && stride != 96) return -1;
  G1Affine* bases = (G1Affine*)malloc(sizeof(G1Affine) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* src = (const uint8_t*)bases_ + i * stride;
    memset(&bases[i], 0, sizeof(G1Affine)); memcpy(&bases[i], src, 96);
    bases[i].infinity = (stride == 104) ? (src[96] != 0) : 0;
  }
  const u64* scalars = (const u64*)scalars_;
  /* variant 2/3 = variant 0/1 with the points cut into `parts` contiguous ranges, every range a full MSM of its own (its own window
   * size from its own length, its windows as parallel jobs, its own Horner), the range results added: the reference parallelises over
   * windows only, i.e. <= ceil(253/c) threads; this lets the CPU baseline use a whole box. */
  int parts = 1;
  { unsigned c0 = n < 32 ? 3 : ln_without_floats(n) + 2; int nw0 = (FR_BITS + c0 - 1) / c0;
    if (variant >= 2) { variant -= 2; parts = threads > 1 ? (4 * threads + nw0 - 1) / nw0 : 1;          /* ~4 jobs per thread: the last round of jobs stays short */
      if ((size_t)parts > n / 4096 + 1) parts = (int)(n / 4096 + 1); } }
  unsigned* pc = (unsigned*)malloc(sizeof(unsigned) * parts); int* pw0 = (int*)malloc(sizeof(int) * (parts + 1)); pw0[0] = 0;
  for (int p = 0; p < parts; ++p) {
    size_t np = n * (size_t)(p + 1) / parts - n * (size_t)p / parts;
    pc[p] = np < 32 ? 3 : ln_without_floats(np) + 2; pw0[p + 1] = pw0[p] + (int)((FR_BITS + pc[p] - 1) / pc[p]);
  }
  window_job* jobs = (window_job*)calloc((size_t)pw0[parts], sizeof(window_job));
  for (int p = 0; p < parts; ++p) for (int w = 0; w < pw0[p + 1] - pw0[p]; ++w) {
    window_job* J = &jobs[pw0[p] + w]; size_t lo = n * (size_t)p / parts, hi = n * (size_t)(p + 1) / parts;
    J->bases = bases + lo; J->scalars = scalars + 4 * lo; J->n = hi - lo; J->c = pc[p]; J->w_start = w * pc[p]; J->batched = variant;
  }
  job_pool P; P.jobs = jobs; P.njobs = pw0[parts]; P.next = 0; pthread_mutex_init(&P.mu, NULL);
  if (threads <= 1) pool_worker(&P);
  else {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, pool_worker, &P);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(th);
  }
  /* standard.rs: lowest + fold(rest.rev(), |total, w| { total += w; c doublings }), per range */
  G1Proj total; g1p_zero(&total);
  for (int p = 0; p < parts; ++p) {
    G1Proj t; g1p_zero(&t); const window_job* Jp = jobs + pw0[p]; int nw = pw0[p + 1] - pw0[p]; unsigned c = pc[p];
    for (int w = nw - 1; w >= 1; --w) { g1p_add(&t, &Jp[w].out); for (unsigned d = 0; d < c; ++d) g1p_double(&t); }
    g1p_add(&t, &Jp[0].out);
    g1p_add(&total, &t);
  }
  free(pc); free(pw0);
  memcpy(out, &total, sizeof total);
  free(jobs); free(bases);
  return 0;
}

/* Naive double-and-add reference: sum s_i * P_i with no windows (cross-check for the Pippenger paths). */
int oracle_msm_g1_naive(void* out, const void* bases_, size_t stride, const void* scalars_, size_t n) {
  G1Proj total; g1p_zero(&total);
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* src = (const uint8_t*)bases_ + i * stride;
    G1Affine b; memset(&b, 0, sizeof b); memcpy(&b, src, 96); b.infinity = (stride == 104) ? (src[96] != 0) : 0;
    const u64* s = (const u64*)scalars_ + 4 * i;
    G1Proj acc; g1p_zero(&acc);
    for (int bit = 255; bit >= 0; --bit) { g1p_double(&acc); if ((s[bit / 64] >> (bit % 64)) & 1) g1p_add_mixed(&acc, &b); }
    g1p_add(&total, &acc);
  }
  memcpy(out, &total, sizeof total);
  return 0;
}

/* Jacobian -> snarkVM Affine (104 bytes, Montgomery) */
void oracle_g1_to_affine(void* out104, const void* jac144) {
  G1Proj p; memcpy(&p, jac144, sizeof p); G1Affine a; g1p_to_affine(&a, &p); memcpy(out104, &a, 104);
}
int oracle_g1_on_curve(const void* aff104) { G1Affine a; memcpy(&a, aff104, 104); return g1a_on_curve(&a); }
/* scalar * base for one pair, affine out (used to build structured bases) */
void oracle_g1_mul(void* out104, const void* base104, const void* scalar32) {
  G1Proj t; oracle_msm_g1_naive(&t, base104, 104, scalar32, 1); oracle_g1_to_affine(out104, &t);
}
/* out[i] = (start + i*step... ) : consecutive multiples P_i = (i+1) * G by repeated mixed addition */
void oracle_g1_multiples(void* out104, const void* base104, size_t n) {
  G1Affine g; memcpy(&g, base104, 104);
  G1Proj* acc = (G1Proj*)malloc(sizeof(G1Proj) * (n ? n : 1));
  G1Proj run; g1p_zero(&run);
  for (size_t i = 0; i < n; ++i) { g1p_add_mixed(&run, &g); acc[i] = run; }
  G1Affine* aff = (G1Affine*)malloc(sizeof(G1Affine) * (n ? n : 1));
  g1p_batch_normalize(aff, acc, n);
  memcpy(out104, aff, n * 104);
  free(acc); free(aff);
}

/* Montgomery <-> canonical conversions for tests (n elements of 4 or 6 limbs) */
void oracle_fr_to_mont(void* io, size_t n) { Fr* p = (Fr*)io; for (size_t i = 0; i < n; ++i) Fr_to_mont(&p[i], &p[i]); }
void oracle_fr_from_mont(void* io, size_t n) { Fr* p = (Fr*)io; for (size_t i = 0; i < n; ++i) Fr_from_mont(&p[i], &p[i]); }
void oracle_fq_to_mont(void* io, size_t n) { Fq* p = (Fq*)io; for (size_t i = 0; i < n; ++i) Fq_to_mont(&p[i], &p[i]); }
void oracle_fq_from_mont(void* io, size_t n) { Fq* p = (Fq*)io; for (size_t i = 0; i < n; ++i) Fq_from_mont(&p[i], &p[i]); }
void oracle_fr_mul(void* r, const void* a, const void* b, size_t n) { for (size_t i = 0; i < n; ++i) Fr_mul((Fr*)r + i, (const Fr*)a + i, (const Fr*)b + i); }
void oracle_fq_mul(void* r, const void* a, const void* b, size_t n) { for (size_t i = 0; i < n; ++i) Fq_mul((Fq*)r + i, (const Fq*)a + i, (const Fq*)b + i); }

/* ------------------------------------------------------------------------------------------------
 * EvaluationDomain (algorithms/src/fft/domain.rs).  Data: n Fr elements, Montgomery form, in place.
 * order: 0 NN (in-order in/out — what fft_in_place / ifft_in_place expose), 1 NR, 2 RN, 3 RR
 *        (N natural, R bit-reversed; mirrors snarkvm-algorithms-cuda NTTInputOutputOrder)
 * direction: 0 forward, 1 inverse (multiplies by size_inv).  type: 0 standard, 1 coset (shift g = 22:
 * coset_fft = distribute_powers(g) then fft; coset_ifft = ifft then distribute_powers(g^-1)).
 * ---------------------------------------------------------------------------------------------- */
static void fr_from_u64(Fr* r, u64 v) { Fr t; memset(&t, 0, sizeof t); t.l[0] = v; Fr_to_mont(r, &t); }
static void bitrev_permute(Fr* x, size_t n, unsigned lg) {
  for (size_t i = 0; i < n; ++i) {
    size_t j = 0; for (unsigned b = 0; b < lg; ++b) j |= ((i >> b) & 1) << (lg - 1 - b);
    if (i < j) { Fr t = x[i]; x[i] = x[j]; x[j] = t; }
  }
}
static void distribute_powers(Fr* x, size_t n, const Fr* g) {
  Fr pw = Fr_R1; for (size_t i = 0; i < n; ++i) { Fr_mul(&x[i], &x[i], &pw); Fr_mul(&pw, &pw, g); }
}
int oracle_ntt_fr(void* inout, unsigned lg_n, int order, int direction, int type) {
  if (lg_n > FR_TWO_ADICITY) return -1;
  size_t n = (size_t)1 << lg_n; Fr* x = (Fr*)inout;
  /* group_gen = TWO_ADIC_ROOT^(2^(47-lg_n)) */
  Fr w; Fr_to_mont(&w, &FR_ROOT_CANON);
  for (unsigned i = lg_n; i < FR_TWO_ADICITY; ++i) Fr_sqr(&w, &w);
  Fr g; fr_from_u64(&g, FR_GENERATOR);
  if (direction == 1) Fr_inv(&w, &w);
  int in_rev = (order == 2 || order == 3), out_rev = (order == 1 || order == 3);
  if (in_rev) bitrev_permute(x, n, lg_n);                       /* bring input to natural order */
  if (direction == 0 && type == 1) distribute_powers(x, n, &g);
  /* iterative radix-2 DIT on bit-reversed input -> natural output */
  bitrev_permute(x, n, lg_n);
  Fr* roots = (Fr*)malloc(sizeof(Fr) * (n / 2 ? n / 2 : 1));
  if (n >= 2) { roots[0] = Fr_R1; for (size_t i = 1; i < n / 2; ++i) Fr_mul(&roots[i], &roots[i - 1], &w); }
  for (size_t len = 2; len <= n; len <<= 1) {
    size_t half = len / 2, step = n / len;
    for (size_t s = 0; s < n; s += len)
      for (size_t k = 0; k < half; ++k) {
        Fr t; Fr_mul(&t, &x[s + k + half], &roots[k * step]);
        Fr u = x[s + k];
        Fr_add(&x[s + k], &u, &t); Fr_sub(&x[s + k + half], &u, &t);
      }
  }
  free(roots);
  if (direction == 1) {
    Fr ninv, nn; fr_from_u64(&nn, (u64)n); Fr_inv(&ninv, &nn);
    for (size_t i = 0; i < n; ++i) Fr_mul(&x[i], &x[i], &ninv);
    if (type == 1) { Fr gi; Fr_inv(&gi, &g); distribute_powers(x, n, &gi); }
  }
  if (out_rev) bitrev_permute(x, n, lg_n);
  return 0;
}

/* The same transform on `threads` host threads (EvaluationDomain's rayon-parallel butterflies, restated with a barrier per stage): every stage's
 * n/2 butterflies, the powers, the permutations and the scalings are cut into contiguous ranges, one per thread.  Same results as oracle_ntt_fr
 * (tests/test_oracle.py); used for the all-cores CPU baseline of the proof schedule. */
typedef struct { Fr* x; Fr* roots; size_t n; unsigned lg; int order, direction, type, threads, id; Fr w, g; pthread_barrier_t* bar; } ntt_job;
static void fr_pow_u64(Fr* r, const Fr* a, u64 e) { Fr acc = Fr_R1, b = *a; while (e) { if (e & 1) Fr_mul(&acc, &acc, &b); Fr_sqr(&b, &b); e >>= 1; } *r = acc; }
static void bitrev_range(Fr* x, size_t lo, size_t hi, unsigned lg) {
  for (size_t i = lo; i < hi; ++i) {
    size_t j = 0; for (unsigned b = 0; b < lg; ++b) j |= ((i >> b) & 1) << (lg - 1 - b);
    if (i < j) { Fr t = x[i]; x[i] = x[j]; x[j] = t; }                /* the pair (i, j) is touched only by the owner of min(i, j) */
  }
}
static void powers_range(Fr* x, size_t lo, size_t hi, const Fr* g) {     /* x[i] *= g^i */
  Fr pw; fr_pow_u64(&pw, g, (u64)lo);
  for (size_t i = lo; i < hi; ++i) { Fr_mul(&x[i], &x[i], &pw); Fr_mul(&pw, &pw, g); }
}
static void* ntt_worker(void* arg) {
  ntt_job* J = (ntt_job*)arg; const size_t n = J->n, T = (size_t)J->threads, id = (size_t)J->id; Fr* x = J->x;
  const size_t lo = n * id / T, hi = n * (id + 1) / T, hlo = (n / 2) * id / T, hhi = (n / 2) * (id + 1) / T;
  const int in_rev = (J->order == 2 || J->order == 3), out_rev = (J->order == 1 || J->order == 3);
  if (in_rev) { bitrev_range(x, lo, hi, J->lg); pthread_barrier_wait(J->bar); }
  if (J->direction == 0 && J->type == 1) { powers_range(x, lo, hi, &J->g); pthread_barrier_wait(J->bar); }
  bitrev_range(x, lo, hi, J->lg);
  { Fr pw; fr_pow_u64(&pw, &J->w, (u64)hlo); for (size_t i = hlo; i < hhi; ++i) { J->roots[i] = pw; Fr_mul(&pw, &pw, &J->w); } }
  pthread_barrier_wait(J->bar);
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len / 2, step = n / len;
    for (size_t b = hlo; b < hhi; ++b) {
      const size_t s = (b / half) * len, k = b % half;
      Fr t; Fr_mul(&t, &x[s + k + half], &J->roots[k * step]);
      Fr u = x[s + k];
      Fr_add(&x[s + k], &u, &t); Fr_sub(&x[s + k + half], &u, &t);
    }
    pthread_barrier_wait(J->bar);
  }
  if (J->direction == 1) {
    Fr ninv, nn; fr_from_u64(&nn, (u64)n); Fr_inv(&ninv, &nn);
    for (size_t i = lo; i < hi; ++i) Fr_mul(&x[i], &x[i], &ninv);
    if (J->type == 1) { Fr gi; Fr_inv(&gi, &J->g); powers_range(x, lo, hi, &gi); }
    pthread_barrier_wait(J->bar);
  }
  if (out_rev) bitrev_range(x, lo, hi, J->lg);
  return NULL;
}
int oracle_ntt_fr_mt(void* inout, unsigned lg_n, int order, int direction, int type, int threads) {
  if (lg_n > FR_TWO_ADICITY) return -1;
  const size_t n = (size_t)1 << lg_n;
  if (threads > (int)(n / 2)) threads = (int)(n / 2);
  if (threads <= 1) return oracle_ntt_fr(inout, lg_n, order, direction, type);
  Fr w; Fr_to_mont(&w, &FR_ROOT_CANON);
  for (unsigned i = lg_n; i < FR_TWO_ADICITY; ++i) Fr_sqr(&w, &w);
  if (direction == 1) Fr_inv(&w, &w);
  Fr* roots = (Fr*)malloc(sizeof(Fr) * (n / 2));
  pthread_barrier_t bar; pthread_barrier_init(&bar, NULL, (unsigned)threads);
  ntt_job* jobs = (ntt_job*)malloc(sizeof(ntt_job) * threads); pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  for (int t = 0; t < threads; ++t) {
    ntt_job* J = &jobs[t]; J->x = (Fr*)inout; J->roots = roots; J->n = n; J->lg = lg_n; J->order = order; J->direction = direction; J->type = type;
    J->threads = threads; J->id = t; J->w = w; fr_from_u64(&J->g, FR_GENERATOR); J->bar = &bar;
    pthread_create(&th[t], NULL, ntt_worker, J);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
  pthread_barrier_destroy(&bar); free(jobs); free(th); free(roots);
  return 0;
}

/* KZG10::commit shape (polycommit/kzg10): coefficients arrive in Montgomery form, are converted to canonical
 * bigints, and fed to VariableBase::msm over powers_of_beta_g[..len]; result -> affine. */
int oracle_kzg_commit(void* out104, const void* bases104, const void* coeffs_mont, size_t n, int threads) {
  Fr* c = (Fr*)malloc(sizeof(Fr) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) Fr_from_mont(&c[i], (const Fr*)coeffs_mont + i);
  G1Proj t; int rc = oracle_msm_g1(&t, bases104, 104, c, n, threads, 1);
  free(c); if (rc) return rc;
  oracle_g1_to_affine(out104, &t); return 0;
}

/* snarkvm_fields::batch_inversion semantics (fields/src/lib.rs [UPSTREAM-RECALL]): every non-zero element is replaced
 * by its inverse, zeros stay zero.  Element-wise add/sub/mul of Montgomery Fr vectors (Evaluations::*_assign). */
void oracle_fr_batch_inverse(void* io, size_t n) {
  Fr* p = (Fr*)io;
  for (size_t i = 0; i < n; ++i) if (!Fr_is_zero(&p[i])) Fr_inv(&p[i], &p[i]);
}
/* y = M x, CSR (row_ptr u32[rows+1], col u32[nnz], vals Fr mont): the matrix-row inner products of the Varuna prover. */
void oracle_fr_spmv(void* y, const uint32_t* row_ptr, const uint32_t* col, const void* vals, const void* x, size_t rows) {
  for (size_t r = 0; r < rows; ++r) {
    Fr acc; memset(&acc, 0, sizeof acc);
    for (uint32_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) { Fr t; Fr_mul(&t, (const Fr*)vals + k, (const Fr*)x + col[k]); Fr_add(&acc, &acc, &t); }
    ((Fr*)y)[r] = acc;
  }
}
void oracle_fr_vec_op(void* r, const void* a, const void* b, size_t n, int op) {
  for (size_t i = 0; i < n; ++i) {
    if (op == 0) Fr_mul((Fr*)r + i, (const Fr*)a + i, (const Fr*)b + i);
    else if (op == 1) Fr_add((Fr*)r + i, (const Fr*)a + i, (const Fr*)b + i);
    else Fr_sub((Fr*)r + i, (const Fr*)a + i, (const Fr*)b + i);
  }
}

/* KZG10 witness polynomial (polycommit/kzg10 compute_witness_polynomial [UPSTREAM-RECALL]): w(X) = (p(X) - p(z)) / (X - z) by
 * synthetic division, coefficients Montgomery, low degree first.  q receives n - 1 coefficients, *eval = p(z). */
void oracle_fr_divide_by_linear(void* q, void* eval, const void* p, size_t n, const void* z) {
  Fr s; memset(&s, 0, sizeof s);
  for (size_t j = n; j-- > 0;) {
    Fr t; Fr_mul(&t, &s, (const Fr*)z); Fr_add(&s, &t, (const Fr*)p + j);      /* s_j = p_j + z s_(j+1) */
    if (j) ((Fr*)q)[j - 1] = s; else *(Fr*)eval = s;
  }
  if (n == 0) *(Fr*)eval = s;
}

/* ------------------------------------------------------------------------------------------------
 * G2 (curves/src/bls12_377/{fq2,g2}.rs + the same short_weierstrass_jacobian templates [UPSTREAM-RECALL]):
 * Fq2 = Fq[u] / (u^2 + 5)  (NONRESIDUE = -5), y^2 = x^3 + B with B = (0, 155198...874906) — constants checked in
 * tests/test_oracle.py (generator on the curve, r * G2 = O).  VariableBase::msm sends every curve except BLS12-377 G1 to
 * standard::msm, restated below for G2Affine = {x: Fq2, y: Fq2, infinity: bool} (200-byte stride), result Jacobian (288 bytes).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { Fq c0, c1; } Fq2;
typedef struct { Fq2 x, y; uint8_t infinity; uint8_t pad[7]; } G2Affine;   /* sizeof == 200 */
typedef struct { Fq2 x, y, z; } G2Proj;

static inline int Fq2_is_zero(const Fq2* a) { return Fq_is_zero(&a->c0) && Fq_is_zero(&a->c1); }
static inline int Fq2_eq(const Fq2* a, const Fq2* b) { return Fq_eq(&a->c0, &b->c0) && Fq_eq(&a->c1, &b->c1); }
static inline void Fq2_add(Fq2* r, const Fq2* a, const Fq2* b) { Fq_add(&r->c0, &a->c0, &b->c0); Fq_add(&r->c1, &a->c1, &b->c1); }
static inline void Fq2_sub(Fq2* r, const Fq2* a, const Fq2* b) { Fq_sub(&r->c0, &a->c0, &b->c0); Fq_sub(&r->c1, &a->c1, &b->c1); }
static inline void Fq2_dbl(Fq2* r, const Fq2* a) { Fq2_add(r, a, a); }
static inline void Fq2_neg(Fq2* r, const Fq2* a) { Fq_neg(&r->c0, &a->c0); Fq_neg(&r->c1, &a->c1); }
static inline void Fq_mul5(Fq* r, const Fq* a) { Fq t; Fq_dbl(&t, a); Fq_dbl(&t, &t); Fq_add(r, &t, a); }
static void Fq2_mul(Fq2* r, const Fq2* a, const Fq2* b) {              /* schoolbook: (a0 b0 - 5 a1 b1, a0 b1 + a1 b0) */
  Fq v0, v1, t0, t1, n5; Fq_mul(&v0, &a->c0, &b->c0); Fq_mul(&v1, &a->c1, &b->c1);
  Fq_mul(&t0, &a->c0, &b->c1); Fq_mul(&t1, &a->c1, &b->c0);
  Fq_mul5(&n5, &v1); Fq_sub(&r->c0, &v0, &n5); Fq_add(&r->c1, &t0, &t1);
}
static inline void Fq2_sqr(Fq2* r, const Fq2* a) { Fq2 t = *a; Fq2_mul(r, &t, &t); }
static void Fq2_inv(Fq2* r, const Fq2* a) {                             /* (a0 - a1 u) / (a0^2 + 5 a1^2) */
  Fq n, t, s5; Fq_sqr(&n, &a->c0); Fq_sqr(&t, &a->c1); Fq_mul5(&s5, &t); Fq_add(&n, &n, &s5); Fq_inv(&n, &n);
  Fq_mul(&r->c0, &a->c0, &n); Fq_mul(&t, &a->c1, &n); Fq_neg(&r->c1, &t);
}
static const Fq2 FQ2_ONE_INIT = {{{0x02cdffffffffff68ULL, 0x51409f837fffffb1ULL, 0x9f7db3a98a7d3ff2ULL, 0x7b4e97b76e7c6305ULL, 0x4cf495bf803c84e8ULL, 0x008d6661e2fdf49aULL}}, {{0, 0, 0, 0, 0, 0}}};
static inline void g2p_zero(G2Proj* p) { p->x = FQ2_ONE_INIT; p->y = FQ2_ONE_INIT; memset(&p->z, 0, sizeof p->z); }
static inline int g2p_is_zero(const G2Proj* p) { return Fq2_is_zero(&p->z); }
static void g2p_double(G2Proj* p) {                                     /* dbl-2009-l (a = 0) */
  if (g2p_is_zero(p)) return;
  Fq2 A, B, C, D, E, F, t; Fq2_sqr(&A, &p->x); Fq2_sqr(&B, &p->y); Fq2_sqr(&C, &B);
  Fq2_add(&t, &p->x, &B); Fq2_sqr(&t, &t); Fq2_sub(&t, &t, &A); Fq2_sub(&t, &t, &C); Fq2_dbl(&D, &t);
  Fq2_dbl(&E, &A); Fq2_add(&E, &E, &A); Fq2_sqr(&F, &E);
  Fq2 z3; Fq2_mul(&z3, &p->y, &p->z); Fq2_dbl(&z3, &z3);
  Fq2 x3; Fq2_dbl(&t, &D); Fq2_sub(&x3, &F, &t);
  Fq2 c8; Fq2_dbl(&c8, &C); Fq2_dbl(&c8, &c8); Fq2_dbl(&c8, &c8);
  Fq2 y3; Fq2_sub(&t, &D, &x3); Fq2_mul(&y3, &E, &t); Fq2_sub(&y3, &y3, &c8);
  p->x = x3; p->y = y3; p->z = z3;
}
static void g2p_add_mixed(G2Proj* p, const G2Affine* q) {               /* madd-2007-bl */
  if (q->infinity) return;
  if (g2p_is_zero(p)) { p->x = q->x; p->y = q->y; p->z = FQ2_ONE_INIT; return; }
  Fq2 z1z1, u2, s2, t; Fq2_sqr(&z1z1, &p->z); Fq2_mul(&u2, &q->x, &z1z1); Fq2_mul(&t, &q->y, &p->z); Fq2_mul(&s2, &t, &z1z1);
  if (Fq2_eq(&p->x, &u2) && Fq2_eq(&p->y, &s2)) { g2p_double(p); return; }
  Fq2 h, hh, i, j, r, v; Fq2_sub(&h, &u2, &p->x); Fq2_sqr(&hh, &h); Fq2_dbl(&i, &hh); Fq2_dbl(&i, &i); Fq2_mul(&j, &h, &i);
  Fq2_sub(&r, &s2, &p->y); Fq2_dbl(&r, &r); Fq2_mul(&v, &p->x, &i);
  Fq2 x3, y3, z3; Fq2_sqr(&x3, &r); Fq2_sub(&x3, &x3, &j); Fq2_dbl(&t, &v); Fq2_sub(&x3, &x3, &t);
  Fq2_sub(&t, &v, &x3); Fq2_mul(&y3, &r, &t); Fq2_mul(&t, &p->y, &j); Fq2_dbl(&t, &t); Fq2_sub(&y3, &y3, &t);
  Fq2_add(&z3, &p->z, &h); Fq2_sqr(&z3, &z3); Fq2_sub(&z3, &z3, &z1z1); Fq2_sub(&z3, &z3, &hh);
  p->x = x3; p->y = y3; p->z = z3;
}
static void g2p_add(G2Proj* p, const G2Proj* q) {                       /* add-2007-bl */
  if (g2p_is_zero(q)) return;
  if (g2p_is_zero(p)) { *p = *q; return; }
  Fq2 z1z1, z2z2, u1, u2, s1, s2, t; Fq2_sqr(&z1z1, &p->z); Fq2_sqr(&z2z2, &q->z);
  Fq2_mul(&u1, &p->x, &z2z2); Fq2_mul(&u2, &q->x, &z1z1);
  Fq2_mul(&t, &p->y, &q->z); Fq2_mul(&s1, &t, &z2z2); Fq2_mul(&t, &q->y, &p->z); Fq2_mul(&s2, &t, &z1z1);
  if (Fq2_eq(&u1, &u2) && Fq2_eq(&s1, &s2)) { g2p_double(p); return; }
  Fq2 h, i, j, r, v; Fq2_sub(&h, &u2, &u1); Fq2_dbl(&i, &h); Fq2_sqr(&i, &i); Fq2_mul(&j, &h, &i);
  Fq2_sub(&r, &s2, &s1); Fq2_dbl(&r, &r); Fq2_mul(&v, &u1, &i);
  Fq2 x3, y3, z3; Fq2_sqr(&x3, &r); Fq2_sub(&x3, &x3, &j); Fq2_dbl(&t, &v); Fq2_sub(&x3, &x3, &t);
  Fq2_sub(&t, &v, &x3); Fq2_mul(&y3, &r, &t); Fq2_mul(&t, &s1, &j); Fq2_dbl(&t, &t); Fq2_sub(&y3, &y3, &t);
  Fq2_add(&z3, &p->z, &q->z); Fq2_sqr(&z3, &z3); Fq2_sub(&z3, &z3, &z1z1); Fq2_sub(&z3, &z3, &z2z2); Fq2_mul(&z3, &z3, &h);
  p->x = x3; p->y = y3; p->z = z3;
}
static void g2p_to_affine(G2Affine* a, const G2Proj* p) {
  memset(a, 0, sizeof *a);
  if (g2p_is_zero(p)) { a->y = FQ2_ONE_INIT; a->infinity = 1; return; }
  Fq2 zi, zi2, zi3; Fq2_inv(&zi, &p->z); Fq2_sqr(&zi2, &zi); Fq2_mul(&zi3, &zi2, &zi);
  Fq2_mul(&a->x, &p->x, &zi2); Fq2_mul(&a->y, &p->y, &zi3);
}
static void g2_load(G2Affine* b, const uint8_t* src, size_t stride) { memset(b, 0, sizeof *b); memcpy(b, src, 192); b->infinity = (stride == 200) ? (src[192] != 0) : 0; }

/* standard::msm over G2 (windows run one after the other; `threads` is accepted for symmetry and ignored) */
int oracle_msm_g2(void* out, const void* bases_, size_t stride, const void* scalars_, size_t n, int threads) {
  (void)threads;
  if (stride != 200 && stride != 192) return -1;
  G2Affine* bases = (G2Affine*)malloc(sizeof(G2Affine) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) g2_load(&bases[i], (const uint8_t*)bases_ + i * stride, stride);
  const u64* scalars = (const u64*)scalars_;
  unsigned c = n < 32 ? 3 : ln_without_floats(n) + 2;
  int nw = (FR_BITS + c - 1) / c; size_t nb = ((size_t)1 << c) - 1;
  G2Proj* win = (G2Proj*)malloc(sizeof(G2Proj) * nw); G2Proj* buckets = (G2Proj*)malloc(sizeof(G2Proj) * nb);
  for (int w = 0; w < nw; ++w) {
    G2Proj res; g2p_zero(&res);
    for (size_t i = 0; i < nb; ++i) g2p_zero(&buckets[i]);
    for (size_t i = 0; i < n; ++i) {
      const u64* s = scalars + 4 * i;
      if (scalar_is_zero(s)) continue;
      if (scalar_is_one(s)) { if (w == 0) g2p_add_mixed(&res, &bases[i]); continue; }
      u64 d = scalar_window(s, w * c, c);
      if (d != 0) g2p_add_mixed(&buckets[d - 1], &bases[i]);
    }
    G2Proj running; g2p_zero(&running);
    for (size_t b = nb; b-- > 0;) { g2p_add(&running, &buckets[b]); g2p_add(&res, &running); }
    win[w] = res;
  }
  G2Proj total; g2p_zero(&total);
  for (int w = nw - 1; w >= 1; --w) { g2p_add(&total, &win[w]); for (unsigned d = 0; d < c; ++d) g2p_double(&total); }
  g2p_add(&total, &win[0]);
  memcpy(out, &total, sizeof total);
  free(win); free(buckets); free(bases);
  return 0;
}
void oracle_g2_to_affine(void* out200, const void* jac288) { G2Proj p; memcpy(&p, jac288, sizeof p); G2Affine a; g2p_to_affine(&a, &p); memcpy(out200, &a, 200); }
/* P_i = (i + 1) * base by repeated mixed addition, affine out (200-byte rows) */
void oracle_g2_multiples(void* out200, const void* base200, size_t n) {
  G2Affine g; g2_load(&g, (const uint8_t*)base200, 200);
  G2Proj run; g2p_zero(&run);
  for (size_t i = 0; i < n; ++i) { g2p_add_mixed(&run, &g); G2Affine a; g2p_to_affine(&a, &run); memcpy((uint8_t*)out200 + 200 * i, &a, 200); }
}
