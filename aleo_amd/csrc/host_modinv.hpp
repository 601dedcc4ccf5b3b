// host_modinv.hpp — modular inversion on the host by Bernstein-Yang divsteps ("safegcd", variable time), 62 divsteps per outer iteration on signed 62-bit
// limbs — ~1-2 us for Fq (377 bits) against ~20 us for the Fermat chain a^(q-2) (570 Montgomery products) that host_field.hpp used through round 4.
// The inversion sits on the critical path of every commitment's host tail (affine normalisation of the k results: one shared inversion) and of the round
// constants (1 / alpha, 1 / beta, the openings' denominators); replaces the same work snarkVM does with Field::inverse (fields/src/fp_384.rs, fp_256.rs:
// a binary extended Euclid there) [UPSTREAM-RECALL].  Algorithm: D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and modular inversion"
// (2019), in the batched 62-bit form that is public knowledge from several libraries' modinv64; written here from the paper's recurrences:
//   divstep(delta, f, g) = (1 - delta, g, (g - f) / 2)          if delta > 0 and g odd
//                          (1 + delta, f, (g + (g mod 2) f) / 2)  otherwise
// with eta = -delta, 62 steps at a time on the low limbs giving a 2 x 2 transition matrix t (entries < 2^62 in size), then (f, g) <- t (f, g) / 2^62 exactly and
// (d, e) <- t (d, e) / 2^62 mod p.  Starting from (f, g, d, e) = (p, x, 0, 1), g reaches 0 with f = +-1 and d = +-x^-1.
// The result is checked against the Fermat chain in tests/test_host_field.py (through aleo_mi355x_selftest_host_inverse) — same bytes, including 0 -> 0.
#pragma once
#include <cstdint>
#include <cstring>

namespace aleo_mi355x { namespace host {

template <int N> struct ModInv {                           // N 64-bit limbs of modulus; L = limbs of 62 bits that hold 64 N bits + sign
  static constexpr int L = (64 * N + 61) / 62 + 1 > 7 ? 7 : (64 * N + 61) / 62 + 1;      // N = 4: 6 (only 5 needed: the top one stays 0 / -1), N = 6: 7
  typedef __int128 i128;
  struct S62 { int64_t v[L]; };
  static constexpr int64_t M62 = (int64_t)(~0ull >> 2);

  static S62 from_limbs(const uint64_t* a) {               // 64-bit limbs -> 62-bit limbs (non-negative)
    S62 r; std::memset(r.v, 0, sizeof r.v);
    for (int i = 0; i < L; ++i) {
      const int bit = 62 * i, w = bit >> 6, sh = bit & 63;
      if (w >= N) break;
      uint64_t x = a[w] >> sh;
      if (sh > 2 && w + 1 < N) x |= a[w + 1] << (64 - sh);
      r.v[i] = (int64_t)(x & (uint64_t)M62);
    }
    return r;
  }
  static void to_limbs(uint64_t* out, const S62& a) {      // a in [0, 2^(64 N))
    std::memset(out, 0, 8 * N);
    for (int i = 0; i < L; ++i) {
      const int bit = 62 * i, w = bit >> 6, sh = bit & 63;
      if (w >= N) break;
      out[w] |= (uint64_t)a.v[i] << sh;
      if (sh > 2 && w + 1 < N) out[w + 1] |= (uint64_t)a.v[i] >> (64 - sh);
    }
  }
  struct T2 { int64_t u, v, q, r; };

  // 62 divsteps on the low 64 bits of f and g (f odd); returns the new eta and the transition matrix scaled by 2^62
  static int64_t divsteps62(int64_t eta, uint64_t f0, uint64_t g0, T2* t) {
    uint64_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0, m; uint32_t w; int i = 62, limit, zeros;
    for (;;) {
      zeros = __builtin_ctzll(g | (~0ull << i));           // strip the zero bits of g (at most i of them)
      g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
      if (i == 0) break;
      if (eta < 0) {                                       // delta > 0, g odd: swap
        uint64_t tmp;
        eta = -eta;
        tmp = f; f = g; g = (uint64_t)(-(int64_t)tmp);
        tmp = u; u = q; q = (uint64_t)(-(int64_t)tmp);
        tmp = v; v = r; r = (uint64_t)(-(int64_t)tmp);
        limit = ((int)eta + 1) > i ? i : ((int)eta + 1);   // up to 6 bits of g cancelled at once
        m = (~0ull >> (64 - limit)) & 63u;
        w = (uint32_t)((f * g * (f * f - 2)) & m);           // w = -g / f mod 2^6 (f * (f^2 - 2) is -1/f mod 2^6 for odd f)
      } else {
        limit = ((int)eta + 1) > i ? i : ((int)eta + 1);   // up to 4 bits at once
        m = (~0ull >> (64 - limit)) & 15u;
        w = (uint32_t)(f + (((f + 1) & 4) << 1));            // 1 / f mod 2^4 (negated below)
        w = (uint32_t)((-(uint64_t)w * g) & m);
      }
      g += f * w; q += u * w; r += v * w;
    }
    t->u = (int64_t)u; t->v = (int64_t)v; t->q = (int64_t)q; t->r = (int64_t)r;
    return eta;
  }
  // (d, e) <- t (d, e) / 2^62 mod p, with d, e kept in (-2p, p)
  static void update_de(S62* d, S62* e, const T2* t, const S62& mod, uint64_t mod_inv62) {
    const int64_t u = t->u, v = t->v, q = t->q, r = t->r;
    int64_t di, ei, md, me, sd, se; i128 cd, ce;
    sd = d->v[L - 1] >> 63; se = e->v[L - 1] >> 63;        // -1 for a negative value
    md = (u & sd) + (v & se); me = (q & sd) + (r & se);    // multiples of the modulus that bring negative inputs back
    di = d->v[0]; ei = e->v[0];
    cd = (i128)u * di + (i128)v * ei; ce = (i128)q * di + (i128)r * ei;
    md -= (int64_t)((mod_inv62 * (uint64_t)cd + (uint64_t)md) & (uint64_t)M62);      // make the low 62 bits of t (d, e) + mod (md, me) vanish
    me -= (int64_t)((mod_inv62 * (uint64_t)ce + (uint64_t)me) & (uint64_t)M62);
    cd += (i128)mod.v[0] * md; ce += (i128)mod.v[0] * me;
    cd >>= 62; ce >>= 62;
    for (int i = 1; i < L; ++i) {
      di = d->v[i]; ei = e->v[i];
      cd += (i128)u * di + (i128)v * ei; ce += (i128)q * di + (i128)r * ei;
      cd += (i128)mod.v[i] * md; ce += (i128)mod.v[i] * me;
      d->v[i - 1] = (int64_t)cd & M62; cd >>= 62;
      e->v[i - 1] = (int64_t)ce & M62; ce >>= 62;
    }
    d->v[L - 1] = (int64_t)cd; e->v[L - 1] = (int64_t)ce;
  }
  // (f, g) <- t (f, g) / 2^62 (exact) on the first len limbs
  static void update_fg(int len, S62* f, S62* g, const T2* t) {
    const int64_t u = t->u, v = t->v, q = t->q, r = t->r;
    int64_t fi = f->v[0], gi = g->v[0]; i128 cf, cg;
    cf = (i128)u * fi + (i128)v * gi; cg = (i128)q * fi + (i128)r * gi;
    cf >>= 62; cg >>= 62;                                  // the low 62 bits are zero by construction
    for (int i = 1; i < len; ++i) {
      fi = f->v[i]; gi = g->v[i];
      cf += (i128)u * fi + (i128)v * gi; cg += (i128)q * fi + (i128)r * gi;
      f->v[i - 1] = (int64_t)cf & M62; cf >>= 62;
      g->v[i - 1] = (int64_t)cg & M62; cg >>= 62;
    }
    f->v[len - 1] = (int64_t)cf; g->v[len - 1] = (int64_t)cg;
  }
  // r in (-2p, p) -> [0, p), negated first when `sign` < 0
  static void normalize(S62* r, int64_t sign, const S62& mod) {
    int64_t c[L]; std::memcpy(c, r->v, sizeof c);
    int64_t cond_add = c[L - 1] >> 63;                     // negative: + p
    for (int i = 0; i < L; ++i) c[i] += mod.v[i] & cond_add;
    const int64_t cond_neg = sign >> 63;
    for (int i = 0; i < L; ++i) c[i] = (c[i] ^ cond_neg) - cond_neg;
    for (int i = 0; i + 1 < L; ++i) { c[i + 1] += c[i] >> 62; c[i] &= M62; }
    cond_add = c[L - 1] >> 63;
    for (int i = 0; i < L; ++i) c[i] += mod.v[i] & cond_add;
    for (int i = 0; i + 1 < L; ++i) { c[i + 1] += c[i] >> 62; c[i] &= M62; }
    std::memcpy(r->v, c, sizeof c);
  }

  // out = x^-1 mod p as 64-bit limbs (plain integers, not Montgomery); x in [0, p); 0 -> 0
  static void inverse(uint64_t* out, const uint64_t* x, const uint64_t* p) {
    const S62 mod = from_limbs(p);
    uint64_t inv = 1, p0 = (uint64_t)mod.v[0];             // p^-1 mod 2^62 (Newton: doubles the correct bits each step)
    for (int i = 0; i < 6; ++i) inv *= 2 - p0 * inv;
    inv &= (uint64_t)M62;
    S62 d, e, f = mod, g = from_limbs(x); std::memset(d.v, 0, sizeof d.v); std::memset(e.v, 0, sizeof e.v); e.v[0] = 1;
    int len = L; int64_t eta = -1;
    for (int iter = 0; iter < 64; ++iter) {                // 1110 divsteps bound a 384-bit inversion: 18 rounds; the loop leaves when g = 0
      T2 t; eta = divsteps62(eta, (uint64_t)f.v[0], (uint64_t)g.v[0], &t);
      update_de(&d, &e, &t, mod, inv);
      update_fg(len, &f, &g, &t);
      if (g.v[0] == 0) { int64_t c = 0; for (int j = 1; j < len; ++j) c |= g.v[j]; if (c == 0) break; }
      const int64_t fn = f.v[len - 1], gn = g.v[len - 1];    // drop a top limb that carries only the sign of both
      int64_t c = ((int64_t)len - 2) >> 63;
      c |= fn ^ (fn >> 63); c |= gn ^ (gn >> 63);
      if (c == 0) { f.v[len - 2] |= (int64_t)((uint64_t)fn << 62); g.v[len - 2] |= (int64_t)((uint64_t)gn << 62); --len; }
    }
    normalize(&d, f.v[len - 1], mod);                      // f = +-1: d = +-x^-1
    to_limbs(out, d);
  }
};

}}  // namespace aleo_mi355x::host
