// ntt.hip — radix-2 number-theoretic transform over the BLS12-377 scalar field Fr for MI355X (gfx950).
//
// Replaces snarkvm-algorithms 0.14.5  algorithms/src/fft/domain.rs  EvaluationDomain::<Fr>::
// {fft,ifft,coset_fft,coset_ifft}_in_place  [UPSTREAM-RECALL; pin /root/reference/Cargo.lock:2200], the second
// operator of Varuna's prover.  Same function: out[k] = sum_j x[j] * w^(jk) with
// w = TWO_ADIC_ROOT_OF_UNITY^(2^(47-lg_n)), natural order in and out; coset variants shift by g = 22;
// the inverse multiplies by n^-1.  The schedule is GPU-first:
//
//   n = n1*n2*n3 (each <= 2^10).  Pass i runs all length-n_i DFTs of its axis inside LDS (one HBM read and one
//   HBM write of the data per pass: 2-3 passes instead of lg_n), as radix-2 DIF butterflies over limb-planar
//   LDS tiles (conflict-free 4-byte accesses), followed by the inter-pass twiddle w^(k*j_rest) taken from a
//   two-level table (hi*lo).  Tiles are T adjacent transforms wide so every HBM access is a >=128-byte run; the
//   last pass writes straight to the natural-order position (index-transposed store), so there is no separate
//   bit-reversal pass.  Values stay lazily reduced in [0, 2r) (fp.h) and are made canonical at the final store.
//
// HBM traffic: 64 bytes per element per pass (read + write), algorithmic minimum 64 bytes per element.
#include "ctx.h"
#include "fp.h"
#include "fr29.h"
#include "host_field.hpp"
#include <cstdlib>

namespace aleo_mi355x {

// LDS tile: TE elements as 8 limb planes (TE x 32 B).  TE = 2048 (64 KiB, 256 threads, 2 blocks per CU) for small
// transforms; TE = 4096 (128 KiB of the CU's 160 KiB, 512 threads) lets 2^19..2^22 run in TWO passes (2^11 x 2^11)
// instead of three: one inter-pass twiddle product and one HBM round trip less per element.
static constexpr uint32_t INNER_MAX_LG = 11;        // longest in-LDS transform

struct NttTables {
  uint32_t lg_n = 0, lo_bits = 0;
  void* d_inner = nullptr;     // w_{2048}^t, t < 1024            (inner butterflies; shorter transforms stride it)
  void* d_tw_hi = nullptr;     // w_n^(e_hi << lo_bits)
  void* d_tw_lo = nullptr;     // w_n^(e_lo)
  void* d_cs_hi = nullptr;     // coset powers: g^(j_hi << lo_bits)        (inverse: g^-(...))
  void* d_cs_lo = nullptr;     // g^(j_lo)                                  (inverse: g^-(j_lo) * n^-1)
  void* d_direct = nullptr;    // two-pass sizes: w_n^(k * b) at the position (k << lgBn) + b of the element it multiplies after pass 1
  uint32_t direct_lgBn = 0;    // the pass split d_direct was built for
  uint32_t scale[8];           // n^-1 (Montgomery): applied at the final store of a plain inverse transform
  // the same tables for the 29-bit-limb kernels (fr29.h): every entry times 2^5, i.e. in the Montgomery form of R = 2^261; packed 32-byte numbers
  // except the inner twiddles (9 limbs at a 48-byte stride)
  void *d_inner29 = nullptr, *d_tw_hi29 = nullptr, *d_tw_lo29 = nullptr, *d_cs_hi29 = nullptr, *d_cs_lo29 = nullptr, *d_direct29 = nullptr;
  uint32_t direct29_lgBn = 0;
  uint32_t scale29[9];         // n^-1 in that form, 29-bit limbs
};

struct FrArg { uint32_t v[8]; };
// Out-of-place, zero-padded input of a transform's FIRST pass (ntt_run_from): transform b of the batch reads element i from p + (b * stride + i) * 32 when
// i < len and takes 0 otherwise; p == nullptr = the ordinary in-place call.  (A prover round pads |H| coefficients to 4|H| and |K| to 2|K| before it
// evaluates: a fill and one copy per polynomial in rounds 1-4 — launches of ~4.5 us each at the sizes of real circuits.)
struct NttSrc { const char* p = nullptr; uint32_t len = 0, stride = 0; };

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// planar LDS tile: limb l of element e at lds[l * TE + sw(e)].  sw() XORs the low five index bits with the next five: a bijection
// that leaves runs of consecutive elements conflict-free and spreads the stride-4 / stride-8 accesses of the last register groups
// (elements base + j * q with q = 1, 4: every lane of a wave used to hit the same 4-8 of the 32 banks) over the banks.
__device__ __forceinline__ uint32_t sw(uint32_t e) { return e ^ ((e >> 5) & 31u); }
template <uint32_t TE> __device__ __forceinline__ Fr lds_load(const uint32_t* lds, uint32_t e) {
  Fr r;
  const uint32_t x = sw(e);
#pragma unroll
  for (int l = 0; l < 8; ++l) r.v[l] = lds[l * TE + x];
  return r;
}
template <uint32_t TE> __device__ __forceinline__ void lds_store(uint32_t* lds, uint32_t e, const Fr& a) {
  const uint32_t x = sw(e);
#pragma unroll
  for (int l = 0; l < 8; ++l) lds[l * TE + x] = a.v[l];
}

// w^e from the two-level table (one product, result < 2r)
__device__ __forceinline__ Fr two_level(const char* hi, const char* lo, uint32_t e, uint32_t lo_bits) {
  Fr a = load_fp<Fr>(hi + (size_t)(e >> lo_bits) * 32), b = load_fp<Fr>(lo + (size_t)(e & ((1u << lo_bits) - 1u)) * 32);
  return Fr::mul(a, b);
}

// G consecutive radix-2 DIF stages (first one = stage s) on 2^G elements held in registers: one LDS read and one LDS
// write per element per GROUP instead of per stage, a third of the barriers, 7 twiddle loads per 8 elements per 3
// stages instead of 12.  Element j of the super-butterfly sits at base + j*q, q = L >> (s + G).
template <uint32_t TE, int G> __device__ __forceinline__ void dif_group(uint32_t* lds, uint32_t lgL, uint32_t s, uint32_t sb, const char* __restrict__ inner) {
  constexpr int K = 1 << G;
  const uint32_t lgq = lgL - s - G, q = 1u << lgq, per_row_lg = lgL - G;
  const uint32_t row = sb >> per_row_lg, w = sb & ((1u << per_row_lg) - 1u);
  const uint32_t grp = w >> lgq, pos = w & (q - 1u);
  const uint32_t base = (row << lgL) + (grp << (lgq + G)) + pos;
  Fr v[K];
#pragma unroll
  for (int j = 0; j < K; ++j) v[j] = lds_load<TE>(lds, base + ((uint32_t)j << lgq));
#pragma unroll
  for (int t = 0; t < G; ++t) {
    constexpr int dummy = 0; (void)dummy;
    const int d = 1 << (G - 1 - t);                               // partner distance in units of q
    const uint32_t lgh = lgq + (uint32_t)(G - 1 - t);             // lg of the butterfly half-size in elements
#pragma unroll
    for (int r = 0; r < d; ++r) {                                 // distinct twiddles of this stage
      Fr tw;
      if (lgh) tw = load_fp<Fr>(inner + (size_t)((((uint32_t)r << lgq) + pos) << (INNER_MAX_LG - 1 - lgh)) * 32);
#pragma unroll
      for (int blk = 0; blk < K / (2 * d); ++blk) {
        const int lo = blk * 2 * d + r, hi = lo + d;
        Fr u = v[lo], x = v[hi];
        v[lo] = Fr::cond_sub<2>(Fr::add(u, x));                   // < 4r -> < 2r
        Fr dif = Fr::sub<2>(u, x);                                // u + 2r - x < 4r
        v[hi] = lgh ? Fr::mul(dif, tw) : Fr::cond_sub<2>(dif);    // 4*1/13.7 + 1 -> < 2r
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) lds_store<TE>(lds, base + ((uint32_t)j << lgq), v[j]);
}

// All radix-2 DIF stages of T length-L transforms held in the tile (row t at [t*L, (t+1)*L)), in register groups of 3
// stages (then 2 or 1).  Output k of row t ends at position t*L + bitrev(k).  Values in and out are < 2r.
// GM = 1 (the latency form for a lone small transform): one stage per LDS round trip, one butterfly per lane — a 512-element tile keeps 256 lanes busy
// for 9 short steps instead of 64 lanes for 3 long ones (12 dependent products each).
template <uint32_t TE, uint32_t NT, int GM = 3> __device__ __forceinline__ void tile_dif(uint32_t* lds, uint32_t lgL, uint32_t T, const char* __restrict__ inner) {
  const uint32_t total = T << lgL;
  if constexpr (GM == 1) {
    for (uint32_t st = 0; st < lgL; ++st) {
      for (uint32_t sb = threadIdx.x; sb < (total >> 1); sb += NT) dif_group<TE, 1>(lds, lgL, st, sb, inner);
      __syncthreads();
    }
    return;
  }
  uint32_t s = 0;
  for (; s + 3 <= lgL; s += 3) {
    for (uint32_t sb = threadIdx.x; sb < (total >> 3); sb += NT) dif_group<TE, 3>(lds, lgL, s, sb, inner);
    __syncthreads();
  }
  if (lgL - s == 2) {
    for (uint32_t sb = threadIdx.x; sb < (total >> 2); sb += NT) dif_group<TE, 2>(lds, lgL, s, sb, inner);
    __syncthreads();
  } else if (lgL - s == 1) {
    for (uint32_t sb = threadIdx.x; sb < (total >> 1); sb += NT) dif_group<TE, 1>(lds, lgL, s, sb, inner);
    __syncthreads();
  }
}

// Pass over axis l of the view [A][L][Bn] (index = (a*L + l)*Bn + b), tile = one a, T adjacent b.
// dst may alias src (same positions).  After the transform, element (k, b) is multiplied by w_n^(tw_scale*k*b).
template <uint32_t TE, uint32_t NT, int GM = 3>
__global__ void __launch_bounds__(NT) k_ntt_strided(const char* src, char* dst, uint32_t lgL, uint32_t lgBn, uint32_t lgT,
                                                     uint32_t tw_scale, uint32_t lg_n, uint32_t lo_bits, const char* __restrict__ inner,
                                                     const char* __restrict__ tw_hi, const char* __restrict__ tw_lo,
                                                     const char* __restrict__ cs_hi, const char* __restrict__ cs_lo, int pre_coset,
                                                     const char* __restrict__ direct, NttSrc ext) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  src = ext.p ? ext.p + (size_t)blockIdx.y * ext.stride * 32 : src + ((size_t)blockIdx.y << (lg_n + 5)); dst += (size_t)blockIdx.y << (lg_n + 5);      // blockIdx.y: independent transform of a batch
  const uint32_t L = 1u << lgL, T = 1u << lgT, tiles_per_a = 1u << (lgBn - lgT);
  const uint32_t a = blockIdx.x / tiles_per_a, b0 = (blockIdx.x % tiles_per_a) << lgT;
  const uint32_t nq = (T * L) << 1;
  for (uint32_t q = threadIdx.x; q < nq; q += NT) {
    uint32_t elem = q >> 1, hf = q & 1, t = elem & (T - 1u), l = elem >> lgT;
    size_t gi = ((((size_t)a << lgL) + l) << lgBn) + b0 + t;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!ext.p || gi < ext.len) v = *(const uint4*)(src + gi * 32 + hf * 16);
    uint32_t e = t * L + l, p = hf * 4;
    const uint32_t ex = sw(e);
    lds[(p + 0) * TE + ex] = v.x; lds[(p + 1) * TE + ex] = v.y; lds[(p + 2) * TE + ex] = v.z; lds[(p + 3) * TE + ex] = v.w;
  }
  __syncthreads();
  if (pre_coset) {   // coset_fft: x[j] *= g^j before the transform (only the first pass: A == 1, j = l*Bn + b)
    for (uint32_t e = threadIdx.x; e < T * L; e += NT) {
      uint32_t t = e >> lgL, l = e & (L - 1u);
      uint32_t j = (l << lgBn) + b0 + t;
      Fr x = lds_load<TE>(lds, e);
      x = Fr::mul(x, two_level(cs_hi, cs_lo, j, lo_bits));   // 1*2/13.7+1 -> < 2r
      lds_store<TE>(lds, e, x);
    }
    __syncthreads();
  }
  tile_dif<TE, NT, GM>(lds, lgL, T, inner);
  const uint32_t nmask = (lg_n >= 32) ? 0xffffffffu : ((1u << lg_n) - 1u);
  for (uint32_t e = threadIdx.x; e < T * L; e += NT) {
    uint32_t t = e & (T - 1u), k = e >> lgT;
    Fr x = lds_load<TE>(lds, t * L + bitrev(k, lgL));
    uint32_t b = b0 + t;
    size_t gi = ((((size_t)a << lgL) + k) << lgBn) + b;
    if (direct) x = Fr::mul(x, load_fp<Fr>(direct + gi * 32));     // the factor sits where its element goes: one coalesced 32-byte read instead of a product
    else {
      uint32_t ex = (uint32_t)(((uint64_t)tw_scale * k * b) & nmask);
      x = Fr::mul(x, two_level(tw_hi, tw_lo, ex, lo_bits));     // 2*2/13.7+1 -> < 2r
    }
    store_fp<Fr>(dst + gi * 32, x);
  }
}

// Last pass: rows a = k1*n2 + k2 are contiguous (Bn == 1); tile = T adjacent k1 at one k2; output k of that row
// goes to natural position k1 + n1*(k2 + n2*k).  src != dst unless n1 == n2 == 1.
template <uint32_t TE, uint32_t NT, int GM = 3>
__global__ void __launch_bounds__(NT) k_ntt_final(const char* src, char* dst, uint32_t lgL, uint32_t lgN1, uint32_t lgN2, uint32_t lgT,
                                                   uint32_t lo_bits, const char* __restrict__ inner, const char* __restrict__ cs_hi, const char* __restrict__ cs_lo,
                                                   int pre_coset, int post_coset, int do_scale, FrArg scale, NttSrc ext) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  src = ext.p ? ext.p + (size_t)blockIdx.y * ext.stride * 32 : src + ((size_t)blockIdx.y << (lgL + lgN1 + lgN2 + 5)); dst += (size_t)blockIdx.y << (lgL + lgN1 + lgN2 + 5);
  const uint32_t L = 1u << lgL, T = 1u << lgT;
  const uint32_t tiles_k1 = 1u << (lgN1 - lgT);
  const uint32_t k2 = blockIdx.x / tiles_k1, k10 = (blockIdx.x % tiles_k1) << lgT;
  const uint32_t nq = (T * L) << 1;
  for (uint32_t q = threadIdx.x; q < nq; q += NT) {
    uint32_t elem = q >> 1, hf = q & 1, l = elem & (L - 1u), t = elem >> lgL;
    size_t row = ((size_t)(k10 + t) << lgN2) + k2;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!ext.p || (row << lgL) + l < ext.len) v = *(const uint4*)(src + ((row << lgL) + l) * 32 + hf * 16);
    uint32_t e = t * L + l, p = hf * 4;
    const uint32_t ex = sw(e);
    lds[(p + 0) * TE + ex] = v.x; lds[(p + 1) * TE + ex] = v.y; lds[(p + 2) * TE + ex] = v.z; lds[(p + 3) * TE + ex] = v.w;
  }
  __syncthreads();
  if (pre_coset) {   // single-pass coset_fft: j = l
    for (uint32_t e = threadIdx.x; e < T * L; e += NT) {
      uint32_t l = e & (L - 1u);
      Fr x = lds_load<TE>(lds, e);
      x = Fr::mul(x, two_level(cs_hi, cs_lo, l, lo_bits));
      lds_store<TE>(lds, e, x);
    }
    __syncthreads();
  }
  tile_dif<TE, NT, GM>(lds, lgL, T, inner);
  Fr sc; for (int i = 0; i < 8; ++i) sc.v[i] = scale.v[i];
  for (uint32_t e = threadIdx.x; e < T * L; e += NT) {
    uint32_t t = e & (T - 1u), k = e >> lgT;
    Fr x = lds_load<TE>(lds, t * L + bitrev(k, lgL));
    size_t o = (size_t)(k10 + t) + (((size_t)k2 + ((size_t)k << lgN2)) << lgN1);
    if (post_coset) x = Fr::mul(x, two_level(cs_hi, cs_lo, (uint32_t)o, lo_bits));   // g^-o * n^-1
    else if (do_scale) x = Fr::mul(x, sc);
    x = Fr::reduce(x);
    store_fp<Fr>(dst + o * 32, x);
  }
}

// ---- the same passes on 29-bit limbs (fr29.h): tiles of 9 limb planes, butterflies without conditional subtractions ------------------------------
// Bounds inside a register group (dif_group29), for values that enter it normalised and below 4.5 r:
//   sum  lo = u + x            lazy: limbs and value double per stage (8 x at most: limbs <= 2^32 - 8, value < 36 r)
//   diff hi = (u + 19 r - x) w  the padded constant keeps every limb non-negative for x with limbs <= 2^30 - 2 and a value < 18 r; the product takes
//                               a multiplicand with limbs < 2^31.4 and returns a normalised value < (37 / 445 + 1) r
//   before the third stage of a three-stage group registers 0 and 1 (sums of sums: limbs < 2^31) are normalised; at the end the sums of the last
//   stage are normalised and register 0 — the only one that was a sum in every stage — is brought below 3 r (f29_reduce_partial).
//   A stage whose twiddle is 1 (half-size 1) normalises and reduces its differences instead of multiplying them.
// So every value a group stores is normalised and below 4.5 r (the largest: a sum of two sums of products, 4.4 r).
template <uint32_t TE> __device__ __forceinline__ F29 lds_load29(const uint32_t* lds, uint32_t e) {
  F29 r; const uint32_t x = sw(e);
#pragma unroll
  for (int l = 0; l < 9; ++l) r.v[l] = lds[l * TE + x];
  return r;
}
template <uint32_t TE> __device__ __forceinline__ void lds_store29(uint32_t* lds, uint32_t e, const F29& a) {
  const uint32_t x = sw(e);
#pragma unroll
  for (int l = 0; l < 9; ++l) lds[l * TE + x] = a.v[l];
}
__device__ __forceinline__ F29 two_level29(const char* hi, const char* lo, uint32_t e, uint32_t lo_bits) {
  return f29_mul(f29_load_packed(hi + (size_t)(e >> lo_bits) * 32), f29_load_packed(lo + (size_t)(e & ((1u << lo_bits) - 1u)) * 32));
}
template <uint32_t TE, int G> __device__ __forceinline__ void dif_group29(uint32_t* lds, uint32_t lgL, uint32_t s, uint32_t sb, const char* __restrict__ inner) {
  constexpr int K = 1 << G;
  const uint32_t lgq = lgL - s - G, q = 1u << lgq, per_row_lg = lgL - G;
  const uint32_t row = sb >> per_row_lg, w = sb & ((1u << per_row_lg) - 1u);
  const uint32_t grp = w >> lgq, pos = w & (q - 1u);
  const uint32_t base = (row << lgL) + (grp << (lgq + G)) + pos;
  F29 v[K];
#pragma unroll
  for (int j = 0; j < K; ++j) v[j] = lds_load29<TE>(lds, base + ((uint32_t)j << lgq));
#pragma unroll
  for (int t = 0; t < G; ++t) {
    const int d = 1 << (G - 1 - t);
    const uint32_t lgh = lgq + (uint32_t)(G - 1 - t);
    if (G == 3 && t == 2) { f29_normalise(v[0]); f29_normalise(v[1]); }
#pragma unroll
    for (int r = 0; r < d; ++r) {
      F29 tw;
      if (lgh) tw = f29_load48(inner + (size_t)((((uint32_t)r << lgq) + pos) << (INNER_MAX_LG - 1 - lgh)) * 48);
#pragma unroll
      for (int blk = 0; blk < K / (2 * d); ++blk) {
        const int lo = blk * 2 * d + r, hi = lo + d;
        const F29 u = v[lo], x = v[hi];
        v[lo] = f29_add(u, x);
        F29 dif = f29_sub_pad(u, x);
        if (lgh) v[hi] = f29_mul(dif, tw);
        else { f29_normalise(dif); f29_reduce_partial(dif); v[hi] = dif; }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K; j += 2) f29_normalise(v[j]);       // the sums of the last stage
  f29_reduce_partial(v[0]);
#pragma unroll
  for (int j = 0; j < K; ++j) lds_store29<TE>(lds, base + ((uint32_t)j << lgq), v[j]);
}
template <uint32_t TE, uint32_t NT, int GM = 3> __device__ __forceinline__ void tile_dif29(uint32_t* lds, uint32_t lgL, uint32_t T, const char* __restrict__ inner) {
#if defined(ALEO_NTT_PROBE) && (ALEO_NTT_PROBE == 1 || ALEO_NTT_PROBE == 2)
  return;                                                   // timing probe: memory phases only
#endif
  const uint32_t total = T << lgL;
  uint32_t s = 0;
  if constexpr (GM == 2) {                                // two-stage groups: twice the lanes per tile (four waves per SIMD on a 4096-element tile)
    for (; s + 2 <= lgL; s += 2) {
      for (uint32_t sb = threadIdx.x; sb < (total >> 2); sb += NT) dif_group29<TE, 2>(lds, lgL, s, sb, inner);
      __syncthreads();
    }
    if (lgL - s == 1) {
      for (uint32_t sb = threadIdx.x; sb < (total >> 1); sb += NT) dif_group29<TE, 1>(lds, lgL, s, sb, inner);
      __syncthreads();
    }
    return;
  }
  for (; s + 3 <= lgL; s += 3) {
    for (uint32_t sb = threadIdx.x; sb < (total >> 3); sb += NT) dif_group29<TE, 3>(lds, lgL, s, sb, inner);
    __syncthreads();
  }
  if (lgL - s == 2) {
    for (uint32_t sb = threadIdx.x; sb < (total >> 2); sb += NT) dif_group29<TE, 2>(lds, lgL, s, sb, inner);
    __syncthreads();
  } else if (lgL - s == 1) {
    for (uint32_t sb = threadIdx.x; sb < (total >> 1); sb += NT) dif_group29<TE, 1>(lds, lgL, s, sb, inner);
    __syncthreads();
  }
}
// k_ntt_strided on 29-bit limbs: same view, same tiles, same index maps; one lane loads and repacks a whole element
template <uint32_t TE, uint32_t NT, int GM = 3>
__global__ void __launch_bounds__(NT) k_ntt29_strided(const char* src, char* dst, uint32_t lgL, uint32_t lgBn, uint32_t lgT,
                                                       uint32_t tw_scale, uint32_t lg_n, uint32_t lo_bits, const char* __restrict__ inner,
                                                       const char* __restrict__ tw_hi, const char* __restrict__ tw_lo,
                                                       const char* __restrict__ cs_hi, const char* __restrict__ cs_lo, int pre_coset,
                                                       const char* __restrict__ direct, NttSrc ext) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  src = ext.p ? ext.p + (size_t)blockIdx.y * ext.stride * 32 : src + ((size_t)blockIdx.y << (lg_n + 5)); dst += (size_t)blockIdx.y << (lg_n + 5);
  const uint32_t L = 1u << lgL, T = 1u << lgT, tiles_per_a = 1u << (lgBn - lgT);
  const uint32_t a = blockIdx.x / tiles_per_a, b0 = (blockIdx.x % tiles_per_a) << lgT;
  for (uint32_t elem = threadIdx.x; elem < T * L; elem += NT) {
    const uint32_t t = elem & (T - 1u), l = elem >> lgT;
    const size_t gi = ((((size_t)a << lgL) + l) << lgBn) + b0 + t;
#if defined(ALEO_NTT_PROBE) && ALEO_NTT_PROBE == 3
    F29 x; for (int i = 0; i < 9; ++i) x.v[i] = (uint32_t)(gi * 2654435761u + i) & 0x1fffffffu;      // timing probe: no HBM reads
#else
    F29 x;
    if (!ext.p || gi < ext.len) x = f29_load_packed(src + gi * 32);
    else { for (int i = 0; i < 9; ++i) x.v[i] = 0u; }
#endif
    if (pre_coset) x = f29_mul(x, two_level29(cs_hi, cs_lo, (l << lgBn) + b0 + t, lo_bits));      // coset_fft: x[j] *= g^j (only the first pass: A == 1, j = l*Bn + b)
    lds_store29<TE>(lds, t * L + l, x);
  }
  __syncthreads();
  tile_dif29<TE, NT, GM>(lds, lgL, T, inner);
  const uint32_t nmask = (lg_n >= 32) ? 0xffffffffu : ((1u << lg_n) - 1u);
  for (uint32_t e = threadIdx.x; e < T * L; e += NT) {
    const uint32_t t = e & (T - 1u), k = e >> lgT;
    F29 x = lds_load29<TE>(lds, t * L + bitrev(k, lgL));
    const uint32_t b = b0 + t;
    const size_t gi = ((((size_t)a << lgL) + k) << lgBn) + b;
#if !(defined(ALEO_NTT_PROBE) && ALEO_NTT_PROBE == 2)
    if (direct) x = f29_mul(x, f29_load_packed(direct + gi * 32));
    else x = f29_mul(x, two_level29(tw_hi, tw_lo, (uint32_t)(((uint64_t)tw_scale * k * b) & nmask), lo_bits));
#endif
#if defined(ALEO_NTT_PROBE) && ALEO_NTT_PROBE == 3
    if (x.v[0] == 0x12345678u && x.v[5] == 0x1u)              // timing probe: (practically) no HBM writes
#endif
    store_fp<Fr>(dst + gi * 32, f29_to_fr(x));               // < 1.1 r: the next pass repacks it
  }
}
// k_ntt_final on 29-bit limbs.  An output that passes a last product (the coset power or n^-1) is below 1.1 r: one conditional subtraction makes it
// canonical; the others (plain forward transform) are reduced below 3 r first.
template <uint32_t TE, uint32_t NT, int GM = 3>
__global__ void __launch_bounds__(NT) k_ntt29_final(const char* src, char* dst, uint32_t lgL, uint32_t lgN1, uint32_t lgN2, uint32_t lgT,
                                                     uint32_t lo_bits, const char* __restrict__ inner, const char* __restrict__ cs_hi, const char* __restrict__ cs_lo,
                                                     int pre_coset, int post_coset, int do_scale, F29Arg scale, NttSrc ext) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  src = ext.p ? ext.p + (size_t)blockIdx.y * ext.stride * 32 : src + ((size_t)blockIdx.y << (lgL + lgN1 + lgN2 + 5)); dst += (size_t)blockIdx.y << (lgL + lgN1 + lgN2 + 5);
  const uint32_t L = 1u << lgL, T = 1u << lgT;
  const uint32_t tiles_k1 = 1u << (lgN1 - lgT);
  const uint32_t k2 = blockIdx.x / tiles_k1, k10 = (blockIdx.x % tiles_k1) << lgT;
  for (uint32_t elem = threadIdx.x; elem < T * L; elem += NT) {
    const uint32_t l = elem & (L - 1u), t = elem >> lgL;
    const size_t row = ((size_t)(k10 + t) << lgN2) + k2;
#if defined(ALEO_NTT_PROBE) && ALEO_NTT_PROBE == 3
    F29 x; for (int i = 0; i < 9; ++i) x.v[i] = (uint32_t)((row + l) * 2654435761u + i) & 0x1fffffffu;
#else
    F29 x;
    if (!ext.p || (row << lgL) + l < ext.len) x = f29_load_packed(src + ((row << lgL) + l) * 32);
    else { for (int i = 0; i < 9; ++i) x.v[i] = 0u; }
#endif
    if (pre_coset) x = f29_mul(x, two_level29(cs_hi, cs_lo, l, lo_bits));      // single-pass coset_fft: j = l
    lds_store29<TE>(lds, t * L + l, x);
  }
  __syncthreads();
  tile_dif29<TE, NT, GM>(lds, lgL, T, inner);
  F29 sc;
#pragma unroll
  for (int i = 0; i < 9; ++i) sc.v[i] = scale.v[i];
  for (uint32_t e = threadIdx.x; e < T * L; e += NT) {
    const uint32_t t = e & (T - 1u), k = e >> lgT;
    F29 x = lds_load29<TE>(lds, t * L + bitrev(k, lgL));
    const size_t o = (size_t)(k10 + t) + (((size_t)k2 + ((size_t)k << lgN2)) << lgN1);
    if (post_coset) x = f29_mul(x, two_level29(cs_hi, cs_lo, (uint32_t)o, lo_bits));       // g^-o * n^-1
    else if (do_scale) x = f29_mul(x, sc);
    else f29_reduce_partial(x);                               // a plain forward transform has no last product: < 4.5 r -> < 3 r here, two conditional subtractions below
    Fr y = f29_to_fr(x);
    if (!post_coset && !do_scale) y = Fr::cond_sub<2>(y);
#if defined(ALEO_NTT_PROBE) && ALEO_NTT_PROBE == 3
    if (y.v[0] == 0x12345678u && y.v[5] == 0x1u)
#endif
    store_fp<Fr>(dst + o * 32, Fr::cond_sub<1>(y));
  }
}
__global__ void __launch_bounds__(256) k_build_direct29(char* __restrict__ out, uint32_t lg_n, uint32_t lgBn, uint32_t lo_bits, const char* __restrict__ tw_hi,
                                                        const char* __restrict__ tw_lo) {
  const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gi >> lg_n) return;
  const uint32_t k = (uint32_t)(gi >> lgBn), b = (uint32_t)(gi & (((size_t)1 << lgBn) - 1));
  const uint32_t nmask = (lg_n >= 32) ? 0xffffffffu : ((1u << lg_n) - 1u);
  store_fp<Fr>(out + gi * 32, Fr::cond_sub<1>(f29_to_fr(two_level29(tw_hi, tw_lo, (uint32_t)(((uint64_t)k * b) & nmask), lo_bits))));
}

// direct[(k << lgBn) + b] = w_n^(k * b): the inter-pass factor of a two-pass transform, one entry per element (built once per
// (size, direction) on first use; 32 n bytes of HBM buy one product per element per transform)
__global__ void __launch_bounds__(256) k_build_direct(char* __restrict__ out, uint32_t lg_n, uint32_t lgBn, uint32_t lo_bits, const char* __restrict__ tw_hi,
                                                      const char* __restrict__ tw_lo) {
  const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gi >> lg_n) return;
  const uint32_t k = (uint32_t)(gi >> lgBn), b = (uint32_t)(gi & (((size_t)1 << lgBn) - 1));
  const uint32_t nmask = (lg_n >= 32) ? 0xffffffffu : ((1u << lg_n) - 1u);
  store_fp<Fr>(out + gi * 32, Fr::reduce(two_level(tw_hi, tw_lo, (uint32_t)(((uint64_t)k * b) & nmask), lo_bits)));
}

__global__ void __launch_bounds__(256) k_bitrev_copy(const char* __restrict__ src, char* __restrict__ dst, uint32_t lg_n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ((size_t)1 << lg_n)) return;
  src += (size_t)blockIdx.y << (lg_n + 5); dst += (size_t)blockIdx.y << (lg_n + 5);
  uint32_t j = bitrev((uint32_t)i, lg_n);
  const uint4* s = (const uint4*)(src + i * 32); uint4* d = (uint4*)(dst + (size_t)j * 32);
  d[0] = s[0]; d[1] = s[1];
}

// ---- host side: tables -----------------------------------------------------------------------------
static int32_t upload_fr(void** d, const std::vector<host::HFr>& v) {
  HIPCHK(hipMalloc(d, v.size() * 32));
  HIPCHK(hipMemcpy(*d, v.data(), v.size() * 32, hipMemcpyHostToDevice));
  return ALEO_MI355X_OK;
}

static int32_t build_tables(NttTables* t, uint32_t lg_n, int direction) {
  using namespace host;
  t->lg_n = lg_n; t->lo_bits = (lg_n + 1) / 2;
  const uint32_t hi_bits = lg_n - t->lo_bits;
  HFr root; std::memcpy(root.l, FR_TWO_ADIC_ROOT_CANON, 32); root = HFr::to_mont(root);
  // w_n = TWO_ADIC_ROOT^(2^(47 - lg_n)); w_2048 likewise
  HFr wn = root; for (uint32_t i = lg_n; i < (uint32_t)FR_TWO_ADICITY; ++i) wn = HFr::sqr(wn);
  HFr w1k = root; for (uint32_t i = INNER_MAX_LG; i < (uint32_t)FR_TWO_ADICITY; ++i) w1k = HFr::sqr(w1k);
  HFr g = HFr::from_u64(FR_GENERATOR);
  HFr ninv = HFr::inv(HFr::from_u64((uint64_t)1 << lg_n));
  if (direction == ALEO_NTT_INVERSE) { wn = HFr::inv(wn); w1k = HFr::inv(w1k); g = HFr::inv(g); }
  const HFr lo_scale = direction == ALEO_NTT_INVERSE ? ninv : HFr::one();
  std::memcpy(t->scale, ninv.l, 32);
  std::vector<HFr> inner(1024), hi((size_t)1 << hi_bits), lo((size_t)1 << t->lo_bits), chi((size_t)1 << hi_bits), clo((size_t)1 << t->lo_bits);
  inner[0] = HFr::one(); for (size_t i = 1; i < inner.size(); ++i) inner[i] = HFr::mul(inner[i - 1], w1k);
  auto fill = [&](std::vector<HFr>& v, HFr first, HFr step) { v[0] = first; for (size_t i = 1; i < v.size(); ++i) v[i] = HFr::mul(v[i - 1], step); };
  HFr wn_hi = wn, g_hi = g;
  for (uint32_t i = 0; i < t->lo_bits; ++i) { wn_hi = HFr::sqr(wn_hi); g_hi = HFr::sqr(g_hi); }
  fill(lo, HFr::one(), wn); fill(hi, HFr::one(), wn_hi);
  fill(clo, lo_scale, g); fill(chi, HFr::one(), g_hi);
  int32_t rc;
  if ((rc = upload_fr(&t->d_inner, inner))) return rc;
  if ((rc = upload_fr(&t->d_tw_hi, hi))) return rc;
  if ((rc = upload_fr(&t->d_tw_lo, lo))) return rc;
  if ((rc = upload_fr(&t->d_cs_hi, chi))) return rc;
  if ((rc = upload_fr(&t->d_cs_lo, clo))) return rc;
  // the 29-bit-limb kernels' copies: x * 2^261 = (x * 2^256) * 2^5 — one host product by 32 per entry; the number itself is what is uploaded
  const HFr k32 = HFr::from_u64(32);
  auto form29 = [&](std::vector<HFr> v) { for (auto& e : v) e = HFr::mul(e, k32); return v; };
  if ((rc = upload_fr(&t->d_tw_hi29, form29(hi)))) return rc;
  if ((rc = upload_fr(&t->d_tw_lo29, form29(lo)))) return rc;
  if ((rc = upload_fr(&t->d_cs_hi29, form29(chi)))) return rc;
  if ((rc = upload_fr(&t->d_cs_lo29, form29(clo)))) return rc;
  auto limbs29 = [](const HFr& e, uint32_t* out) {           // canonical number -> 9 x 29-bit limbs
    for (int i = 0; i < 9; ++i) { const int bit = 29 * i, w = bit >> 6, sh = bit & 63; uint64_t x = e.l[w] >> sh; if (sh > 35 && w + 1 < 4) x |= e.l[w + 1] << (64 - sh); out[i] = (uint32_t)x & 0x1fffffffu; }
  };
  {
    std::vector<uint32_t> in29(12 * inner.size(), 0);
    for (size_t i = 0; i < inner.size(); ++i) limbs29(HFr::mul(inner[i], k32), &in29[12 * i]);
    HIPCHK(hipMalloc(&t->d_inner29, in29.size() * 4));
    HIPCHK(hipMemcpy(t->d_inner29, in29.data(), in29.size() * 4, hipMemcpyHostToDevice));
  }
  limbs29(HFr::mul(ninv, k32), t->scale29);
  return ALEO_MI355X_OK;
}

template <uint32_t TE, uint32_t NT, int GM = 3>
static int32_t run_passes(Ctx* c, char* buf, char* tmp, uint32_t lg_n, uint32_t batch, const NttTables* t, int pre_coset, int post_coset, int do_scale, FrArg sc, hipStream_t s, NttSrc ext = NttSrc{}) {
  constexpr uint32_t lgTE = TE == 4096 ? 12 : (TE == 2048 ? 11 : 9);
  static_assert(TE == 4096 || TE == 2048 || TE == 512, "tile sizes with a kernel instance");
  constexpr size_t lds_bytes = (size_t)TE * 32;
  constexpr int attr_bit = TE == 4096 ? 2 : 1;          // the LDS limit is a property of (kernel, device): remembered per device
  if (lds_bytes > 65536 && !(c->dev->ntt_attr_mask.load() & attr_bit)) {
    HIPCHK(hipFuncSetAttribute((const void*)k_ntt_strided<TE, NT, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    HIPCHK(hipFuncSetAttribute((const void*)k_ntt_final<TE, NT, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    c->dev->ntt_attr_mask.fetch_or(attr_bit);
  }
  const char* inner = (const char*)t->d_inner; const char* twh = (const char*)t->d_tw_hi; const char* twl = (const char*)t->d_tw_lo;
  const char* csh = (const char*)t->d_cs_hi; const char* csl = (const char*)t->d_cs_lo;
  const uint32_t maxL = lgTE < INNER_MAX_LG ? lgTE : INNER_MAX_LG;            // longest in-LDS transform with this tile
  uint32_t npass = lg_n <= maxL ? 1 : (lg_n <= 2 * maxL ? 2 : 3);
  uint32_t s1 = 0, s2 = 0, s3 = 0;
  if (npass == 1) s3 = lg_n;
  else if (npass == 2) { s1 = (lg_n + 1) / 2; s3 = lg_n - s1; }
  else { s1 = (lg_n + 2) / 3; s2 = (lg_n - s1 + 1) / 2; s3 = lg_n - s1 - s2; }
  auto lgT_for = [](uint32_t lgL, uint32_t lg_limit) { uint32_t lgT = lgTE - lgL; return lgT < lg_limit ? lgT : lg_limit; };
  if (npass == 1) {
    hipLaunchKernelGGL((k_ntt_final<TE, NT, GM>), dim3(1, batch), dim3(NT), lds_bytes, s, buf, buf, s3, 0u, 0u, 0u, t->lo_bits, inner, csh, csl, pre_coset, post_coset, do_scale, sc, ext);
  } else if (npass == 2) {
    uint32_t lgBn = s3, lgT = lgT_for(s1, lgBn);
    const char* direct = nullptr;
    if (lg_n >= 12 && lg_n <= 20) {                 // (2^21, 2^22: the extra 32 B per element of HBM reads cost what the product saves) lazily built, shared by all slots; the split (s1, s3) is a function of lg_n and the tile, stored with the table
      std::lock_guard<std::mutex> lk(c->dev->mu);
      NttTables* tm = const_cast<NttTables*>(t);
      if (!tm->d_direct) {
        HIPCHK(hipMalloc(&tm->d_direct, (size_t)32 << lg_n));
        hipLaunchKernelGGL(k_build_direct, dim3((uint32_t)((((size_t)1 << lg_n) + 255) / 256)), dim3(256), 0, s, (char*)tm->d_direct, lg_n, lgBn, t->lo_bits, twh, twl);
        HIPCHK(hipStreamSynchronize(s));           // once per (size, direction): later calls on other streams may use it at once
        tm->direct_lgBn = lgBn;
      }
      if (tm->direct_lgBn == lgBn) direct = (const char*)tm->d_direct;      // (always: the split of a two-pass size is (lg_n + 1) / 2 whatever the tile)
    }
    hipLaunchKernelGGL((k_ntt_strided<TE, NT, GM>), dim3(1u << (lgBn - lgT), batch), dim3(NT), lds_bytes, s, buf, tmp, s1, lgBn, lgT, 1u, lg_n, t->lo_bits, inner, twh, twl, csh, csl, pre_coset, direct, ext);
    uint32_t lgTf = lgT_for(s3, s1);
    hipLaunchKernelGGL((k_ntt_final<TE, NT, GM>), dim3(1u << (s1 - lgTf), batch), dim3(NT), lds_bytes, s, tmp, buf, s3, s1, 0u, lgTf, t->lo_bits, inner, csh, csl, 0, post_coset, do_scale, sc, NttSrc{});
  } else {
    uint32_t lgBn1 = s2 + s3, lgT1 = lgT_for(s1, lgBn1);
    hipLaunchKernelGGL((k_ntt_strided<TE, NT, GM>), dim3(1u << (lgBn1 - lgT1), batch), dim3(NT), lds_bytes, s, buf, buf, s1, lgBn1, lgT1, 1u, lg_n, t->lo_bits, inner, twh, twl, csh, csl, pre_coset, (const char*)nullptr, ext);
    uint32_t lgBn2 = s3, lgT2 = lgT_for(s2, lgBn2);
    hipLaunchKernelGGL((k_ntt_strided<TE, NT, GM>), dim3((1u << s1) << (lgBn2 - lgT2), batch), dim3(NT), lds_bytes, s, buf, tmp, s2, lgBn2, lgT2, 1u << s1, lg_n, t->lo_bits, inner, twh, twl, csh, csl, 0, (const char*)nullptr, NttSrc{});
    uint32_t lgTf = lgT_for(s3, s1);
    hipLaunchKernelGGL((k_ntt_final<TE, NT, GM>), dim3((1u << s2) << (s1 - lgTf), batch), dim3(NT), lds_bytes, s, tmp, buf, s3, s1, s2, lgTf, t->lo_bits, inner, csh, csl, 0, post_coset, do_scale, sc, NttSrc{});
  }
  return ALEO_MI355X_OK;
}


// run_passes on the 29-bit-limb kernels (tiles of 9 planes: 144 KiB / 72 KiB); same pass splits, same launch geometry
template <uint32_t TE, uint32_t NT, int GM = 3>
static int32_t run_passes29(Ctx* c, char* buf, char* tmp, uint32_t lg_n, uint32_t batch, const NttTables* t, int pre_coset, int post_coset, int do_scale, hipStream_t s, NttSrc ext = NttSrc{}) {
  constexpr uint32_t lgTE = TE == 4096 ? 12 : 11;
  static_assert(TE == 4096 || TE == 2048, "tile sizes with a kernel instance");
  constexpr size_t lds_bytes = (size_t)TE * 36;
  constexpr int attr_bit = (TE == 4096 ? 8 : 4) << (GM == 2 ? 2 : 0);
  if (!(c->dev->ntt_attr_mask.load() & attr_bit)) {
    HIPCHK(hipFuncSetAttribute((const void*)k_ntt29_strided<TE, NT, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    HIPCHK(hipFuncSetAttribute((const void*)k_ntt29_final<TE, NT, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    c->dev->ntt_attr_mask.fetch_or(attr_bit);
  }
  const char* inner = (const char*)t->d_inner29; const char* twh = (const char*)t->d_tw_hi29; const char* twl = (const char*)t->d_tw_lo29;
  const char* csh = (const char*)t->d_cs_hi29; const char* csl = (const char*)t->d_cs_lo29;
  F29Arg sc; std::memcpy(sc.v, t->scale29, 36);
  const uint32_t maxL = lgTE < INNER_MAX_LG ? lgTE : INNER_MAX_LG;
  uint32_t npass = lg_n <= maxL ? 1 : (lg_n <= 2 * maxL ? 2 : 3);
  uint32_t s1 = 0, s2 = 0, s3 = 0;
  if (npass == 1) s3 = lg_n;
  else if (npass == 2) { s1 = (lg_n + 1) / 2; s3 = lg_n - s1; }
  else { s1 = (lg_n + 2) / 3; s2 = (lg_n - s1 + 1) / 2; s3 = lg_n - s1 - s2; }
  auto lgT_for = [](uint32_t lgL, uint32_t lg_limit) { uint32_t lgT = lgTE - lgL; return lgT < lg_limit ? lgT : lg_limit; };
  if (npass == 1) {
    hipLaunchKernelGGL((k_ntt29_final<TE, NT, GM>), dim3(1, batch), dim3(NT), lds_bytes, s, buf, buf, s3, 0u, 0u, 0u, t->lo_bits, inner, csh, csl, pre_coset, post_coset, do_scale, sc, ext);
  } else if (npass == 2) {
    uint32_t lgBn = s3, lgT = lgT_for(s1, lgBn);
    const char* direct = nullptr;
    static const uint32_t direct_max = [] { const char* e = std::getenv("ALEO_MI355X_NTT_DIRECT_MAX"); const int k = e ? std::atoi(e) : 21; return (uint32_t)(k >= 0 && k <= 22 ? k : 21); }();      // 2^21: 0.286 -> 0.271 ms with the table, 2^22: 0.549 -> 0.564 ms (the extra 32 B per element of HBM reads cost more than the product; round 4: fetching a lane's eight table entries in one go ahead of the products changes nothing — 2^22 0.581 / 0.588 ms without / with the table, 2^21 0.277 / 0.283 ms with the entries fetched late / ahead)
    if (lg_n >= 12 && lg_n <= direct_max) {
      std::lock_guard<std::mutex> lk(c->dev->mu);
      NttTables* tm = const_cast<NttTables*>(t);
      if (!tm->d_direct29) {
        HIPCHK(hipMalloc(&tm->d_direct29, (size_t)32 << lg_n));
        hipLaunchKernelGGL(k_build_direct29, dim3((uint32_t)((((size_t)1 << lg_n) + 255) / 256)), dim3(256), 0, s, (char*)tm->d_direct29, lg_n, lgBn, t->lo_bits, twh, twl);
        HIPCHK(hipStreamSynchronize(s));
        tm->direct29_lgBn = lgBn;
      }
      if (tm->direct29_lgBn == lgBn) direct = (const char*)tm->d_direct29;
    }
    hipLaunchKernelGGL((k_ntt29_strided<TE, NT, GM>), dim3(1u << (lgBn - lgT), batch), dim3(NT), lds_bytes, s, buf, tmp, s1, lgBn, lgT, 1u, lg_n, t->lo_bits, inner, twh, twl, csh, csl, pre_coset, direct, ext);
    uint32_t lgTf = lgT_for(s3, s1);
    hipLaunchKernelGGL((k_ntt29_final<TE, NT, GM>), dim3(1u << (s1 - lgTf), batch), dim3(NT), lds_bytes, s, tmp, buf, s3, s1, 0u, lgTf, t->lo_bits, inner, csh, csl, 0, post_coset, do_scale, sc, NttSrc{});
  } else {
    uint32_t lgBn1 = s2 + s3, lgT1 = lgT_for(s1, lgBn1);
    hipLaunchKernelGGL((k_ntt29_strided<TE, NT, GM>), dim3(1u << (lgBn1 - lgT1), batch), dim3(NT), lds_bytes, s, buf, buf, s1, lgBn1, lgT1, 1u, lg_n, t->lo_bits, inner, twh, twl, csh, csl, pre_coset, (const char*)nullptr, ext);
    uint32_t lgBn2 = s3, lgT2 = lgT_for(s2, lgBn2);
    hipLaunchKernelGGL((k_ntt29_strided<TE, NT, GM>), dim3((1u << s1) << (lgBn2 - lgT2), batch), dim3(NT), lds_bytes, s, buf, tmp, s2, lgBn2, lgT2, 1u << s1, lg_n, t->lo_bits, inner, twh, twl, csh, csl, 0, (const char*)nullptr, NttSrc{});
    uint32_t lgTf = lgT_for(s3, s1);
    hipLaunchKernelGGL((k_ntt29_final<TE, NT, GM>), dim3((1u << s2) << (s1 - lgTf), batch), dim3(NT), lds_bytes, s, tmp, buf, s3, s1, s2, lgTf, t->lo_bits, inner, csh, csl, 0, post_coset, do_scale, sc, NttSrc{});
  }
  return ALEO_MI355X_OK;
}

// dst[c][r] = src[r][c] for 32-byte elements (rows x cols -> cols x rows): the three layout changes of a 4-step transform (columns contiguous for the
// column transforms, rows for the row transforms, natural order out).  32 x 32-element tiles through LDS: both sides move 1-KiB runs.
__global__ void __launch_bounds__(256) k_transpose32(const char* __restrict__ src, char* __restrict__ dst, uint64_t rows, uint64_t cols) {
  __shared__ uint4 tile[32][33][2];
  const uint64_t r0 = (uint64_t)blockIdx.y * 32, c0 = (uint64_t)blockIdx.x * 32;
  const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;                     // 32 x 8 threads
  for (uint32_t j = ty; j < 32; j += 8) {
    const uint64_t r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) { const uint4* e = (const uint4*)(src + (r * cols + c) * 32); tile[j][tx][0] = e[0]; tile[j][tx][1] = e[1]; }
  }
  __syncthreads();
  for (uint32_t j = ty; j < 32; j += 8) {
    const uint64_t c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) { uint4* e = (uint4*)(dst + (c * rows + r) * 32); e[0] = tile[tx][j][0]; e[1] = tile[tx][j][1]; }
  }
}
int32_t fr_transpose(Ctx* c, void* d_dst, const void* d_src, uint64_t rows, uint64_t cols, hipStream_t s) {
  (void)c;
  if (!rows || !cols) return ALEO_MI355X_OK;
  if ((rows + 31) / 32 > 65535) { g_last_error = "fr_transpose: more than 2^21 rows"; return ALEO_MI355X_ERR_BAD_ARG; }
  hipLaunchKernelGGL(k_transpose32, dim3((uint32_t)((cols + 31) / 32), (uint32_t)((rows + 31) / 32)), dim3(256), 0, s, (const char*)d_src, (char*)d_dst, rows, cols);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

// ---- sharded (4-step) transform support: per-element factors over a 2-D block of the size-n index space -----------
// mode 0: x[r][c] *= w_n^((row0 + r) * (col0 + c))       (the twiddle between the column and the row transforms)
// mode 1: x[r][c] *= g^((row0 + r) * ld + col0 + c)      (coset shift of a block of the coefficient matrix)
// inverse direction: w^-1 / g^-1 (tables of the inverse domain; the n^-1 folded into its coset table is cancelled by K = n).
__global__ void __launch_bounds__(256) k_grid_scale(char* __restrict__ data, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0, uint64_t ld,
                                                    int mode, uint32_t lg_n, uint32_t lo_bits, const char* __restrict__ hi, const char* __restrict__ lo,
                                                    int use_k, FrArg K) {
  const uint64_t total = rows * cols, nmask = ((uint64_t)1 << lg_n) - 1u;
  Fr k; for (int i = 0; i < 8; ++i) k.v[i] = K.v[i];
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
    const uint64_t r = row0 + i / cols, c = col0 + i % cols;
    const uint32_t e = (uint32_t)((mode == 0 ? r * c : r * ld + c) & nmask);
    Fr f = two_level(hi, lo, e, lo_bits);                     // < 2r
    if (use_k) f = Fr::mul(f, k);
    Fr x = load_fp<Fr>(data + i * 32);
    store_fp<Fr>(data + i * 32, Fr::reduce(Fr::mul(x, f)));
  }
}

static int32_t get_tables(Ctx* c, uint32_t lg_n, int32_t direction, NttTables** out);

int32_t fr_grid_scale(Ctx* c, void* d_data, uint32_t lg_n, uint64_t rows, uint64_t cols, uint64_t row0, uint64_t col0, uint64_t ld, int32_t mode,
                      int32_t direction, hipStream_t s) {
  if (rows == 0 || cols == 0) return ALEO_MI355X_OK;
  NttTables* t = nullptr; int32_t rc;
  if ((rc = get_tables(c, lg_n, direction, &t))) return rc;
  FrArg K; std::memset(K.v, 0, 32); int use_k = 0;
  if (mode == 1 && direction == ALEO_NTT_INVERSE) {           // cs_lo of the inverse domain carries n^-1: multiply it back out
    host::HFr nn = host::HFr::from_u64((uint64_t)1 << lg_n); std::memcpy(K.v, nn.l, 32); use_k = 1;
  }
  const uint64_t total = rows * cols; const uint32_t grid = (uint32_t)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  const char* hi = (const char*)(mode == 0 ? t->d_tw_hi : t->d_cs_hi); const char* lo = (const char*)(mode == 0 ? t->d_tw_lo : t->d_cs_lo);
  hipLaunchKernelGGL(k_grid_scale, dim3(grid), dim3(256), 0, s, (char*)d_data, rows, cols, row0, col0, ld, mode, lg_n, t->lo_bits, hi, lo, use_k, K);
  HIPCHK(hipGetLastError());
  return ALEO_MI355X_OK;
}

static int32_t get_tables(Ctx* c, uint32_t lg_n, int32_t direction, NttTables** out) {
  uint64_t key = ((uint64_t)lg_n << 1) | (uint64_t)direction;
  std::lock_guard<std::mutex> lk(c->dev->mu);       // tables are shared by all slots and immutable once built
  auto it = c->dev->ntt_tables.find(key);
  if (it == c->dev->ntt_tables.end()) {
    NttTables* t = new NttTables();
    int32_t rc = build_tables(t, lg_n, direction);
    if (rc) { delete t; return rc; }
    c->dev->ntt_tables[key] = t; *out = t;
  } else *out = it->second;
  return ALEO_MI355X_OK;
}

static int32_t ntt_run_chunk(Ctx* c, void* d_inout, uint32_t lg_n, uint32_t batch, int32_t order, int32_t direction, int32_t type, hipStream_t s, NttSrc ext = NttSrc{});
// `batch` independent transforms of 2^lg_n elements each, contiguous in d_inout (blockIdx.y walks them)
int32_t ntt_run(Ctx* c, void* d_inout, uint32_t lg_n, size_t batch_total, int32_t order, int32_t direction, int32_t type, hipStream_t s) {
  if (lg_n == 0 || batch_total == 0) return ALEO_MI355X_OK;     // n = 1: every variant is the identity (g^0 = 1, 1^-1 = 1)
  const size_t n = (size_t)1 << lg_n;
  for (size_t b0 = 0; b0 < batch_total; b0 += 32768) {            // gridDim.y <= 65535
    const uint32_t batch = (uint32_t)(batch_total - b0 < 32768 ? batch_total - b0 : 32768);
    int32_t rcb = ntt_run_chunk(c, (char*)d_inout + b0 * n * 32, lg_n, batch, order, direction, type, s);
    if (rcb) return rcb;
  }
  return ALEO_MI355X_OK;
}

// The same with the first pass reading transform b's coefficients from d_src + b * src_stride elements, zero beyond src_len (natural order in and out; d_out
// must not overlap d_src): what `memset + copy + ntt_run` did in three or more launches.
int32_t ntt_run_from(Ctx* c, void* d_out, const void* d_src, size_t src_stride, size_t src_len, uint32_t lg_n, size_t batch_total, int32_t direction, int32_t type, hipStream_t s) {
  if (batch_total == 0) return ALEO_MI355X_OK;
  const size_t n = (size_t)1 << lg_n;
  if (src_len > n || src_stride >= (1ull << 32) || !d_src || !d_out) { g_last_error = "ntt_run_from: bad argument"; return ALEO_MI355X_ERR_BAD_ARG; }
  if (lg_n == 0) { for (size_t b = 0; b < batch_total; ++b) { if (src_len) HIPCHK(hipMemcpyAsync((char*)d_out + b * 32, (const char*)d_src + b * src_stride * 32, 32, hipMemcpyDeviceToDevice, s)); else HIPCHK(hipMemsetAsync((char*)d_out + b * 32, 0, 32, s)); } return ALEO_MI355X_OK; }
  for (size_t b0 = 0; b0 < batch_total; b0 += 32768) {
    const uint32_t batch = (uint32_t)(batch_total - b0 < 32768 ? batch_total - b0 : 32768);
    NttSrc ext; ext.p = (const char*)d_src + b0 * src_stride * 32; ext.len = (uint32_t)src_len; ext.stride = (uint32_t)src_stride;
    const int32_t rcb = ntt_run_chunk(c, (char*)d_out + b0 * n * 32, lg_n, batch, ALEO_NTT_ORDER_NN, direction, type, s, ext);
    if (rcb) return rcb;
  }
  return ALEO_MI355X_OK;
}

// Calls of at most 2^wide_lg() elements in all (transforms of 2^10 … 2^18 points) take the one-butterfly-per-lane tiles: 2^12 52 -> 22 us, 2^15 64 -> 27 us,
// 3 x 2^15 67 -> 30 us, 8 x 2^16 85 -> 72 us; from 2^20 elements on the three-stage register groups win again (8 x 2^17: 138 against 142 us)
// (tools/ntt_small_probe.py, profiles/r02_ntt_small_probe.jsonl).  ALEO_MI355X_NTT_WIDE_LG overrides the cut, 0 = never.
static uint32_t wide_lg() { static const uint32_t v = [] { const char* e = std::getenv("ALEO_MI355X_NTT_WIDE_LG"); int k = e ? std::atoi(e) : 19; return (uint32_t)(k >= 0 && k <= 24 ? k : 19); }(); return v; }
static int32_t ntt_run_chunk(Ctx* c, void* d_inout, uint32_t lg_n, uint32_t batch, int32_t order, int32_t direction, int32_t type, hipStream_t s, NttSrc ext) {
  const size_t n = (size_t)1 << lg_n, bytes = n * 32 * batch;
  int32_t rc;
  NttTables* t = nullptr;
  if ((rc = get_tables(c, lg_n, direction, &t))) return rc;
  if ((rc = scratch_acquire(c, c->ntt_tmp, bytes, s))) return rc;      // ordered after the slot's previous asynchronous user
  char* buf = (char*)d_inout; char* tmp = c->ntt_tmp.as<char>();
  const bool in_rev = (order == ALEO_NTT_ORDER_RN || order == ALEO_NTT_ORDER_RR), out_rev = (order == ALEO_NTT_ORDER_NR || order == ALEO_NTT_ORDER_RR);
  const dim3 gperm((uint32_t)((n + 255) / 256), batch);
  if (in_rev) {
    hipLaunchKernelGGL(k_bitrev_copy, gperm, dim3(256), 0, s, buf, tmp, lg_n);
    HIPCHK(hipMemcpyAsync(buf, tmp, bytes, hipMemcpyDeviceToDevice, s));
  }
  const int coset = (type == ALEO_NTT_COSET), inv = (direction == ALEO_NTT_INVERSE);
  const int pre_coset = coset && !inv, post_coset = coset && inv, do_scale = inv && !coset;   // cs_lo carries n^-1 for coset_ifft
  FrArg sc; std::memcpy(sc.v, t->scale, 32);
  // 128 KiB tiles: 2^20..2^22 (at 2^19 they are only 128 blocks for 256 CUs: 242 against 343 GB/s with 64 KiB tiles); beyond 2^22 three
  // passes are needed either way and two 64 KiB blocks per CU overlap their HBM phases better (2^24: 3.5 vs 4.0 ms)
  // small transforms are latency-bound on the few blocks a 2048-element tile leaves them (2^16: 32 blocks on 256 CUs): 512-element
  // tiles of one wave each spread them over the chip (2^16: 0.085 -> see profiles/); batches already have the blocks
  static const int force_tile = [] { const char* e = std::getenv("ALEO_MI355X_NTT_TILE"); return e ? std::atoi(e) : 0; }();      // experiments only
  static const bool limbs29 = [] { const char* e = std::getenv("ALEO_MI355X_NTT29"); return !(e && e[0] == '0'); }();      // A/B switch: 0 = the 32-bit-limb kernels everywhere
  const bool big_tile_default = !(lg_n >= 10 && lg_n <= 18 && ((size_t)batch << lg_n) <= ((size_t)1 << 18)) &&
                                !(lg_n >= 10 && lg_n <= 18 && ((size_t)batch << lg_n) <= ((size_t)1 << wide_lg()));
  if (limbs29 && (force_tile == 2048 || force_tile == 4096 || (force_tile == 0 && big_tile_default))) {
    static const int gm = [] { const char* e = std::getenv("ALEO_MI355X_NTT29_GM"); return e ? std::atoi(e) : 3; }();      // experiment knob
    if ((force_tile == 4096 || (force_tile == 0 && lg_n >= 20 && lg_n <= 22)) && gm == 2) rc = run_passes29<4096, 1024, 2>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, s, ext);
    else if (force_tile == 4096 || (force_tile == 0 && lg_n >= 20 && lg_n <= 22)) rc = run_passes29<4096, 512>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, s, ext);
    else rc = run_passes29<2048, 256>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, s, ext);
  }
  else if (force_tile == 2048) rc = run_passes<2048, 256>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, sc, s, ext);
  else if (force_tile == 4096) rc = run_passes<4096, 512>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, sc, s, ext);
  else if (force_tile == 512 && lg_n <= 18) rc = run_passes<512, 64>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, sc, s, ext);
  else if (lg_n >= 20 && lg_n <= 22) rc = run_passes<4096, 512>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, sc, s, ext);
  else if (lg_n >= 10 && lg_n <= 18 && ((size_t)batch << lg_n) <= ((size_t)1 << wide_lg())) rc = run_passes<512, 256, 1>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, sc, s, ext);      // latency form
  else if (lg_n >= 10 && lg_n <= 18 && ((size_t)batch << lg_n) <= ((size_t)1 << 18)) rc = run_passes<512, 64>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, sc, s, ext);
  else rc = run_passes<2048, 256>(c, buf, tmp, lg_n, batch, t, pre_coset, post_coset, do_scale, sc, s, ext);
  if (rc) return rc;
  if (out_rev) {
    hipLaunchKernelGGL(k_bitrev_copy, gperm, dim3(256), 0, s, buf, tmp, lg_n);
    HIPCHK(hipMemcpyAsync(buf, tmp, bytes, hipMemcpyDeviceToDevice, s));
  }
  HIPCHK(hipGetLastError());
  return scratch_release(c, s);
}

}  // namespace aleo_mi355x
